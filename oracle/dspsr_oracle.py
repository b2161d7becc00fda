"""CPU ORACLE (test infrastructure -- NOT product code).

A numpy restatement of the reference algorithm for the one hot path
    8-bit voltages -> dsp::Filterbank (-F N:D) -> dsp::Detection -> dsp::Fold
of demorest/dspsr.  Only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py may import this module; the product (dspsr_amd/) must never do so.

PARITY STATUS: "parity unpinned" at the PSRCHIVE boundary.
  The reference holds no golden vectors / known-answer tests for this path
   and cannot be built here (needs PSRCHIVE + autotools +
  FFTW).  What IS pinned:
    * cross_detect / stokes_detect / optimal_fft_length: checked against the
      reference's own C files compiled unmodified into oracle/_ref (tests/test_oracle_ref.py);
    * the FFT conventions (forward e^{-i}, backward e^{+i}, both unnormalised) are those the
      reference's own CUDA twin requests from cuFFT (Signal/General/FilterbankCUDA.cu:92,232,258)
      and are cross-checked between numpy pocketfft and the independent C FFT in oracle_c.c;
    * everything else is restated line by line with file:line citations below.
  Third-party pieces NOT in /root/reference (PSRCHIVE, version unpinned by configure.ac:74-78):
    FTransform (FFT), Pulsar::Predictor (TEMPO polyco), JenetAnderson98 (8-bit LUT spacing), MJD.

All citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------------------
# Observation + DADA header  (Kernel/Classes/ASCIIObservation.C:82-415, DADAFile.C:33-180)
# --------------------------------------------------------------------------------------

DADA_HDR_SIZE = 4096  # DADAFile.C:33-110 default header size


@dataclass
class Observation:
    """Minimal dsp::Observation (Kernel/Classes/dsp/Observation.h) for this path."""
    centre_frequency: float = 1382.0   # MHz
    bandwidth: float = -400.0          # MHz, signed
    nchan: int = 1
    npol: int = 2
    ndim: int = 1                      # 1 => Nyquist (real), 2 => Analytic (complex)  ASCIIObservation.C:227-244
    nbit: int = 8
    tsamp_us: float = 0.00125
    dispersion_measure: float = 0.0
    machine: str = "DADA"
    source: str = "J0000+0000"
    telescope: str = "PKS"
    utc_start: str = "2010-04-13-02:05:45"
    obs_offset: int = 0
    dual_sideband: int = -1            # Observation.C:63,80-87
    dc_centred: bool = False
    swap: bool = False
    scale: float = 1.0
    start_seconds: float = 0.0         # seconds since utc_start of first sample

    @property
    def rate(self) -> float:           # samples per second  ASCIIObservation.C:298-377
        return 1e6 / self.tsamp_us

    @property
    def state(self) -> str:
        return "Nyquist" if self.ndim == 1 else "Analytic"

    def get_dual_sideband(self) -> bool:   # Observation.C:80-87
        if self.dual_sideband != -1:
            return self.dual_sideband == 1
        return self.state == "Analytic"


def parse_dada_header(raw: bytes) -> dict:
    """ascii_header_get semantics: 'KEY value' per line, '#' starts a comment."""
    out = {}
    text = raw.split(b"\0", 1)[0].decode("ascii", "replace")
    for line in text.splitlines():
        line = line.split("#", 1)[0].strip()
        if not line:
            continue
        parts = line.split(None, 1)
        if len(parts) == 2:
            out[parts[0]] = parts[1].strip()
    return out


def observation_from_header(h: dict) -> Observation:
    o = Observation()
    o.centre_frequency = float(h["FREQ"])
    o.bandwidth = float(h["BW"])
    o.nchan = int(h.get("NCHAN", 1))
    o.npol = int(h["NPOL"])
    o.ndim = int(h.get("NDIM", 1))
    o.nbit = int(h["NBIT"])
    o.tsamp_us = float(h["TSAMP"])
    o.machine = h.get("INSTRUMENT", "DADA")
    o.source = h.get("SOURCE", "")
    o.telescope = h.get("TELESCOPE", "")
    o.utc_start = h.get("UTC_START", o.utc_start)
    o.obs_offset = int(h.get("OBS_OFFSET", 0))
    if "DSB" in h:
        o.dual_sideband = 1 if int(h["DSB"]) == 1 else 0
    if "DM" in h:
        o.dispersion_measure = float(h["DM"])
    bytes_per_samp = o.nchan * o.npol * o.ndim * o.nbit // 8
    o.start_seconds = (o.obs_offset // bytes_per_samp) / o.rate
    return o


# --------------------------------------------------------------------------------------
# 8-bit unpack (a15)
#   BitTable::generate_unique_values  Kernel/Classes/BitTable.C:165-218
#   GenericEightBitUnpackerCUDA.cu:45,88-93:  float = (int8 + 0.5) * scale
#   byte order: BitUnpacker.C:48-80  (nskip = npol*nchan*ndim, offset (c*npol+p)*ndim+d)
#   CASPSR: 4 bytes pol0 then 4 bytes pol1  CASPSRUnpacker.C:132-187
# --------------------------------------------------------------------------------------

# JenetAnderson98::get_optimal_spacing(8) lives in PSRCHIVE (ext, unpinned).  Value of the
# published JA98 table for 8 bits; it only fixes one global scale constant S8.
JA98_SPACING_8BIT = 0.02957


def _normal_cdf(x: float) -> float:
    return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))


def eight_bit_scale(spacing: float = JA98_SPACING_8BIT) -> float:
    """BitTable::generate_unique_values scale for nbit=8 TwosComplement (BitTable.C:165-218)."""
    unique_values = 256
    output_spacing = 1.0 / unique_values
    output_middle = (unique_values - 1) / 2.0
    input_middle = unique_values // 2
    cumulative_probability = 0.0
    variance = 0.0
    for i in range(unique_values):
        output = (i - output_middle) * output_spacing
        if i < input_middle:
            threshold = float((i + 1) - input_middle) * spacing
            cumulative = _normal_cdf(threshold)
            interval = cumulative - cumulative_probability
            cumulative_probability = cumulative
            variance += output * output * interval
    variance *= 2.0
    scale = 1.0 / math.sqrt(variance)
    return scale * output_spacing


S8 = np.float32(eight_bit_scale())


def unpack_8bit(raw: np.ndarray, obs: Observation, scale=S8) -> np.ndarray:
    """raw int8 bytes -> float32 [nchan][npol][ndat*ndim] (FPT order).

    value = (float(int8) + 0.5f) * scale  (GenericEightBitUnpackerCUDA.cu:45)."""
    raw = np.asarray(raw).view(np.int8)
    nchan, npol, ndim = obs.nchan, obs.npol, obs.ndim
    if obs.machine == "CASPSR":
        assert nchan == 1 and npol == 2 and ndim == 1
        g = raw[: (raw.size // 8) * 8].reshape(-1, 2, 4)
        v = g.transpose(1, 0, 2).reshape(1, 2, -1)
    else:
        nskip = nchan * npol * ndim
        ndat = raw.size // nskip
        v = raw[: ndat * nskip].reshape(ndat, nchan, npol, ndim).transpose(1, 2, 0, 3)
        v = v.reshape(nchan, npol, ndat * ndim)
    v = np.ascontiguousarray(v)      # C order: callers hand the buffer to C / the device
    return ((v.astype(np.float32) + np.float32(0.5)) * np.float32(scale)).astype(np.float32)


# --------------------------------------------------------------------------------------
# optimal FFT length (a5)   Signal/General/optimize_fft.c:63-127
# --------------------------------------------------------------------------------------

def unpack_uwb16(raw: np.ndarray, npol: int = 2) -> np.ndarray:
    """dsp::UWBUnpacker::unpack (Kernel/Formats/uwb/UWBUnpacker.C:180-215; GPU twin UWBUnpackerCUDA.cu:24-75):
    16-bit offset-binary complex samples in blocks of 2048 per polarisation, value = float(int16(x ^ 0x8000)),
    no scale.  raw: uint16/int16 [nblock][npol][2048][2] flattened -> float32 [1][npol][ndat*2]."""
    w = np.ascontiguousarray(raw).view(np.uint16).reshape(-1, npol, 2048, 2)
    v = (w ^ np.uint16(0x8000)).view(np.int16).astype(np.float32)
    return np.ascontiguousarray(v.transpose(1, 0, 2, 3).reshape(1, npol, -1))


def optimal_fft_length(nbadperfft: int, nfft_max: int = 0) -> int:
    if not nbadperfft:
        return -1
    nfft_min = int(math.pow(2.0, math.ceil(math.log(nbadperfft) / math.log(2.0))))
    if nfft_max and nfft_max < nfft_min:
        return -1
    nfft = nfft_min
    # at nfft == nbadperfft the C code divides by zero -> +inf timescale
    def ts(n):
        d = float(n - nbadperfft)
        o = float(n) * math.log(n)
        return o / d if d != 0.0 else math.inf
    timescale = ts(nfft)
    while nfft_max == 0 or nfft * 2 < nfft_max:
        prev = timescale
        nfft *= 2
        timescale = ts(nfft)
        if timescale > prev:
            nfft //= 2
            break
    return nfft


# --------------------------------------------------------------------------------------
# Dedispersion response (a4, a5, a6)
# --------------------------------------------------------------------------------------

DM_DISPERSION = 2.41e-4          # Dedispersion.C:28
SMEARING_BUFFER = 0.1            # Dedispersion.C:30
SMEARING_SAMPLES_THRESHOLD = 16 * 1024 * 1024   # Dedispersion.C:214


class OracleError(Exception):
    """Stands in for the reference's `Error` exceptions (ext, PSRCHIVE)."""


@dataclass
class Dedispersion:
    """dsp::Dedispersion + the dsp::Response bookkeeping it inherits."""
    centre_frequency: float = 0.0
    bandwidth: float = 0.0
    dispersion_measure: float = 0.0
    nchan: int = 1
    ndat: int = 0                 # frequency resolution (freq_res)
    dc_centred: bool = False
    doppler_shift: float = 1.0
    frequency_resolution_set: bool = False
    ndat_max: int = 0
    impulse_pos: int = 0
    impulse_neg: int = 0
    whole_swapped: bool = False
    swap_divisions: int = 0
    buffer: np.ndarray | None = None     # complex64 [nchan*ndat]
    supported_channels: list = field(default_factory=list)

    # -- Dedispersion.C:383-430 -------------------------------------------------------
    def smearing_time(self, half: int) -> float:
        abs_bw = abs(self.bandwidth)
        ch_abs_bw = abs_bw / float(self.nchan)
        lower_ch_cfreq = self.centre_frequency - (abs_bw - ch_abs_bw) / 2.0
        for ok in self.supported_channels:
            if ok:
                break
            lower_ch_cfreq += ch_abs_bw
        if half:
            ch_abs_bw /= 2.0
            lower_ch_cfreq += float(half) * ch_abs_bw
        f1 = lower_ch_cfreq - abs(0.5 * ch_abs_bw)        # :343-346
        f2 = lower_ch_cfreq + abs(0.5 * ch_abs_bw)
        dispersion = self.dispersion_measure / DM_DISPERSION   # :354-355
        return dispersion * (1.0 / (f1 * f1) - 1.0 / (f2 * f2))

    # -- Dedispersion.C:432-475 -------------------------------------------------------
    def smearing_samples(self, half: int) -> int:
        tsmear = self.smearing_time(half)
        ch_abs_bw = abs(self.bandwidth) / float(self.nchan)
        sampling_rate = ch_abs_bw * 1e6
        tsmear *= (1.0 + SMEARING_BUFFER)
        return int(math.ceil(tsmear * sampling_rate))

    # -- Dedispersion.C:216-248 -------------------------------------------------------
    def prepare(self):
        threshold = SMEARING_SAMPLES_THRESHOLD // self.nchan
        self.supported_channels = [True] * self.nchan
        ichan = 0
        while True:
            self.impulse_neg = self.smearing_samples(-1)
            if self.impulse_neg <= threshold:
                break
            self.supported_channels[ichan] = False
            ichan += 1
            if ichan == self.nchan:
                raise OracleError("dsp::Dedispersion::prepare smearing samples=%u exceeds threshold=%u"
                                  % (self.impulse_neg, threshold))
        self.impulse_pos = self.smearing_samples(1)

    # -- Response.C:259-275 -----------------------------------------------------------
    def get_minimum_ndat(self) -> int:
        impulse_tot = float(self.impulse_pos + self.impulse_neg)
        if impulse_tot == 0:
            return 0
        m = int(math.pow(2.0, math.ceil(math.log(impulse_tot) / math.log(2.0))))
        while m <= impulse_tot:
            m *= 2
        return m

    # -- Response.C:328-344 -----------------------------------------------------------
    def check_ndat(self):
        if self.ndat_max and self.ndat > self.ndat_max:
            raise OracleError("Response::check_ndat specified maximum ndat (%d) < specified ndat (%d)"
                              % (self.ndat_max, self.ndat))
        ndat_min = self.get_minimum_ndat()
        if self.ndat < ndat_min:
            raise OracleError("dsp::Response::check_ndat specified ndat (%d) < required minimum ndat (%d)"
                              % (self.ndat, ndat_min))

    # -- Response.C:282-311 -----------------------------------------------------------
    def set_optimal_ndat(self):
        ndat_min = self.get_minimum_ndat()
        if self.ndat_max and self.ndat_max < ndat_min:
            raise OracleError("Response::set_optimal_ndat specified maximum ndat (%d) < required minimum ndat (%d)"
                              % (self.ndat_max, ndat_min))
        n = optimal_fft_length(self.impulse_pos + self.impulse_neg, self.ndat_max)
        if n < 0:
            raise OracleError("Response::set_optimal_ndat optimal_fft_length failed")
        self.ndat = n

    def set_frequency_resolution(self, nfft: int):      # Dedispersion.C:133-140
        self.ndat = nfft
        self.frequency_resolution_set = True

    # -- Dedispersion.C:478-556 : phases in double, stored as float -------------------
    def build_phases(self, ndat: int, nchan: int) -> np.ndarray:
        centrefreq = self.centre_frequency / self.doppler_shift
        bw = self.bandwidth / self.doppler_shift
        sign = bw / abs(bw)
        chanwidth = bw / float(nchan)
        binwidth = chanwidth / float(ndat)
        lower_cfreq = centrefreq - 0.5 * bw
        if not self.dc_centred:
            lower_cfreq += 0.5 * chanwidth
        dispersion_per_mhz = 1e6 * self.dispersion_measure / DM_DISPERSION
        highest_freq = centrefreq + 0.5 * abs(bw - chanwidth)            # :504
        samp_int = 1.0 / chanwidth                                       # :506
        phases = np.empty(ndat * nchan, dtype=np.float32)
        ipt = np.arange(ndat, dtype=np.float64)
        freq = ipt * binwidth - 0.5 * chanwidth
        for ichan in range(nchan):
            chan_cfreq = lower_cfreq + float(ichan) * chanwidth
            delay = 0.0
            if getattr(self, "fractional_delay", False):                  # -K  :524-533
                delay = dispersion_per_mhz * (1.0 / (chan_cfreq * chan_cfreq) - 1.0 / (highest_freq * highest_freq))
                delay = -math.fmod(delay, samp_int)
            coeff = -sign * 2 * math.pi * dispersion_per_mhz / (chan_cfreq * chan_cfreq)
            ph = coeff * (freq * freq) / (chan_cfreq + freq) + (-2.0 * math.pi * freq * delay)   # :543-545
            phases[ichan * ndat:(ichan + 1) * ndat] = ph.astype(np.float32)
        return phases

    # -- Dedispersion.C:291-331 -------------------------------------------------------
    def build(self):
        if self.frequency_resolution_set:
            self.check_ndat()
        else:
            self.set_optimal_ndat()
        phases = self.build_phases(self.ndat, self.nchan)
        # std::polar(float(1.0), phase): float cos/sin of the float phase
        ph64 = phases.astype(np.float64)
        buf = (np.cos(ph64).astype(np.float32) + 1j * np.sin(ph64).astype(np.float32)).astype(np.complex64)
        buf[0] = 0                                    # :323 always zap DC channel
        self.buffer = buf
        self.whole_swapped = False
        self.swap_divisions = 0

    # -- Response.C:649-700 -----------------------------------------------------------
    def doswap(self, divisions: int = 1):
        npts = self.ndat * self.nchan
        half = npts // (2 * divisions)
        b = self.buffer.reshape(divisions, 2, half)
        self.buffer = np.ascontiguousarray(b[:, ::-1, :]).reshape(-1)
        if divisions == 1:
            self.whole_swapped = not self.whole_swapped
        elif divisions == self.swap_divisions:
            self.swap_divisions = 0
        else:
            self.swap_divisions = divisions

    # -- Shape.C:222-266 : Shape[i] = Shape[i+rotbin] (complex elements) --------------
    def rotate(self, rotbin: int):
        self.buffer = np.roll(self.buffer, -rotbin)

    # -- Response.C:132-181 -----------------------------------------------------------
    def response_match(self, obs: Observation):
        if obs.nchan == 1:
            if obs.get_dual_sideband() and not self.whole_swapped:
                self.doswap()
        else:
            # Response::dc_centred is the member Dedispersion::prepare already set from the input
            # (Dedispersion.C:179, Response.h:170), so this branch only fires for other Responses
            if obs.dc_centred and not self.dc_centred:
                if self.swap_divisions:
                    self.doswap(self.swap_divisions)
                self.rotate(-int(self.ndat // 2))
                self.dc_centred = True
            if obs.get_dual_sideband() and self.swap_divisions != obs.nchan:
                self.doswap(obs.nchan)
            if obs.swap and not self.whole_swapped:
                self.doswap()

    # -- Dedispersion.C:167-197,261-283 -----------------------------------------------
    def match(self, obs: Observation, channels: int = 0):
        self.centre_frequency = obs.centre_frequency
        self.bandwidth = obs.bandwidth
        self.dispersion_measure = obs.dispersion_measure
        self.dc_centred = obs.dc_centred
        self.nchan = channels if channels else obs.nchan
        self.prepare()
        self.build()
        self.response_match(obs)
        self.buffer[0] = 0                           # :278 buffer[0] = buffer[1] = 0.0
        return self


# --------------------------------------------------------------------------------------
# Filterbank (a1, a2, a3)
# --------------------------------------------------------------------------------------

@dataclass
class FilterbankPlan:
    """Quantities derived in dsp::Filterbank::make_preparations (Filterbank.C:55-263)."""
    nchan: int
    input_nchan: int
    nchan_subband: int
    freq_res: int
    n_fft: int
    nfilt_pos: int
    nfilt_neg: int
    nfilt_tot: int
    nsamp_fft: int
    nsamp_overlap: int
    nsamp_step: int
    nkeep: int
    scalefac: float
    real_input: bool


def filterbank_plan(obs: Observation, nchan: int, response: Dedispersion | None,
                    freq_res: int = 1) -> FilterbankPlan:
    if nchan < obs.nchan:
        raise OracleError("dsp::Filterbank::make_preparations output nchan=%d < input nchan=%d" % (nchan, obs.nchan))
    if nchan % obs.nchan != 0:
        raise OracleError("dsp::Filterbank::make_preparations output nchan=%d not a multiple of input nchan=%d"
                          % (nchan, obs.nchan))
    nchan_subband = nchan // obs.nchan                       # :68
    nfilt_pos = nfilt_neg = 0
    if response is not None:
        nfilt_pos, nfilt_neg = response.impulse_pos, response.impulse_neg   # :90-91
        freq_res = response.ndat                                             # :93
        if freq_res == 0:
            raise OracleError("dsp::Filterbank::make_preparations Response.ndat = 0")
    n_fft = nchan_subband * freq_res                         # :107
    scalefac = float(n_fft) * float(freq_res)                # :124-125 (FTransform unnormalized)
    nfilt_tot = nfilt_pos + nfilt_neg                        # :131
    if obs.state == "Nyquist":                               # :139-148
        nsamp_fft = 2 * n_fft
        nsamp_overlap = 2 * nfilt_tot * nchan_subband
    elif obs.state == "Analytic":
        nsamp_fft = n_fft
        nsamp_overlap = nfilt_tot * nchan_subband
    else:
        raise OracleError("dsp::Filterbank::make_preparations invalid input data state")
    nsamp_step = nsamp_fft - nsamp_overlap                   # :155
    return FilterbankPlan(nchan, obs.nchan, nchan_subband, freq_res, n_fft, nfilt_pos, nfilt_neg, nfilt_tot,
                          nsamp_fft, nsamp_overlap, nsamp_step, freq_res - nfilt_tot, scalefac,
                          obs.state == "Nyquist")


def filterbank_npart(plan: FilterbankPlan, ndat: int) -> int:
    """Filterbank::resize_output (Filterbank.C:389-430)."""
    if plan.nsamp_step == 0:
        raise OracleError("dsp::Filterbank::resize_output nsamp_step == 0 ... not properly prepared")
    if ndat > plan.nsamp_overlap:
        return (ndat - plan.nsamp_overlap) // plan.nsamp_step
    return 0


def filterbank_output_observation(obs: Observation, plan: FilterbankPlan) -> Observation:
    """Filterbank::prepare_output metadata (Filterbank.C:265-379)."""
    out = Observation(**obs.__dict__)
    out.nchan = plan.nchan
    out.ndim = 2
    out.scale = obs.scale * plan.scalefac                    # :328 rescale
    ratechange = float(plan.freq_res) / float(plan.nsamp_fft)     # :338-339
    out.tsamp_us = 1e6 / (obs.rate * ratechange)
    if plan.freq_res == 1:
        out.dual_sideband = 1
    out.dc_centred = bool(plan.freq_res % 2)                 # :348
    if obs.get_dual_sideband():                              # :358-364
        if obs.nchan > 1:
            pass   # nsub_swap: bookkeeping only
        else:
            out.swap = True
    out.start_seconds = obs.start_seconds + plan.nfilt_pos / out.rate    # :370 change_start_time(nfilt_pos)
    return out


def filterbank(unpacked: np.ndarray, plan: FilterbankPlan, kernel: np.ndarray | None,
               npart: int | None = None, dtype=np.float32) -> np.ndarray:
    """dsp::Filterbank::filterbank CPU branch (Filterbank.C:561-662) + Response::operate (Response.C:385-444).

    unpacked: float [input_nchan][npol][ndat*ndim] -> complex [nchan][npol][npart*nkeep].
    dtype float32 follows the reference (FFTW single precision); float64 is the high-precision twin."""
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    input_nchan, npol, nfloat = unpacked.shape
    ndim = 1 if plan.real_input else 2
    ndat = nfloat // ndim
    if npart is None:
        npart = filterbank_npart(plan, ndat)
    N, M, C = plan.n_fft, plan.freq_res, plan.nchan_subband
    out = np.zeros((plan.nchan, npol, npart * plan.nkeep), dtype=cdt)
    in_step = plan.nsamp_step * ndim                         # :517
    for ichan in range(input_nchan):
        for ipart in range(npart):
            for ipol in range(npol):
                x = unpacked[ichan, ipol, ipart * in_step: ipart * in_step + plan.nsamp_fft * ndim].astype(dtype)
                if plan.real_input:
                    spec = np.fft.rfft(x)[:N]                 # frc1d: first N of the N+1 bins  :591
                else:
                    spec = np.fft.fft(x.view(cdt))            # fcc1d  :593
                spec = spec.astype(cdt)
                if kernel is not None:                        # Response::operate  :611-613
                    spec = (spec * kernel[ichan * N:(ichan + 1) * N].astype(cdt)).astype(cdt)
                if M == 1:                                    # :621-631
                    out[ichan * C:(ichan + 1) * C, ipol, ipart] = spec
                    continue
                # bcc1d unnormalised backward FFT per sub-channel  :640-652
                t = (np.fft.ifft(spec.reshape(C, M), axis=1) * M).astype(cdt)
                out[ichan * C:(ichan + 1) * C, ipol, ipart * plan.nkeep:(ipart + 1) * plan.nkeep] = \
                    t[:, plan.nfilt_pos: plan.nfilt_pos + plan.nkeep]
    return out


def convolution(unpacked: np.ndarray, response_ndat: int, nfilt_pos: int, nfilt_neg: int, kernel: np.ndarray,
                real_input: bool, npart: int | None = None, dtype=np.float32) -> np.ndarray:
    """dsp::Convolution::transformation, scalar (non-matrix) response, no apodization
    (Convolution.C:105-283 prepare, :338-461 transformation).

    unpacked: float [nchan][npol][ndat*ndim] -> complex [nchan][npol][npart*nsamp_good], where for every
    (chan, pol, part): forward FFT of nsamp_fft samples (frc1d for Nyquist input, n_fft = nsamp_fft/2 bins used;
    fcc1d for Analytic), response->operate (kernel of channel ichan, Response.C:385-444), unnormalised
    backward FFT of n_fft points, copy of nsamp_step*ndim floats from complex sample nfilt_pos (:441-446)."""
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    nchan, npol, nfloat = unpacked.shape
    ndim = 1 if real_input else 2
    n_fft = response_ndat                                    # :187
    nfilt_tot = nfilt_pos + nfilt_neg
    if real_input:                                           # :190-199
        nsamp_fft, nsamp_overlap = 2 * n_fft, 2 * nfilt_tot
    else:
        nsamp_fft, nsamp_overlap = n_fft, nfilt_tot
    nsamp_step = nsamp_fft - nsamp_overlap                   # :207
    ngood = nsamp_step * ndim // 2                           # complex samples written per part
    ndat = nfloat // ndim
    if npart is None:
        npart = (ndat - nsamp_overlap) // nsamp_step if ndat > nsamp_overlap else 0     # :304-306
    out = np.zeros((nchan, npol, npart * ngood), dtype=cdt)
    step = nsamp_step * ndim                                 # :386
    for ichan in range(nchan):
        for ipol in range(npol):
            for ipart in range(npart):
                x = unpacked[ichan, ipol, ipart * step: ipart * step + nsamp_fft * ndim].astype(dtype)
                spec = (np.fft.rfft(x)[:n_fft] if real_input else np.fft.fft(x.view(cdt))).astype(cdt)   # :412-416
                spec = (spec * kernel[ichan * n_fft:(ichan + 1) * n_fft].astype(cdt)).astype(cdt)        # :432
                t = (np.fft.ifft(spec) * n_fft).astype(cdt)                                               # :446
                out[ichan, ipol, ipart * ngood:(ipart + 1) * ngood] = t[nfilt_pos: nfilt_pos + ngood]    # :451
    return out


# --------------------------------------------------------------------------------------
# Detection (a7, a8)   cross_detect.ic:23-43, stokes_detect.ic:21-44, Detection.C:218-474
# --------------------------------------------------------------------------------------

def detect_products(fb: np.ndarray, state: str = "Coherence") -> np.ndarray:
    """complex [nchan][2][ndat] -> float [nchan][4][ndat] (product-major, layout-free)."""
    p, q = fb[:, 0, :], fb[:, 1, :]
    f = np.float32 if fb.dtype == np.complex64 else np.float64
    pr, pi, qr, qi = p.real.astype(f), p.imag.astype(f), q.real.astype(f), q.imag.astype(f)
    pp = pr * pr + pi * pi
    qq = qr * qr + qi * qi
    re = pr * qr + pi * qi
    im = pr * qi - pi * qr
    if state == "Coherence":
        prod = [pp, qq, re, im]
    elif state == "Stokes":
        two = f(2.0)
        prod = [pp + qq, pp - qq, two * re, two * im]
    else:
        raise OracleError("dsp::Detection invalid state " + state)
    return np.stack(prod, axis=1)


def detect_layout(prod: np.ndarray, ndim: int) -> np.ndarray:
    """Arrange [nchan][4][ndat] products as the TimeSeries Detection writes (Detection.C:423-474):
       ndim=1 -> [nchan][npol=4][ndat]; ndim=2 -> [nchan][npol=2][ndat][2]; ndim=4 -> [nchan][npol=1][ndat][4]."""
    nchan, _, ndat = prod.shape
    if ndim == 1:
        return prod.copy()
    if ndim == 2:
        return np.stack([np.stack([prod[:, 0], prod[:, 1]], axis=-1),
                         np.stack([prod[:, 2], prod[:, 3]], axis=-1)], axis=1)
    if ndim == 4:
        return np.stack([prod[:, 0], prod[:, 1], prod[:, 2], prod[:, 3]], axis=-1)[:, None]
    raise OracleError("dsp::Detection::get_result_pointers invalid ndim=%d" % ndim)


def square_law(fb: np.ndarray, state: str = "PPQQ") -> np.ndarray:
    """Detection::square_law for Analytic input (Detection.C:218-320): [nchan][npol][ndat] complex ->
    PPQQ: [nchan][2][ndat]; Intensity: [nchan][1][ndat] (pol sum)."""
    f = np.float32 if fb.dtype == np.complex64 else np.float64
    re, im = fb.real.astype(f), fb.imag.astype(f)
    out = re * re
    out = out + im * im
    if state == "Intensity" and fb.shape[1] == 2:
        return (out[:, 0:1] + out[:, 1:2])
    return out


# --------------------------------------------------------------------------------------
# Search mode front end: TFPFilterbank + pscrunch + TScrunch
# --------------------------------------------------------------------------------------

def tfp_filterbank(unpacked: np.ndarray, nchan: int, pscrunch: bool, dtype=np.float32) -> np.ndarray:
    """dsp::TFPFilterbank::filterbank (TFPFilterbank.C:27-101), real input: per pol and part forward FFT of
    nsamp_fft = 2*nchan samples, Re^2 then += Im^2 of bins 0..nchan-1.  -> [npart][nchan][npol_out]."""
    _, npol, ndat = unpacked.shape
    nsamp_fft = 2 * nchan
    npart = ndat // nsamp_fft
    x = unpacked[0, :, :npart * nsamp_fft].astype(dtype).reshape(npol, npart, nsamp_fft)
    spec = np.fft.rfft(x, axis=2)[:, :, :nchan]
    f = np.float32 if dtype == np.float32 else np.float64
    re, im = spec.real.astype(f), spec.imag.astype(f)
    out = re * re
    out = out + im * im                                   # [npol][npart][nchan]
    if npol == 2 and pscrunch:
        return (out[0] + out[1])[:, :, None]              # :79-80  outdat[i] += outdat[i+nfloat]
    return np.ascontiguousarray(out.transpose(1, 2, 0))


def tscrunch_tfp(x: np.ndarray, sfactor: int) -> np.ndarray:
    """dsp::TScrunch::tfp_tscrunch (TScrunch.C:180-206): out = in[0]; out += in[1]; ... (sequential)."""
    nout = x.shape[0] // sfactor
    out = np.empty((nout,) + x.shape[1:], dtype=x.dtype)
    for o in range(nout):
        acc = x[o * sfactor].copy()
        for j in range(1, sfactor):
            acc = acc + x[o * sfactor + j]
        out[o] = acc
    return out


def tscrunch_fpt(x: np.ndarray, sfactor: int) -> np.ndarray:
    """dsp::TScrunch::fpt_tscrunch (Signal/General/TScrunch.C:148-178), ndim 1: x [nchan][npol][ndat] ->
    [nchan][npol][ndat // sfactor]; out = in[0]; out += in[1]; ... sequentially in float.  (The ndat % sfactor samples left over
    stay with the caller: TScrunch's input buffering re-presents them in front of the next block, :110-111.)"""
    x = np.asarray(x, np.float32)
    nout = x.shape[2] // sfactor
    acc = x[:, :, 0:nout * sfactor:sfactor].copy()
    for j in range(1, sfactor):
        acc = acc + x[:, :, j:nout * sfactor:sfactor]
    return acc


def fscrunch_fpt(x: np.ndarray, sfactor: int) -> np.ndarray:
    """dsp::FScrunch::fpt_fscrunch (Signal/General/FScrunch.C:117-145): x [nchan][npol][nfloat] -> [nchan // sfactor][npol][nfloat];
    out row = in row c*sfactor, += rows c*sfactor + 1 ... in order."""
    x = np.asarray(x, np.float32)
    nout = x.shape[0] // sfactor
    acc = x[0:nout * sfactor:sfactor].copy()
    for j in range(1, sfactor):
        acc = acc + x[j:nout * sfactor:sfactor]
    return acc


def sigproc_digitize_fpt(x: np.ndarray, nbit: int, **kw) -> np.ndarray:
    """dsp::SigProcDigitizer::pack, FPT branch (SigProcDigitizer.C:238-290; pack_float :346-358): x [nchan][npol][ndat].  The same
    expression per sample as the TFP branch and the same TPF byte order (outidx = idat*nchan*npol + ipol*nchan + ichan)."""
    return sigproc_digitize(np.ascontiguousarray(np.asarray(x, np.float32).transpose(2, 0, 1)), nbit, **kw)


class DigifilCoherent:
    """digifil with a convolving filterbank, `digifil -F N:D [-x M] [-K] -t T -b nbit` (Signal/General/LoadToFil.C:185-222,234-362),
    on the OUTPUT of the filterbank: [SampleDelay, -K, :236-247: in place on the complex rows, the unshifted tail buffered for the
    next block] -> Detection (:250-279: Intensity, npol 1; PPQQ, npol 2; Coherence with ndim 1, npol 4) -> [FScrunch] -> TScrunch
    (FPT, left-over samples buffered) -> Rescale (FPT: the same statistics per (chan, pol) as TFP) -> SigProcDigitizer (FPT branch).
    Feed it the complex filterbank rows [nchan][npol][ndat] block by block; it returns the packed bytes of each block."""

    def __init__(self, tscrunch: int = 1, fscrunch: int = 0, nbit: int = 8, npol_out: int = 1, rescale_interval: int = 0,
                 rescale_constant: bool = False, rescale: bool = True, scale_fac: float = 1.0, flip_band: bool = False,
                 delays=None, input_scale: float = 1.0):
        self.tscrunch, self.fscrunch, self.nbit, self.npol_out = tscrunch, fscrunch, nbit, npol_out
        self.scale_fac, self.flip_band = scale_fac, flip_band
        self.rescale = Rescale(rescale_interval, rescale_constant) if rescale else None
        self.left = None                                          # TScrunch input buffering: samples not yet scrunched
        self.input_scale = input_scale                            # (Filterbank scale x scrunch factors; Rescale resets it to 1, Rescale.C:204)
        self.delays = None if delays is None else np.asarray(delays, np.int64)
        self.sd_left = None                                       # SampleDelay input buffering: the last total_delay samples

    def detect_scrunch(self, fb: np.ndarray) -> np.ndarray:
        if self.delays is not None:                                                     # LoadToFil.C:236-247
            if self.sd_left is not None:
                fb = np.concatenate([self.sd_left, fb], axis=2)
            out, _zero, total = sample_delay(fb, self.delays)
            self.sd_left = fb[:, :, out.shape[2]:].copy() if total else None            # InputBuffering: SampleDelay.C:117,146
            fb = out
        if self.npol_out == 4:                                                          # :273-277 (ndim stays 1)
            det = np.ascontiguousarray(detect_layout(detect_products(fb, "Coherence"), 1))
        else:
            det = square_law(fb, "Intensity" if self.npol_out == 1 else "PPQQ")       # :262-269
        if self.fscrunch:                                                               # :286-294
            det = fscrunch_fpt(det, self.fscrunch)
        if self.tscrunch and self.tscrunch > 1:                                         # :296-304
            if self.left is not None:
                det = np.concatenate([self.left, det], axis=2)
            nout = det.shape[2] // self.tscrunch
            self.left = det[:, :, nout * self.tscrunch:].copy()
            det = tscrunch_fpt(det, self.tscrunch)
        return det

    def process(self, fb: np.ndarray) -> np.ndarray:
        det = self.detect_scrunch(fb)
        tfp = np.ascontiguousarray(det.transpose(2, 0, 1))                             # same per-(chan, pol) arithmetic in either order
        if self.rescale is not None:                                                    # :306-316
            tfp = self.rescale.transform(tfp)
        return sigproc_digitize(tfp, self.nbit, use_digi_scales=self.rescale is not None,
                                input_scale=1.0 if self.rescale is not None else self.input_scale, scale_fac=self.scale_fac,
                                flip_band=self.flip_band)


# --------------------------------------------------------------------------------------
# Integer-sample inter-channel delay, -K (f-4)
# --------------------------------------------------------------------------------------

def observation_channel_frequency(obs: "Observation", ichan: int, nchan: int | None = None, swap: bool = False,
                                  nsub_swap: int = 0) -> float:
    """Observation::get_centre_frequency(ichan) (Kernel/Classes/Observation.C:420-451)."""
    nchan = nchan or obs.nchan
    c = ichan
    if swap:
        c = (c + nchan // 2) % nchan
    if nsub_swap:
        sub = nchan // nsub_swap
        c = (c // sub) * sub + (c % sub + sub // 2) % sub
    base = obs.centre_frequency - 0.5 * obs.bandwidth
    if not obs.dc_centred:
        base += 0.5 * obs.bandwidth / float(nchan)
    return base + float(c) * obs.bandwidth / float(nchan)


def dedispersion_sample_delays(obs: "Observation", nchan: int, rate_hz: float, swap: bool = False, nsub_swap: int = 0):
    """Dedispersion::SampleDelay::match (Signal/General/DedispersionSampleDelay.C:24-75)."""
    if rate_hz == 0 or obs.bandwidth == 0 or obs.centre_frequency == 0:
        raise OracleError("dsp::Dedispersion::SampleDelay::match invalid input")
    dispersion = obs.dispersion_measure / DM_DISPERSION
    out = np.zeros(nchan, np.int64)
    for ichan in range(nchan):
        freq = observation_channel_frequency(obs, ichan, nchan, swap, nsub_swap)
        delay = dispersion * (1.0 / (obs.centre_frequency * obs.centre_frequency) - 1.0 / (freq * freq))
        out[ichan] = int(math.floor(delay * rate_hz + 0.5))
    return out


def sample_delay(x: np.ndarray, delays: np.ndarray, absolute: bool = False):
    """dsp::SampleDelay::build + transformation (Signal/General/SampleDelay.C:52-195).

    x: [nchan][npol][ndat](...) ; delays: [nchan] or [nchan][npol] as SampleDelayFunction::get_delay returns.
    Returns (output [nchan][npol][ndat - total_delay](...), zero_delay, total_delay)."""
    nchan, npol, ndat = x.shape[:3]
    d = np.asarray(delays, np.int64)
    if d.ndim == 1:
        d = np.repeat(d[:, None], npol, axis=1)
    if absolute:                                            # :60-73
        zero_delay, total_delay = 0, int(d.max())
        applied = d
    else:                                                   # :75-99
        zero_delay = int(d.max())
        applied = zero_delay - d if zero_delay else d       # :166-172
        total_delay = int((zero_delay - d).max())
    assert applied.min() >= 0                               # :174
    nout = max(0, ndat - total_delay)                       # :137-145
    out = np.empty((nchan, npol, nout) + x.shape[3:], x.dtype)
    for c in range(nchan):
        for p in range(npol):
            a = int(applied[c, p])
            out[c, p] = x[c, p, a:a + nout]
    return out, zero_delay, total_delay


# --------------------------------------------------------------------------------------
# Search-mode output stage (f-1): Rescale + SigProcDigitizer
# --------------------------------------------------------------------------------------


class Rescale:
    """dsp::Rescale on TFP-ordered detected data (Signal/General/Rescale.C:157-420), `exact`/decay off.

    Per (pol, chan) running sums of x and x*x (the square is taken in float, the sums are doubles: :243-244) over
    intervals of `interval_samples` samples (0: the length of the first block, :100-105); at the end of an interval,
    and right after the first block segment, compute_various (:390-420) sets offset = -mean and
    scale = 1/sqrt(variance) (1 if the variance is 0) unless `constant` froze the first estimate; each sample
    leaves as (x + offset) * scale in float (:352).  Arrays are indexed [chan][pol] like the TFP data."""

    def __init__(self, interval_samples: int = 0, constant: bool = False):
        self.interval_samples, self.constant = interval_samples, constant
        self.nsample = 0
        self.isample = 0
        self.total = self.totalsq = self.offset = self.scale = None

    def transform(self, x: np.ndarray) -> np.ndarray:
        """x: float32 [ndat][nchan][npol]; returns the rescaled block."""
        x = np.asarray(x, np.float32)
        ndat = x.shape[0]
        out = np.empty_like(x)
        first_call = self.nsample == 0                                   # :174
        if first_call:                                                    # init, :94-130
            self.nsample = self.interval_samples or ndat
            if not self.nsample:
                raise OracleError("dsp::Rescale::init nsample == 0")
            self.isample = 0
            self.total = np.zeros(x.shape[1:], np.float64)
            self.totalsq = np.zeros(x.shape[1:], np.float64)
            self.offset = np.zeros(x.shape[1:], np.float32)
            self.scale = np.ones(x.shape[1:], np.float32)
        if not ndat:
            return out
        start = 0
        while True:                                                       # :217-380
            end = min(ndat, start + self.nsample - self.isample)
            seg = x[start:end]
            # sequential double accumulation, sample by sample (:236-247); cumsum adds in that order
            self.total = (self.total[None] + np.cumsum(seg.astype(np.float64), axis=0))[-1] if len(seg) else self.total
            sq = (seg * seg).astype(np.float64)                           # float product, then widened
            self.totalsq = (self.totalsq[None] + np.cumsum(sq, axis=0))[-1] if len(seg) else self.totalsq
            self.isample += end - start
            if self.isample == self.nsample or first_call:               # :298-326
                mean = self.total / self.isample
                variance = self.totalsq / self.isample - mean * mean
                if not self.constant or first_call:
                    self.offset = (-mean).astype(np.float32)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        self.scale = np.where(variance == 0.0, 1.0, 1.0 / np.sqrt(variance)).astype(np.float32)
                self.isample = 0
                first_call = False
                self.total = np.zeros_like(self.total)
                self.totalsq = np.zeros_like(self.totalsq)
            out[start:end] = (seg + self.offset[None]) * self.scale[None]   # float32 add, float32 multiply (:352)
            start = end
            if end >= ndat:
                break
        return out


def pscrunch_tfp(x: np.ndarray) -> np.ndarray:
    """dsp::PScrunch::transformation, TFP branch (Signal/General/PScrunch.C:36-90): (p0 + p1) * float(1/sqrt(2))."""
    x = np.asarray(x, np.float32)
    if x.shape[2] == 1:
        raise OracleError("dsp::PScrunch::transformation invalid npol=1")
    scale = np.float32(1.0 / math.sqrt(2.0))
    return ((x[:, :, 0] + x[:, :, 1]) * scale)[:, :, None]


def channel_sort(nchan: int, flip_band: bool, swap_band: bool) -> np.ndarray:
    """ChannelSort (Kernel/Formats/sigproc/SigProcDigitizer.C:38-66, nsub_swap <= 1): input channel of each output
    channel; flip_band = input bandwidth > 0, swap_band = input->get_swap()."""
    k = np.arange(nchan)
    if swap_band:
        k = (k + nchan // 2) % nchan
    if flip_band:
        k = nchan - k - 1
    return k


def sigproc_digitize(x: np.ndarray, nbit: int, use_digi_scales: bool = True, input_scale: float = 1.0,
                     scale_fac: float = 1.0, flip_band: bool = False, swap_band: bool = False) -> np.ndarray:
    """dsp::SigProcDigitizer::pack, TFP branch (SigProcDigitizer.C:80-246; pack_float :309-342).

    x: float32 [ndat][nchan][npol].  Returns the raw output bytes (uint8; nbit 16: uint16; nbit -32: float32) in
    [ndat][npol][nchan] order, sub-byte samples packed LSB first."""
    x = np.asarray(x, np.float32)
    ndat, nchan, npol = x.shape
    sel = x[:, channel_sort(nchan, flip_band, swap_band), :].transpose(0, 2, 1)      # [ndat][npol][out chan]
    if nbit == -32:
        return (sel / np.float32(input_scale)).astype(np.float32)
    table = {1: (0.5, 1.0, 1), 2: (1.5, 1.0, 3), 4: (7.5, None, 15), 8: (127.5, None, 255), 16: (32768.0, None, 65535)}
    if nbit not in table:
        raise OracleError("dsp::SigProcDigitizer::set_nbit nbit=%d not understood" % nbit)
    digi_mean, digi_scale, digi_max = table[nbit]
    digi_mean = np.float32(digi_mean)
    digi_scale = np.float32(digi_scale) if digi_scale is not None else digi_mean / np.float32(6)    # :112-143
    xpol_offset = np.float32(0)
    if not use_digi_scales:                                                                           # :148-154
        xpol_offset, digi_mean, digi_scale = digi_mean, np.float32(0), np.float32(1)
    digi_scale = np.float32(np.float64(digi_scale) / (np.float64(input_scale) * np.float64(np.float32(scale_fac))))   # :158
    mean = np.full(npol, digi_mean, np.float32)
    mean[2:] += xpol_offset                                                                           # :176-177
    with np.errstate(invalid="ignore", over="ignore"):
        y = (sel * digi_scale + mean[None, :, None]).astype(np.float64) + 0.5                         # :198
        bad = ~((y < 2147483648.0) & (y > -2147483649.0))                  # x86 cvttsd2si: indefinite integer
        r = np.where(bad, -2147483648, np.trunc(np.where(bad, 0, y))).astype(np.int64)
    r = np.clip(r, 0, digi_max)
    if nbit == 16:
        return r.astype(np.uint16)
    if nbit == 8:
        return r.astype(np.uint8)
    spb = 8 // nbit
    r = r.reshape(ndat, npol, nchan // spb, spb)
    return (r << (np.arange(spb) * nbit)).sum(axis=-1).astype(np.uint8)

# --------------------------------------------------------------------------------------
# Fold (a9 - a13)
# --------------------------------------------------------------------------------------


@dataclass
class Polyco:
    """TEMPO polyco block (ext: PSRCHIVE Pulsar::Predictor); restated from the TEMPO definition:
       phase(t) = RPHASE + 60 DT F0 + sum_k c_k DT^k ; freq(t) = F0 + (1/60) sum_k k c_k DT^(k-1),
       DT = (t - TMID) * 1440 minutes.  File layout: Benchmark/vela.polyco:1-7."""
    tmid_day: int
    tmid_frac: float
    rphase_int: int
    rphase_frac: float
    f0: float
    span_min: float
    coef: np.ndarray

    @staticmethod
    def parse(text: str) -> "Polyco":
        tok = text.replace("D", "E").split()
        # line1: name date utc tmid dm doppler log10rms ; line2: rphase f0 site span ncoef freq
        tmid = tok[3]
        day, frac = tmid.split(".")
        rph = tok[7]
        ri, rf = rph.split(".")
        ncoef = int(tok[11])
        coef = np.array([float(t) for t in tok[13:13 + ncoef]], dtype=np.float64)
        return Polyco(int(day), float("0." + frac), int(ri), float("0." + rf), float(tok[8]), float(tok[10]), coef)

    def _dt_min(self, mjd_day: int, mjd_sec: float) -> float:
        return ((mjd_day - self.tmid_day) + (mjd_sec / 86400.0 - self.tmid_frac)) * 1440.0

    def phase_frac(self, mjd_day: int, mjd_sec: float) -> float:
        dt = self._dt_min(mjd_day, mjd_sec)
        poly = 0.0
        for c in self.coef[::-1]:
            poly = poly * dt + c
        spin = 60.0 * dt * self.f0
        ph = (self.rphase_frac + (spin - math.floor(spin))) + poly
        return ph - math.floor(ph)

    def frequency(self, mjd_day: int, mjd_sec: float) -> float:
        dt = self._dt_min(mjd_day, mjd_sec)
        d = 0.0
        for k in range(len(self.coef) - 1, 0, -1):
            d = d * dt + k * self.coef[k]
        return self.f0 + d / 60.0


class ChebyPredictor:
    """TEMPO2 predictor text (ChebyModelSet), restated from the published definition of tempo2's T2Predictor -- ext, absent
    from /root/reference (Fold.C:229-262 only asks Pulsar::Generator for a predictor): phase(t, f) = primed double Chebyshev
    sum over TIME_RANGE x FREQ_RANGE + DISPERSION_CONSTANT / f^2, frequency = d phase / dt.  Written independently of the
    product's pipeline.ChebyPredictor (numpy's chebval2d / chebder here, explicit recurrences there).  "parity unpinned"."""

    def __init__(self, text: str, observing_frequency: float | None = None):
        from fractions import Fraction
        self.seg = []
        cur = None
        for line in text.splitlines():
            t = line.split()
            if not t:
                continue
            if t[:2] == ["ChebyModel", "BEGIN"]:
                cur = {"v": []}
            elif t[:2] == ["ChebyModel", "END"]:
                nx, ny = cur["nx"], cur["ny"]
                c = np.array([float(v) for v in cur["v"]], np.float64).reshape(nx, ny)
                const = Fraction(cur["v"][0]) / 4
                cur["ci"] = const.numerator // const.denominator
                cur["cf"] = float(const - cur["ci"])
                c[0, :] *= 0.5
                c[:, 0] *= 0.5
                c[0, 0] = 0.0
                cur["c"] = c
                self.seg.append(cur)
                cur = None
            elif cur is not None:
                if t[0] == "TIME_RANGE":
                    cur["t"] = [(int(v.partition(".")[0]), float("0." + (v.partition(".")[2] or "0"))) for v in t[1:3]]
                elif t[0] == "FREQ_RANGE":
                    cur["f"] = (float(t[1]), float(t[2]))
                elif t[0] == "DISPERSION_CONSTANT":
                    cur["dc"] = float(t[1])
                elif t[0] == "NCOEFF_TIME":
                    cur["nx"] = int(t[1])
                elif t[0] == "NCOEFF_FREQ":
                    cur["ny"] = int(t[1])
                elif t[0] == "COEFFS":
                    cur["v"] += t[1:]
                else:
                    try:
                        float(t[0])
                        cur["v"] += t
                    except ValueError:
                        pass
        self.observing_frequency = observing_frequency if observing_frequency is not None else 0.5 * sum(self.seg[0]["f"])

    def _xy(self, day, sec):
        for s in self.seg:
            a = (day - s["t"][0][0]) + (sec / 86400.0 - s["t"][0][1])
            b = (s["t"][1][0] - day) + (s["t"][1][1] - sec / 86400.0)
            if a >= 0 and b >= 0:
                y = -1.0 + 2.0 * (self.observing_frequency - s["f"][0]) / (s["f"][1] - s["f"][0])
                return s, -1.0 + 2.0 * a / (a + b), y, a + b
        raise ValueError("ChebyPredictor: epoch outside every TIME_RANGE")

    def phase(self, day, sec):
        s, x, y, _ = self._xy(day, sec)
        v = float(np.polynomial.chebyshev.chebval2d(x, y, s["c"])) + s.get("dc", 0.0) / self.observing_frequency ** 2
        fr = s["cf"] + v
        fi = math.floor(fr)
        return int(s["ci"] + fi), fr - fi

    def phase_frac(self, day, sec):
        return self.phase(day, sec)[1]

    def frequency(self, day, sec):
        s, x, y, span = self._xy(day, sec)
        d = np.polynomial.chebyshev.chebder(s["c"], axis=0)
        return float(np.polynomial.chebyshev.chebval2d(x, y, d)) * 2.0 / (span * 86400.0)


def utc_to_mjd(utc: str) -> tuple[int, float]:
    """YYYY-MM-DD-hh:mm:ss -> (integer MJD, seconds of day).  (MJD class is ext.)"""
    y, mo, d, hms = utc.split("-")
    h, mi, s = hms.split(":")
    y, mo, d = int(y), int(mo), int(d)
    a = (14 - mo) // 12
    yy = y + 4800 - a
    mm = mo + 12 * a - 3
    jdn = d + (153 * mm + 2) // 5 + 365 * yy + yy // 4 - yy // 100 + yy // 400 - 32045
    return jdn - 2400001, int(h) * 3600.0 + int(mi) * 60.0 + float(s)


@dataclass
class PhaseSeries:
    """dsp::PhaseSeries (Signal/Pulsar/dsp/PhaseSeries.h:163-200): profile sums + hits."""
    nchan: int
    npol: int
    ndim: int
    nbin: int
    data: np.ndarray = None          # [nchan][npol][nbin][ndim]
    hits: np.ndarray = None          # [nbin] uint32
    integration_length: float = 0.0
    ndat_total: int = 0
    folding_period: float = 0.0

    def __post_init__(self):
        if self.data is None:
            self.data = np.zeros((self.nchan, self.npol, self.nbin, self.ndim), dtype=np.float32)
        if self.hits is None:
            self.hits = np.zeros(self.nbin, dtype=np.uint32)

    def zero(self):
        self.data[...] = 0
        self.hits[...] = 0
        self.integration_length = 0.0
        self.ndat_total = 0

    def combine(self, other: "PhaseSeries"):          # PhaseSeries.C:442-484
        self.data += other.data
        self.hits += other.hits
        self.integration_length += other.integration_length
        self.ndat_total += other.ndat_total


@dataclass
class FoldConfig:
    nbin: int
    folding_period: float = 0.0          # seconds; >0 => constant period (Fold.C:945-947)
    polyco: Polyco | None = None
    reference_phase: float = 0.0         # Fold.C:75-76
    reference_epoch_seconds: float = 0.0  # seconds relative to utc_start (MJD zero is ext)


def fold_phase(cfg: FoldConfig, obs: Observation, t_seconds: float) -> tuple[float, float]:
    """get_phi / get_pfold (Fold.C:943-958).  t_seconds is relative to obs.utc_start."""
    if cfg.folding_period > 0.0:
        phi = math.fmod(t_seconds - cfg.reference_epoch_seconds, cfg.folding_period) / cfg.folding_period \
            - cfg.reference_phase
        return phi, cfg.folding_period
    day, sec = utc_to_mjd(obs.utc_start)
    sec += t_seconds
    phi = cfg.polyco.phase_frac(day, sec) - cfg.reference_phase
    return phi, 1.0 / cfg.polyco.frequency(day, sec)


def fold_binplan(phi: float, phase_per_sample: float, nbin: int, ndat_fold: int) -> np.ndarray:
    """The sequential double recurrence of Fold.C:744-787 (no weights, no zeroed samples)."""
    plan = np.empty(ndat_fold, dtype=np.uint32)
    double_nbin = float(nbin)
    for i in range(ndat_fold):
        phi -= math.floor(phi)
        ibin = int(phi * double_nbin)
        phi += phase_per_sample
        assert ibin < nbin
        plan[i] = ibin
    return plan


def fold_binplan_weighted(phi: float, phase_per_sample: float, nbin: int, idat_start: int, ndat_fold: int,
                          weights: np.ndarray, ndatperweight: int, weight_idat: int = 0) -> np.ndarray:
    """Fold.C:686-716,744-787 with a WeightedTimeSeries input (not zeroed_samples): sample idat belongs to weight
    (idat + weight_idat) / ndatperweight; samples of a zero weight get binplan = nbin (not folded, no hit)."""
    plan = np.empty(ndat_fold, dtype=np.uint32)
    double_nbin = float(nbin)
    iweight = (idat_start + weight_idat) // ndatperweight
    idat_nextweight = (iweight + 1) * ndatperweight - weight_idat
    assert iweight < len(weights)
    bad = weights[iweight] == 0
    for idat in range(idat_start, idat_start + ndat_fold):
        if idat >= idat_nextweight:
            iweight += 1
            assert iweight < len(weights)
            bad = weights[iweight] == 0
            idat_nextweight += ndatperweight
        phi -= math.floor(phi)
        ibin = int(phi * double_nbin)
        phi += phase_per_sample
        assert ibin < nbin
        plan[idat - idat_start] = nbin if bad else ibin
    return plan


def fold(detected: np.ndarray, obs: Observation, cfg: FoldConfig, out: PhaseSeries,
         idat_start: int = 0, ndat_fold: int | None = None) -> np.ndarray:
    """dsp::Fold::fold (Fold.C:626-906), FPT order.  detected: [nchan][npol][ndat][ndim] float.
    Accumulates strictly in time order per (chan,pol,bin,dim) like the CPU loop :835-891."""
    nchan, npol, ndat, ndim = detected.shape
    if ndat_fold is None:
        ndat_fold = ndat - idat_start
    if idat_start + ndat_fold > ndat:
        raise OracleError("dsp::Fold:fold idat_start + ndat_fold > ndat")
    mid = float(idat_start) + 0.5                             # :651
    t0 = obs.start_seconds + mid / obs.rate                   # :653-654
    phi, pfold = fold_phase(cfg, obs, t0)
    out.folding_period = pfold
    sampling_interval = 1.0 / obs.rate                        # :718
    phase_per_sample = sampling_interval / pfold              # :720
    plan = fold_binplan(phi, phase_per_sample, cfg.nbin, ndat_fold)
    out.hits += np.bincount(plan, minlength=cfg.nbin).astype(np.uint32)      # :783
    out.integration_length += float(ndat_fold) / obs.rate     # :792,802
    out.ndat_total += ndat_fold                               # :803
    seg = detected[:, :, idat_start:idat_start + ndat_fold, :]
    # sequential-in-time accumulation, vectorised over (chan,pol,dim): identical add order per bin
    order = np.argsort(plan, kind="stable")
    sp = plan[order]
    bounds = np.flatnonzero(np.diff(sp)) + 1
    starts = np.concatenate(([0], bounds))
    ends = np.concatenate((bounds, [sp.size]))
    acc_dtype = out.data.dtype
    for s, e in zip(starts, ends):
        b = int(sp[s])
        acc = out.data[:, :, b, :].copy()
        for i in order[s:e]:
            acc = (acc + seg[:, :, i, :].astype(acc_dtype)).astype(acc_dtype)
        out.data[:, :, b, :] = acc
    return plan


def choose_nbin(folding_period: float, rate: float, requested_nbin: int = 0, maximum_nbin: int = 1024,
                minimum_bin_width: float = 1.2, power_of_two: bool = True, force_sensible_nbin: bool = False) -> int:
    """dsp::Fold::choose_nbin (Fold.C:291-382)."""
    if folding_period <= 0.0:
        raise OracleError("dsp::Fold::choose_nbin invalid folding period=%f" % folding_period)
    sampling_period = 1.0 / rate
    binwidth = minimum_bin_width * sampling_period
    sensible = int(folding_period / binwidth)
    if power_of_two:
        log2bin = math.log(folding_period / binwidth) / math.log(2.0)
        sensible = int(math.pow(2.0, math.floor(log2bin)))
    if sensible == 0:
        sensible = 1
    if requested_nbin > 1:
        nbin = requested_nbin
        if requested_nbin > sensible and force_sensible_nbin:
            nbin = sensible
        return nbin
    if maximum_nbin and sensible > maximum_nbin:
        return maximum_nbin
    return sensible


def archive_profile(ps: PhaseSeries, scale: float) -> np.ndarray:
    """dsp::Archiver::set normalisation (Archiver.C:773-893): amp = sum / (scale * hits);
    zero-hit bins take the mean of the others."""
    hits = ps.hits.astype(np.float64)
    amps = np.zeros_like(ps.data, dtype=np.float64)
    ok = hits > 0
    amps[:, :, ok, :] = ps.data[:, :, ok, :].astype(np.float64) / (scale * hits[ok])[None, None, :, None]
    if (~ok).any() and ok.any():
        amps[:, :, ~ok, :] = amps[:, :, ok, :].mean(axis=2, keepdims=True)
    return amps.astype(np.float32)


# --------------------------------------------------------------------------------------
# Sub-integration division in seconds (a12)   TimeDivide.C:440-459,503-540
# --------------------------------------------------------------------------------------

def subint_boundaries(obs: Observation, division_seconds: float, t_seconds: float) -> tuple[int, float, int]:
    """-> (division index, lower bound in seconds snapped to samples, division_ndat).
    Division start_time is the observation start (first call, TimeDivide.C:380-436)."""
    start = obs.start_seconds
    seconds = max(t_seconds, start) - start
    division = int(seconds / division_seconds)
    mjd1 = start + float(division) * division_seconds
    mjd2 = start + float(division + 1) * division_seconds
    rate = obs.rate
    samples = int(round((mjd1 - start) * rate))               # lrint
    lower = start + samples / rate
    division_ndat = int(round((mjd2 - lower) * rate))
    return division, lower, division_ndat


def subint_sample_bounds(obs: Observation, division_seconds: float, division: int) -> tuple[int, int]:
    """[first, last) output sample of a division, following TimeDivide::set_boundaries (TimeDivide.C:503-540):
    lower = start + lrint(k*L*rate)/rate ; division_ndat = lrint((start + (k+1)*L - lower)*rate)."""
    rate = obs.rate
    lower = int(round(float(division) * division_seconds * rate))
    ndat = int(round((float(division + 1) * division_seconds - lower / rate) * rate))
    return lower, lower + ndat


# --------------------------------------------------------------------------------------
# Sub-integration division in turns (a12, dspsr -s / -turns)   TimeDivide.C:360-436,461-500
# --------------------------------------------------------------------------------------
# Pulsar::Predictor::phase / iphase are PSRCHIVE (ext, absent): phase(t) is the TEMPO polyco of class Polyco above with
# integer and fractional turns carried separately, iphase its inverse by Newton iteration on phase(t) - target = 0 with
# slope frequency(t) (the method PSRCHIVE's polyco uses).  "parity unpinned" for these two, like the polyco itself.

def predictor_phase(cfg: "FoldConfig", obs: Observation, t_seconds: float) -> tuple[int, float]:
    """Absolute pulse phase (integer turns, fractional turns in [0,1)) at t_seconds after obs.utc_start."""
    if cfg.folding_period > 0.0:
        turns = (t_seconds - cfg.reference_epoch_seconds) / cfg.folding_period
        i = math.floor(turns)
        return int(i), turns - i
    day, sec = utc_to_mjd(obs.utc_start)
    sec += t_seconds
    pc = cfg.polyco
    if isinstance(pc, ChebyPredictor):
        return pc.phase(day, sec)
    dt = pc._dt_min(day, sec)
    poly = 0.0
    for c in pc.coef[::-1]:
        poly = poly * dt + c
    spin = 60.0 * dt * pc.f0
    si = math.floor(spin)
    fr = pc.rphase_frac + (spin - si) + poly
    fi = math.floor(fr)
    return int(pc.rphase_int + si + fi), fr - fi


def predictor_iphase(cfg: "FoldConfig", obs: Observation, phase: tuple[int, float], t_guess: float) -> float:
    """Time (seconds after obs.utc_start) at which the phase equals `phase` = (int turns, frac turns)."""
    if cfg.folding_period > 0.0:
        return cfg.reference_epoch_seconds + (phase[0] + phase[1]) * cfg.folding_period
    t = t_guess
    for _ in range(20):
        pi, pf = predictor_phase(cfg, obs, t)
        dphi = (pi - phase[0]) + (pf - phase[1])
        day, sec = utc_to_mjd(obs.utc_start)
        f = cfg.polyco.frequency(day, sec + t)
        step = dphi / f
        t -= step
        if abs(step) < 1e-12:
            break
    return t


def subint_turns_start(cfg: "FoldConfig", obs: Observation, division_turns: float, fractional_pulses: bool = False):
    """First call of TimeDivide::set_boundaries in turns mode (TimeDivide.C:360-436): with division_turns >= 1 the divisions
    start at the first epoch at or after the observation start where the fractional phase equals reference_phase
    (unless fractional_pulses); with division_turns < 1 at the first boundary reference_phase + N*division_turns after
    the current phase.  -> (start_phase (int, frac), start_time seconds)."""
    pi, pf = predictor_phase(cfg, obs, obs.start_seconds)
    if division_turns < 1.0:
        # phase-resolved divisions (TimeDivide.C:374-425): X = R + N*D, the first division boundary after the current phase
        x_minus_r = pf - cfg.reference_phase
        if pf < cfg.reference_phase:
            x_minus_r += 1.0
            pi -= 1
        n = int(math.ceil(x_minus_r / division_turns))
        x = cfg.reference_phase + n * division_turns
        xi = math.floor(x)                                    # Pulsar::Phase (turns, fracturns) settles the carry
        start_phase = (pi + int(xi), x - xi)
    else:
        if not fractional_pulses and pf > cfg.reference_phase:
            pi += 1
        start_phase = (pi, cfg.reference_phase)
    return start_phase, predictor_iphase(cfg, obs, start_phase, obs.start_seconds)


def subint_turns_sample_bounds(cfg: "FoldConfig", obs: Observation, division_turns: float, division: int,
                               fractional_pulses: bool = False) -> tuple[int, int]:
    """[first, last) output sample of division k in turns mode: boundaries iphase(start_phase + k*D) and
    iphase(start_phase + (k+1)*D) (TimeDivide.C:461-500), snapped to samples like set_boundaries(mjd1, mjd2) (:503-540)."""
    (pi, pf), t0 = subint_turns_start(cfg, obs, division_turns, fractional_pulses)
    rate = obs.rate

    def at(turns):
        tot = pf + turns
        ti = math.floor(tot)
        guess = cfg.folding_period or (1.0 / cfg.polyco.f0 if hasattr(cfg.polyco, "f0") else
                                       1.0 / cfg.polyco.frequency(*[a + b for a, b in zip(utc_to_mjd(obs.utc_start), (0, t0))]))
        return predictor_iphase(cfg, obs, (pi + int(ti), tot - ti), t0 + turns * guess)
    mjd1, mjd2 = at(division * division_turns), at((division + 1) * division_turns)
    samples = int(round((mjd1 - obs.start_seconds) * rate))
    lower = obs.start_seconds + samples / rate
    division_ndat = int(round((mjd2 - lower) * rate))
    return samples, samples + division_ndat


# --------------------------------------------------------------------------------------
# End-to-end convenience used by tests / cpu_baseline
# --------------------------------------------------------------------------------------

def run_pipeline(raw: np.ndarray, obs: Observation, nchan: int, dm: float, nbin: int,
                 folding_period: float = 0.0, polyco: Polyco | None = None, freq_res: int = 0,
                 state: str = "Coherence", ndim: int = 4, dtype=np.float32, block_parts: int = 0):
    """raw bytes -> PhaseSeries, one block (or blocks of `block_parts` parts with overlap carry-over,
    reproducing Filterbank.C:443-444 InputBuffering semantics)."""
    obs = Observation(**obs.__dict__)
    obs.dispersion_measure = dm
    resp = Dedispersion()
    if freq_res:
        resp.set_frequency_resolution(freq_res)
    resp.match(obs, nchan)
    plan = filterbank_plan(obs, nchan, resp)
    unpacked = unpack_8bit(raw, obs)
    fb = filterbank(unpacked, plan, resp.buffer, dtype=dtype)
    fobs = filterbank_output_observation(obs, plan)
    prod = detect_products(fb, state)
    det = detect_layout(prod, ndim)
    if ndim == 1:
        det = det[..., None]
    ps = PhaseSeries(nchan, det.shape[1], det.shape[3], nbin,
                     data=np.zeros((nchan, det.shape[1], nbin, det.shape[3]), dtype=dtype))
    cfg = FoldConfig(nbin=nbin, folding_period=folding_period, polyco=polyco)
    fold(det, fobs, cfg, ps)
    return ps, plan, resp, fb, det
