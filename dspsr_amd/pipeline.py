"""Host pipeline for the hot path: the caller of the engines, mirroring what
dsp::LoadToFold / SingleThread::run do around them (Signal/Pulsar/LoadToFold1.C:117-880,
Signal/General/SingleThread.C:355-497):

  per block:  Filterbank (+ fused Detection) -> Fold::fold plan + accumulate
  per sub-integration (Subint<Fold>, Signal/Pulsar/dsp/Subint.h:234-309): emit + zero the profile;
      with more than one rank this is where the single RCCL reduce of the per-sub-band
      profiles happens (PhaseSeries::combine semantics, PhaseSeries.C:442-484).

Only plumbing lives here; all arithmetic on samples is done by libdspsr_amd.so.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np

from . import _lib
from .engine import (Context, ConvolutionEngine, Dedispersion, DspsrAmdError, FilterbankEngine, FoldEngine, Rescale, SampleDelay, add_fpt, copy_data_fpt,
                     dedispersion_sample_delays, eight_bit_scale, fscrunch_fpt, pscrunch_tfp, sigproc_digitize, sigproc_digitize_fpt,
                     tfp_filterbank, tscrunch_fpt)


@dataclass
class Config:
    """The dspsr command-line options that matter on this path (dspsr.C:207-510)."""
    nchan: int = 1024                 # -F nchan:D
    dispersion_measure: float = 1000  # -D
    nbin: int = 1024                  # -b
    folding_period: float = 0.0       # -c  (seconds); 0 => polyco
    freq_res: int = 0                 # -x  (0 => optimal)
    subint_seconds: float = 0.0       # -L  (0 => single integration)
    subint_turns: float = 0.0         # -s (1 turn: single pulses) / -turns N: sub-integrations of N pulse periods
    fractional_pulses: bool = False   # -fractional_pulses: start the first division at the first sample, not at phase 0
    stokes: bool = False              # -4? (default Coherence, LoadToFold1.C:1119-1134)
    ndim: int = 4                     # detected layout (CPU default 4, CUDA engine 2; LoadToFoldConfig.C:104)
    parts_per_block: int = 64         # block size in overlap-save parts (LoadToFold1.C:825-835 sizes blocks likewise)
    max_parts: int = 32               # parts per launch group (persistent kernels: 32 amortise ramp-up and tail)
    fused_fold: bool = True           # fold inside the last filterbank pass when possible (identical sums, no
                                      # detected time series in HBM); False = Detection and Fold as separate ops
    force_fused: bool = False         # fuse also where the channel tiles do not fill the chip (slower there, same sums)
    two_pass: bool = True             # short responses (complex dual-pol input, nchan_subband * freq_res^2 == 2^27): forward and
                                      # inverse transforms in two workgroup tiles; False = the three-pass kernels (A/B runs)
    interchan_dedispersion: bool = False   # -K: remove the inter-channel dispersion delay (LoadToFold1.C:605-624)
    convolve_when: str = "during"          # -F N:D = "during" (the response inside the filterbank, LoadToFold1.C:318-323); -F N = "after":
                                           # non-convolving filterbank (freq_res 1), then dsp::Convolution on its channels
                                           # (Filterbank::Config::After, FilterbankConfig.C:56, LoadToFold1.C:337-380); -F N:B = "before":
                                           # dsp::Convolution of the whole input channel, then the non-convolving filterbank
                                           # (Config::Before, FilterbankConfig.C:76-78, LoadToFold1.C:326-385); "never": the filterbank
                                           # alone (no coherent dedispersion)
    record_time: bool = False              # -r: time every operation (Operation.C:90-113); each one then ends with a stream
                                           # synchronisation so that the wall times are honest (FilterbankCUDA.cu:302-303)


@dataclass
class InputInfo:
    """What ASCIIObservation.C:82-415 reads from the DADA header."""
    centre_frequency: float = 1382.0
    bandwidth: float = -400.0
    nchan: int = 1
    npol: int = 2
    ndim: int = 1
    tsamp_us: float = 0.00125
    machine: str = "CASPSR"
    start_seconds: float = 0.0        # of the first sample, relative to UTC_START
    mjd_day: int = 55299
    mjd_sec: float = 7545.0
    source: str = "unknown"
    telescope: str = "unknown"

    @property
    def rate(self):
        return 1e6 / self.tsamp_us


class Polyco:
    """TEMPO polyco predictor (the role of Pulsar::Predictor, PSRCHIVE ext): phase and frequency."""

    def __init__(self, text: str):
        tok = text.replace("D", "E").split()
        day, frac = tok[3].split(".")
        ri, rf = tok[7].split(".")
        self.tmid_day, self.tmid_frac = int(day), float("0." + frac)
        self.rphase_int, self.rphase_frac = int(ri), float("0." + rf)
        self.f0 = float(tok[8])
        ncoef = int(tok[11])
        self.coef = [float(t) for t in tok[13:13 + ncoef]]

    def _dt(self, day, sec):
        return ((day - self.tmid_day) + (sec / 86400.0 - self.tmid_frac)) * 1440.0

    def phase_frac(self, day, sec):
        dt = self._dt(day, sec)
        poly = 0.0
        for c in reversed(self.coef):
            poly = poly * dt + c
        spin = 60.0 * dt * self.f0
        ph = (self.rphase_frac + (spin - math.floor(spin))) + poly
        return ph - math.floor(ph)

    def frequency(self, day, sec):
        dt = self._dt(day, sec)
        d = 0.0
        for k in range(len(self.coef) - 1, 0, -1):
            d = d * dt + k * self.coef[k]
        return self.f0 + d / 60.0

    def phase(self, day, sec):
        """Absolute phase as (integer turns, fractional turns) -- Pulsar::Predictor::phase keeps the two apart because
        the total (3.6e9 turns for vela.polyco) does not fit a double to 1e-6 turns."""
        dt = self._dt(day, sec)
        poly = 0.0
        for c in reversed(self.coef):
            poly = poly * dt + c
        spin = 60.0 * dt * self.f0
        si = math.floor(spin)
        fr = self.rphase_frac + (spin - si) + poly
        fi = math.floor(fr)
        return int(self.rphase_int + si + fi), fr - fi

    def iphase(self, phase, day, sec_guess):
        """Pulsar::Predictor::iphase: the epoch (seconds of `day`) at which the phase is `phase` = (int, frac) turns,
        by Newton iteration with slope frequency(t)."""
        sec = sec_guess
        for _ in range(20):
            pi, pf = self.phase(day, sec)
            step = ((pi - phase[0]) + (pf - phase[1])) / self.frequency(day, sec)
            sec -= step
            if abs(step) < 1e-12:
                break
        return sec


class ChebyPredictor:
    """TEMPO2 predictor (`dspsr -P file` with a `ChebyModelSet`; PSRCHIVE `Pulsar::T2Predictor` over tempo2's
    `T2Predictor` library -- both external to the reference tree, `Fold.C:229-262` only asks the generator for one): per
    segment a two-dimensional Chebyshev series in time and observing frequency,
        phase(t, f) = sum'_i sum'_j c[i][j] T_i(x) T_j(y) + DISPERSION_CONSTANT / f^2,
        x = -1 + 2 (t - t0) / (t1 - t0) over TIME_RANGE (MJD),  y = -1 + 2 (f - f0) / (f1 - f0) over FREQ_RANGE (MHz),
    primed sums (first term halved), spin frequency = d phase / dt.  Same duck-typed interface as `Polyco` (phase_frac,
    frequency, phase, iphase), so `LoadToFold(polyco=ChebyPredictor(text))` folds with it.  The constant term (1e9 ... 1e11
    turns) is split into integer and fractional turns from its decimal text, so that the fractional phase keeps double
    precision.  Parity: unpinned (tempo2 is not in the image; the series convention is the one its published
    construction -- coefficients 4/(nx ny) times the cosine sums -- implies)."""

    def __init__(self, text: str, observing_frequency: float | None = None):
        from decimal import Decimal
        self.segments = []
        seg = None
        for line in text.splitlines():
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "ChebyModel" and tok[1] == "BEGIN":
                seg = {"coef": []}
            elif tok[0] == "ChebyModel" and tok[1] == "END":
                nx, ny = seg["nx"], seg["ny"]
                flat = seg.pop("coef")
                if len(flat) != nx * ny:
                    raise DspsrAmdError("ChebyPredictor: %d coefficients, NCOEFF_TIME x NCOEFF_FREQ = %d" % (len(flat), nx * ny))
                # c[i][j]: one COEFFS record per time index i, NCOEFF_FREQ values each; the constant is kept as text
                c00 = Decimal(flat[0]) / 4
                seg["c00_int"] = int(c00 // 1)
                seg["c00_frac"] = float(c00 - (c00 // 1))
                seg["c"] = [[float(flat[i * ny + j]) for j in range(ny)] for i in range(nx)]
                seg["c"][0][0] = 0.0
                self.segments.append(seg)
                seg = None
            elif seg is not None:
                if tok[0] == "TIME_RANGE":
                    seg["t0"], seg["t1"] = self._mjd(tok[1]), self._mjd(tok[2])
                elif tok[0] == "FREQ_RANGE":
                    seg["f0"], seg["f1"] = float(tok[1]), float(tok[2])
                elif tok[0] == "DISPERSION_CONSTANT":
                    seg["dc"] = float(tok[1])
                elif tok[0] == "NCOEFF_TIME":
                    seg["nx"] = int(tok[1])
                elif tok[0] == "NCOEFF_FREQ":
                    seg["ny"] = int(tok[1])
                elif tok[0] == "COEFFS":
                    seg["coef"].extend(tok[1:])
                elif tok[0][0] in "+-.0123456789":       # continuation line of a COEFFS record
                    seg["coef"].extend(tok)
        if not self.segments:
            raise DspsrAmdError("ChebyPredictor: no ChebyModel in the predictor text")
        s0 = self.segments[0]
        self.observing_frequency = observing_frequency if observing_frequency is not None else 0.5 * (s0["f0"] + s0["f1"])

    @staticmethod
    def _mjd(tok):
        day, _, frac = tok.partition(".")
        return int(day), float("0." + (frac or "0"))

    def _segment(self, day, sec):
        t = sec / 86400.0
        for s in self.segments:
            a = (day - s["t0"][0]) + (t - s["t0"][1])
            b = (s["t1"][0] - day) + (s["t1"][1] - t)
            if a >= 0.0 and b >= 0.0:
                return s, a, a + b
        raise DspsrAmdError("ChebyPredictor: MJD %d + %.3f s is outside every TIME_RANGE" % (day, sec))

    def _time_series(self, s):
        """The series in x alone at the observing frequency: a[i] = sum'_j c[i][j] T_j(y); plus the constant parts."""
        key = ("a", self.observing_frequency)
        if s.get("key") != key:
            y = -1.0 + 2.0 * (self.observing_frequency - s["f0"]) / (s["f1"] - s["f0"])
            ty = [1.0, y]
            for j in range(2, s["ny"]):
                ty.append(2.0 * y * ty[-1] - ty[-2])
            s["a"] = [sum((0.5 if j == 0 else 1.0) * s["c"][i][j] * ty[j] for j in range(s["ny"])) for i in range(s["nx"])]
            s["disp"] = s.get("dc", 0.0) / (self.observing_frequency * self.observing_frequency)
            s["key"] = key
        return s["a"]

    def _eval(self, day, sec):
        s, a, span = self._segment(day, sec)
        x = -1.0 + 2.0 * a / span
        co = self._time_series(s)
        tprev, tcur = 1.0, x
        val = 0.5 * co[0] + (co[1] * x if len(co) > 1 else 0.0)
        # derivative with respect to x: T_n' = n U_{n-1};  U_0 = 1, U_1 = 2x, U_n = 2x U_{n-1} - U_{n-2}
        uprev, ucur = 1.0, 2.0 * x
        der = co[1] if len(co) > 1 else 0.0
        for n in range(2, len(co)):
            tprev, tcur = tcur, 2.0 * x * tcur - tprev
            val += co[n] * tcur
            der += co[n] * n * ucur
            uprev, ucur = ucur, 2.0 * x * ucur - uprev
        return s, val + s["disp"], der * 2.0 / (span * 86400.0)

    def phase(self, day, sec):
        s, val, _ = self._eval(day, sec)
        fr = s["c00_frac"] + val
        fi = math.floor(fr)
        return int(s["c00_int"] + fi), fr - fi

    def phase_frac(self, day, sec):
        return self.phase(day, sec)[1]

    def frequency(self, day, sec):
        return self._eval(day, sec)[2]

    def iphase(self, phase, day, sec_guess):
        sec = sec_guess
        for _ in range(20):
            pi, pf = self.phase(day, sec)
            step = ((pi - phase[0]) + (pf - phase[1])) / self.frequency(day, sec)
            sec -= step
            if abs(step) < 1e-12:
                break
        return sec


def choose_nbin(folding_period, rate, requested_nbin=0, maximum_nbin=1024, minimum_bin_width=1.2,
                power_of_two=True, force_sensible_nbin=False):
    """dsp::Fold::choose_nbin (Fold.C:291-382): largest power of two <= period / (1.2 * tsamp), capped at
    maximum_nbin, unless -b requested_nbin was given."""
    if folding_period <= 0.0:
        raise DspsrAmdError("dsp::Fold::choose_nbin invalid folding period=%f" % folding_period)
    ratio = folding_period / (minimum_bin_width / rate)
    sensible = int(2.0 ** math.floor(math.log(ratio) / math.log(2.0))) if power_of_two else int(ratio)
    sensible = max(sensible, 1)
    if requested_nbin > 1:
        return sensible if (force_sensible_nbin and requested_nbin > sensible) else requested_nbin
    return maximum_nbin if (maximum_nbin and sensible > maximum_nbin) else sensible


def subint_bounds(k, division_seconds, rate):
    """[first, last) output sample of division k, with the two roundings of TimeDivide::set_boundaries
    (TimeDivide.C:503-540): lower = lrint(k*L*rate); division_ndat = lrint((k+1)*L*rate - lower) evaluated in
    seconds like the reference.  Python's round() is the same round-half-even as lrint."""
    lower = int(round(float(k) * division_seconds * rate))
    ndat = int(round((float(k + 1) * division_seconds - lower / rate) * rate))
    return lower, lower + ndat


def subint_pieces(first_sample, ndat, division_seconds, rate):
    """Subint<Fold> / TimeDivide in seconds mode (TimeDivide.C:440-459,503-540, Subint.h:234-309), expressed
    in output samples (the observation start is the start of division 0).  Splits the block
    [first_sample, first_sample+ndat) into (idat_start, ndat_fold, division, division_complete) pieces."""
    out = []
    pos = first_sample
    end = first_sample + ndat
    k = max(int(pos / (division_seconds * rate)) - 1, 0)
    while subint_bounds(k, division_seconds, rate)[1] <= pos:
        k += 1
    while pos < end:
        upper = subint_bounds(k, division_seconds, rate)[1]
        stop = min(end, upper)
        out.append((pos - first_sample, stop - pos, k, stop == upper))
        pos = stop
        if pos == upper:
            k += 1
    return out


class TurnsDivider:
    """dsp::TimeDivide in turns mode (dspsr -s / -turns N; TimeDivide.C:360-436,461-500): division k spans pulse phases
    [start_phase + k*D, start_phase + (k+1)*D), start_phase = the first phase at or after the observation start whose
    fractional part is reference_phase (the leading partial turn is not folded) unless fractional_pulses.  Boundaries
    are snapped to output samples like set_boundaries(mjd1, mjd2) (:503-540).  D < 1 (phase-resolved divisions): the first
    boundary reference_phase + N*D after the start (:374-425).  phase(t) -> (int, frac), iphase((int, frac), t_guess) -> t, t in seconds of the stream."""

    def __init__(self, phase, iphase, period_guess, t_start, rate, division_turns, reference_phase=0.0,
                 fractional_pulses=False):
        if division_turns <= 0.0:
            raise DspsrAmdError("dsp::TimeDivide division_turns must be positive")
        self.iphase, self.t_start, self.rate, self.D, self.pguess = iphase, t_start, rate, float(division_turns), period_guess
        pi, pf = phase(t_start)
        if division_turns < 1.0:
            # phase-resolved divisions (:374-425): X = R + N*D, the first division boundary after the current phase
            x_minus_r = pf - reference_phase
            if pf < reference_phase:
                x_minus_r += 1.0
                pi -= 1
            x = reference_phase + int(math.ceil(x_minus_r / division_turns)) * division_turns
            xi = math.floor(x)
            self.start_phase = (pi + int(xi), x - xi)
        else:
            if not fractional_pulses and pf > reference_phase:
                pi += 1
            self.start_phase = (pi, reference_phase)
        self.start_time = iphase(self.start_phase, t_start)
        self._cache = {}

    def _at(self, turns):
        tot = self.start_phase[1] + turns
        ti = math.floor(tot)
        return self.iphase((self.start_phase[0] + int(ti), tot - ti), self.start_time + turns * self.pguess)

    def bounds(self, k):
        """[first, last) output sample of division k (sample 0 = first output sample of the stream)."""
        if k not in self._cache:
            mjd1, mjd2 = self._at(k * self.D), self._at((k + 1) * self.D)
            lower = int(round((mjd1 - self.t_start) * self.rate))                      # lrint
            ndat = int(round((mjd2 - (self.t_start + lower / self.rate)) * self.rate))
            self._cache[k] = (lower, lower + ndat)
            if len(self._cache) > 64:
                self._cache.pop(next(iter(self._cache)))
        return self._cache[k]

    def pieces(self, first_sample, ndat):
        """Subint<Fold> over one block: (idat_start, ndat_fold, division, division_complete) pieces; samples in front of
        division 0 are skipped (TimeDivide::set_bounds, :196-246: idat_start = rint((lower - input_start) * rate))."""
        out = []
        pos, end = first_sample, first_sample + ndat
        pos = max(pos, self.bounds(0)[0])
        if pos >= end:
            return out
        k = max(int((pos - self.bounds(0)[0]) / (self.D * self.pguess * self.rate)) - 1, 0)
        while self.bounds(k)[1] <= pos:
            k += 1
        while pos < end:
            upper = self.bounds(k)[1]
            stop = min(end, upper)
            out.append((pos - first_sample, stop - pos, k, stop == upper))
            pos = stop
            if pos == upper:
                k += 1
        return out


def normalise_profile(profile, hits, scale):
    """dsp::Archiver::set (Archiver.C:773-893): amps = sum / (scale * hits); bins without hits take the mean
    of the others.  profile: [nchan][npol][nbin][ndim] sums, hits: [nbin]."""
    hits = np.asarray(hits, dtype=np.float64)
    prof = np.asarray(profile, dtype=np.float64)
    ok = hits > 0
    amps = np.zeros_like(prof)
    amps[:, :, ok, :] = prof[:, :, ok, :] / (scale * hits[ok])[None, None, :, None]
    if ok.any() and not ok.all():
        amps[:, :, ~ok, :] = amps[:, :, ok, :].mean(axis=2, keepdims=True)
    return amps.astype(np.float32)


# ---- archive hand-off: a PhaseSeries on disk, one file per sub-integration ------------------------
# DSPSR's Archiver needs PSRCHIVE (not available to this library), so a finished sub-integration is handed over as a
# self-describing file that carries exactly the members dsp::Archiver::set reads from a dsp::PhaseSeries
# (Archiver.C:430-893): the Observation keys in the DADA ASCII convention (ASCIIObservation.C:82-415), then
# hits[nbin] (uint32, little endian) and the profile sums [nchan][npol][nbin][ndim] (float32, little endian,
# un-normalised: Archiver divides by scale*hits, :773-893).  INTEGRATION.md shows the reader a maintainer adds.
def subint_profile(sub):
    """The profile of one entry of LoadToFold.subints as a host float32 array, whichever path delivered it: `profile` (host
    memory: the RCCL exchange of dspsr_amd_reduce_profiles_finish, or merge_subints) or `profile_dev` (a device tensor: one
    rank, or the torch.distributed forms)."""
    prof = sub.get("profile")
    if prof is None:
        prof = sub["profile_dev"].cpu().numpy()
    return np.asarray(prof, dtype=np.float32)


PHASE_SERIES_MAGIC = "DSPSR_AMD_PHASESERIES"
PHASE_SERIES_HDR_SIZE = 4096


def write_phase_series(path, sub, info: "InputInfo", cfg: "Config", *, nchan=None, npol=1, scale=1.0, division=0,
                       start_seconds=0.0, folding_period=0.0, reference_phase=0.0, state=None):
    """sub: a dict as LoadToFold.subints holds (hits, integration_length, ndat_total, profile or profile_dev)."""
    prof = sub.get("profile")
    if prof is None:
        prof = sub["profile_dev"].cpu().numpy()
    nchan = nchan or cfg.nchan
    nbin = len(sub["hits"])
    prof = np.ascontiguousarray(np.asarray(prof, dtype="<f4").reshape(nchan, npol, nbin, cfg.ndim))
    keys = [("HDR_MAGIC", PHASE_SERIES_MAGIC), ("HDR_VERSION", "1.0"), ("HDR_SIZE", PHASE_SERIES_HDR_SIZE),
            ("FREQ", repr(float(info.centre_frequency))), ("BW", repr(float(info.bandwidth))), ("NCHAN", nchan),
            ("NPOL", npol), ("NDIM", cfg.ndim), ("NBIN", nbin),
            ("STATE", state or ("Stokes" if cfg.stokes else "Coherence")),
            ("DM", repr(float(cfg.dispersion_measure))), ("SCALE", repr(float(scale))),
            ("MJD_DAY", info.mjd_day), ("MJD_SEC", repr(float(info.mjd_sec))),
            ("OBS_OFFSET_SECONDS", repr(float(start_seconds))), ("DIVISION", division),
            ("INTEGRATION_LENGTH", repr(float(sub["integration_length"]))), ("NDAT_TOTAL", int(sub["ndat_total"])),
            ("FOLDING_PERIOD", repr(float(folding_period))), ("REFERENCE_PHASE", repr(float(reference_phase)))]
    text = "".join("%-20s %s\n" % (k, v) for k, v in keys)
    if len(text) >= PHASE_SERIES_HDR_SIZE:
        raise DspsrAmdError("write_phase_series: header does not fit %d bytes" % PHASE_SERIES_HDR_SIZE)
    with open(path, "wb") as f:
        f.write(text.encode("ascii").ljust(PHASE_SERIES_HDR_SIZE, b"\0"))
        f.write(np.asarray(sub["hits"], dtype="<u4").tobytes())
        f.write(prof.tobytes())


def read_phase_series(path):
    """Returns (header dict, hits uint32[nbin], profile float32 [nchan][npol][nbin][ndim])."""
    with open(path, "rb") as f:
        raw = f.read(PHASE_SERIES_HDR_SIZE)
        hdr = {}
        for line in raw.split(b"\0", 1)[0].decode("ascii").splitlines():
            parts = line.split(None, 1)
            if len(parts) == 2:
                hdr[parts[0]] = parts[1].strip()
        if hdr.get("HDR_MAGIC") != PHASE_SERIES_MAGIC:
            raise DspsrAmdError("read_phase_series: %s is not a %s file" % (path, PHASE_SERIES_MAGIC))
        nchan, npol, nbin, ndim = (int(hdr[k]) for k in ("NCHAN", "NPOL", "NBIN", "NDIM"))
        hits = np.frombuffer(f.read(4 * nbin), dtype="<u4").copy()
        body = f.read(4 * nchan * npol * nbin * ndim)
        if len(hits) != nbin or len(body) != 4 * nchan * npol * nbin * ndim:
            raise DspsrAmdError("read_phase_series: %s is truncated" % path)
        prof = np.frombuffer(body, dtype="<f4").reshape(nchan, npol, nbin, ndim).copy()
    return hdr, hits, prof


class PhaseSeries:
    """The product's dsp::PhaseSeries (Signal/Pulsar/dsp/PhaseSeries.h:163-200): profile sums resident on the device
    ([nchan][npol][nbin*ndim] float32 tensor), hits[] and the integration bookkeeping on the host, plus the Observation
    attributes `combinable` compares.  mixable / combine follow PhaseSeries.C:336-418,442-484; the profile half of
    combine is TimeSeries::operator += on the device (dspsr_amd_add_fpt)."""

    OBS_KEYS = ("centre_frequency", "bandwidth", "nchan", "npol", "ndim", "state", "rate")

    def __init__(self, ctx, obs=None):
        self.ctx = ctx
        self.obs = dict(obs or {})
        self.nbin = 0
        self.profile = None
        self.hits = np.zeros(0, np.uint32)
        self.integration_length = 0.0
        self.ndat_total = 0
        self.start_time = self.end_time = 0.0

    @classmethod
    def from_subint(cls, ctx, sub, obs, start_time=0.0, end_time=0.0):
        """Wrap a dict of LoadToFold.subints (hits, integration_length, ndat_total and the device copy `profile_dev`, or the
        host array `profile` an RCCL exchange delivered -- uploaded here)."""
        ps = cls(ctx, obs)
        ps.nbin = len(sub["hits"])
        dev = sub.get("profile_dev")
        if dev is None:
            import torch
            dev = torch.from_numpy(np.ascontiguousarray(subint_profile(sub)).reshape(-1)).to("cuda:%d" % ctx.device)
        ps.profile = dev.view(obs["nchan"], obs["npol"], ps.nbin * obs["ndim"])
        ps.hits = np.array(sub["hits"], dtype=np.uint32)
        ps.integration_length, ps.ndat_total = float(sub["integration_length"]), int(sub["ndat_total"])
        ps.start_time, ps.end_time = start_time, end_time
        return ps

    def zero(self):                                            # PhaseSeries::zero, PhaseSeries.C:239-256
        self.integration_length, self.ndat_total = 0.0, 0
        self.hits[:] = 0
        if self.profile is not None:
            self.profile.zero_()

    def combinable(self, obs):                                 # Observation::combinable: the attributes that must agree
        return all(self.obs.get(k) == obs.get(k) for k in self.OBS_KEYS)

    def mixable(self, obs, nbin, start_time, end_time):
        """PhaseSeries::mixable (PhaseSeries.C:336-418): an empty integration adopts `obs`, is resized and zeroed (the
        count of dropped samples ndat_total survives); otherwise the observations must be combinable and nbin equal, and
        the time span is extended."""
        if self.integration_length == 0.0:
            import torch
            self.obs = {k: obs.get(k) for k in self.OBS_KEYS}
            keep = self.ndat_total
            self.nbin = nbin
            shape = (obs["nchan"], obs["npol"], nbin * obs["ndim"])
            if self.profile is None or tuple(self.profile.shape) != shape:
                self.profile = torch.empty(shape, dtype=torch.float32, device="cuda:%d" % self.ctx.device)
            self.hits = np.zeros(nbin, np.uint32)
            self.zero()
            self.ndat_total = keep
            self.start_time, self.end_time = start_time, end_time
            return True
        if not self.combinable(obs) or self.nbin != nbin:
            return False
        self.end_time = max(self.end_time, end_time)
        self.start_time = min(self.start_time, start_time)
        return True

    def combine(self, other):
        """PhaseSeries::combine (PhaseSeries.C:442-484)."""
        if other is None or other.nbin == 0:
            return self
        if not self.integration_length:                        # this is empty: *this = *prof
            keep = self.ndat_total
            self.mixable(other.obs, other.nbin, other.start_time, other.end_time)
            self.profile.copy_(other.profile)
            self.hits = other.hits.copy()
            self.integration_length, self.ndat_total = other.integration_length, other.ndat_total + 0 * keep
            return self
        if not self.mixable(other.obs, other.nbin, other.start_time, other.end_time):
            raise DspsrAmdError("PhaseSeries::combine PhaseSeries !mixable")
        add_fpt(self.ctx, self.profile, other.profile)         # TimeSeries::operator +=
        self.hits += other.hits
        self.integration_length += other.integration_length
        self.ndat_total += other.ndat_total
        return self


def combine_phase_series(a, b):
    """dsp::PhaseSeries::combine (PhaseSeries.C:442-484) on sub-integration dicts (host arrays): how UnloaderShare
    merges the pieces of one division folded by time-sliced replicas.  An empty `a` (integration_length 0) becomes a
    copy of `b`; otherwise profiles, hits, integration_length and ndat_total add."""
    if b is None or len(b["hits"]) == 0:
        return a
    pb = b.get("profile")
    if pb is None:
        pb = b["profile_dev"].cpu().numpy()
    if a is None or not a["integration_length"]:
        return {"hits": np.array(b["hits"], dtype=np.uint32), "integration_length": float(b["integration_length"]),
                "ndat_total": int(b["ndat_total"]), "profile": np.array(pb, dtype=np.float32)}
    pa = a.get("profile")
    if pa is None:
        pa = a["profile_dev"].cpu().numpy()
    if np.shape(pa) != np.shape(pb) or len(a["hits"]) != len(b["hits"]):
        raise DspsrAmdError("PhaseSeries::combine PhaseSeries !mixable")
    return {"hits": (np.asarray(a["hits"], dtype=np.uint32) + np.asarray(b["hits"], dtype=np.uint32)),
            "integration_length": float(a["integration_length"]) + float(b["integration_length"]),
            "ndat_total": int(a["ndat_total"]) + int(b["ndat_total"]),
            "profile": (np.asarray(pa, dtype=np.float32) + np.asarray(pb, dtype=np.float32))}


def reduce_subbands(prof, dist=None, rank=0, world=1, gather_buffer=None):
    """torch.distributed form of the sub-band exchange (the product's own is dspsr_amd_reduce_profiles_*, csrc/comm.hip,
    DSPSR_AMD_REDUCE_GATHER; this one serves the gloo CPU tests and one-device rehearsals, where RCCL cannot run).
    Each rank holds the folded profile of its own frequency sub-band (flat [nchan*npol*nbin*ndim] float32); a gather
    delivers the slices to rank 0 in rank order = the concatenated band, which is what PhaseSeries::combine of
    disjoint channel ranges amounts to (PhaseSeries.C:442-484).  hits / integration_length are identical on all ranks
    (channel-independent bin plan, Fold.C:744-787) and are taken from rank 0.
    Returns the full-band tensor on rank 0, None elsewhere; with world == 1 returns `prof` itself."""
    if world <= 1:
        return prof
    if gather_buffer is None or gather_buffer.numel() != world * prof.numel():
        raise DspsrAmdError("reduce_subbands: gather buffer must hold world*profile = %d floats"
                            % (world * prof.numel()))
    mine = prof.reshape(-1).contiguous()
    if rank == 0:
        dist.gather(mine, list(gather_buffer.view(world, -1).unbind(0)), dst=0)
        return gather_buffer
    dist.gather(mine, None, dst=0)
    return None


def check_identical_hits(hits, dist, rank, world):
    """Sub-band sharding: every rank folds with the same channel-independent bin plan (Fold.C:744-787), so hits[] must
    be identical on all ranks ('assert equality in debug').  One small all-reduce pair (MAX and MIN);
    raises on every rank when they differ.  Debug/test aid: not part of the timed path."""
    if world <= 1:
        return
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    h = torch.as_tensor(np.asarray(hits, dtype=np.int64), device=dev)
    hi, lo = h.clone(), h.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    if not bool(torch.equal(hi, lo)):
        raise DspsrAmdError("sub-band ranks disagree on hits[]: the shards are not sample aligned "
                            "(different nfilt_pos/neg, start time or rate)")


def reduce_replicas(prof, hits, integration_length, ndat_total, dist=None, rank=0, world=1):
    """Time-slice replicas (single-channel input has no exchange-free frequency split; the reference runs one thread
    per time block, MultiThread.C:65-82,120-148, and merges the pieces of a division with PhaseSeries::combine,
    PhaseSeries.C:442-484): a true SUM of profiles, hits, integration_length and ndat_total onto rank 0.
    prof is reduced in place (one reduce of the profile buffer -- RCCL over xGMI on GPUs, gloo in the CPU tests --
    plus two tiny ones for the counters).  Returns (prof, hits, integration_length, ndat_total) on rank 0, None elsewhere."""
    if world <= 1:
        return prof, np.asarray(hits, dtype=np.uint32), float(integration_length), int(ndat_total)
    import torch
    dev = prof.device
    dist.reduce(prof, dst=0, op=dist.ReduceOp.SUM)
    cnt = torch.as_tensor(np.concatenate([np.asarray(hits, dtype=np.int64), [int(ndat_total)]]), device=dev)
    dist.reduce(cnt, dst=0, op=dist.ReduceOp.SUM)
    length = torch.tensor([float(integration_length)], dtype=torch.float64, device=dev)
    dist.reduce(length, dst=0, op=dist.ReduceOp.SUM)
    if rank != 0:
        return None
    cnt = cnt.cpu().numpy()
    return prof, cnt[:-1].astype(np.uint32), float(length.item()), int(cnt[-1])


class FilterbankThenConvolution:
    """`dspsr -F N` (Filterbank::Config::After, the default of FilterbankConfig.C:56): dsp::Filterbank with freq_res = 1 -- one
    nsub-point forward transform per output sample, no response (LoadToFold1.C:318-323 sets it only for Config::During) -- then
    dsp::Convolution with the dedispersion response on the filterbank's channels (LoadToFold1.C:337-380, Convolution.C:338-461),
    here the filterbank object with nchan_subband = 1 per channel.  `response` None: the filterbank alone (Config::Never /
    no coherent dedispersion).

    Presents the interface LoadToFold uses of a FilterbankEngine (perform_raw / perform_detect / perform_fold on the 8-bit block,
    nkeep / nsamp_step / nsamp_overlap in INPUT samples), so one part here = one part of the convolution = nsamp_step filterbank
    output samples.  The reference carries the convolution's overlap from block to block (InputBuffering); the blocks this
    pipeline is handed overlap in the raw samples instead (block_bytes), and the filterbank recomputes those few samples."""

    def __init__(self, ctx, nsub, input_nchan, npol, real_input, response, max_parts=1, parts_per_block=1,
                 fused_fold=_lib.FUSED_AUTO):
        import torch
        self.torch, self.ctx = torch, ctx
        self.front = FilterbankEngine(ctx).setup(nsub, 1, 0, 0, input_nchan, npol, real_input, None)
        self.nchan, self.npol = nsub * input_nchan, npol
        self.nsamp_fft_front = self.front.nsamp_fft                      # input samples per filterbank output sample
        self.conv = None
        if response is not None:
            # (the convolution's fused fold -- segment sums inside the second inverse pass -- pays a combine launch per input channel:
            #  with 128 channels the detected block + one Fold launch is faster, 12.4 against 18.2 ms per block of `dspsr -F 128`)
            if self.nchan > 8 and fused_fold == _lib.FUSED_AUTO:
                fused_fold = _lib.FUSED_NEVER
            self.conv = ConvolutionEngine(ctx).setup(1, response.ndat, response.impulse_pos, response.impulse_neg, self.nchan, npol,
                                                     False, response.kernel, max_parts=max_parts, fused_fold=fused_fold)
            self.nkeep, self.part_out, self.ovl_out = self.conv.nkeep, self.conv.nsamp_step, self.conv.nsamp_overlap
        else:
            self.nkeep, self.part_out, self.ovl_out = 1, 1, 0
        self.nsamp_step = self.part_out * self.nsamp_fft_front
        self.nsamp_overlap = self.ovl_out * self.nsamp_fft_front
        self.nsamp_fft = self.nsamp_step + self.nsamp_overlap
        self.channelised = None                                           # the filterbank's output block [chan][pol][2 * ndat]
        self._cap = parts_per_block

    def _front(self, raw, layout, scale, npart):
        ndat = npart * self.part_out + self.ovl_out
        if self.channelised is None or self.channelised.shape[2] < 2 * ndat:
            n = max(ndat, self._cap * self.part_out + self.ovl_out)
            n += (-n) % 64                                  # rows start on 512-byte boundaries: the filterbank's runs of 32 samples are whole lines
            self.channelised = self.torch.empty((self.nchan, self.npol, 2 * n), dtype=self.torch.float32,
                                                device="cuda:%d" % self.ctx.device)
        self.front.perform_raw(raw, layout, scale, self.channelised, ndat)
        return self.channelised

    def perform_raw(self, raw, layout, scale, out, npart, out_step=None):
        if self.conv is None:
            return self.front.perform_raw(raw, layout, scale, out, npart, out_step)
        x = self._front(raw, layout, scale, npart)
        self.conv.perform(x, out, npart, 2 * self.part_out, out_step or 2 * self.nkeep)

    def perform_detect(self, det, npart, state=_lib.COHERENCE, ndim=4, raw=None, layout=_lib.RAW_GENERIC, scale=1.0):
        if self.conv is None:
            return self.front.perform_detect(det, npart, state, ndim, raw=raw, layout=layout, scale=scale)
        x = self._front(raw, layout, scale, npart)
        self.conv.perform_detect(det, npart, state, ndim, inp=x, in_step=2 * self.part_out)

    def perform_fold(self, fold, npart, state=_lib.COHERENCE, raw=None, layout=_lib.RAW_GENERIC, scale=1.0):
        if self.conv is None:
            return self.front.perform_fold(fold, npart, state, raw=raw, layout=layout, scale=scale)
        x = self._front(raw, layout, scale, npart)
        self.conv.perform_fold(fold, npart, state, inp=x, in_step=2 * self.part_out)

    def fold_is_fused(self):
        return (self.conv or self.front).fold_is_fused()

    def npass(self, raw_input=True):
        return 1 + (self.conv.npass(False) if self.conv is not None else 0)

    def finish(self):
        self.ctx.synchronize()

    def close(self):
        self.front.close()
        if self.conv is not None:
            self.conv.close()
        self.channelised = None


class ConvolutionThenFilterbank:
    """`dspsr -F N:B` (Filterbank::Config::Before, FilterbankConfig.C:76-78; LoadToFold1.C:326-385): dsp::Convolution with the dedispersion
    response of the WHOLE input channel first (the filterbank object with nchan_subband = 1 on the 8-bit block: Convolution.C:338-461),
    then the non-convolving filterbank (freq_res = 1, Filterbank.C:614-623) on the dedispersed complex rows.  Every channel then
    refers to the input channel's centre frequency: no dispersion delay is left between the output channels.
    The convolution keeps `ck` complex samples per part and the filterbank consumes `nsub` per output sample; the reference carries
    the remainder from block to block (InputBuffering).  Here one part of the interface LoadToFold uses is P = nsub / gcd(ck, nsub)
    parts of the convolution, which hold a whole number of filterbank transforms -- no remainder, same samples."""

    def __init__(self, ctx, nsub, input_nchan, npol, real_input, response, max_parts=1, parts_per_block=1):
        import torch
        self.torch, self.ctx = torch, ctx
        self.conv = ConvolutionEngine(ctx).setup(1, response.ndat, response.impulse_pos, response.impulse_neg, input_nchan, npol, real_input,
                                                 response.kernel, max_parts=max_parts)
        self.back = FilterbankEngine(ctx).setup(nsub, 1, 0, 0, input_nchan, npol, False, None)
        ck = self.conv.nkeep
        self.P = nsub // math.gcd(ck, nsub)
        self.nsub, self.in_nchan, self.npol = nsub, input_nchan, npol
        self.nkeep = self.P * ck // nsub                         # filterbank output samples per (super) part
        self.nsamp_step = self.P * self.conv.nsamp_step
        self.nsamp_overlap = self.conv.nsamp_overlap
        self.nsamp_fft = self.nsamp_step + self.nsamp_overlap
        self.dedispersed = None                                   # the convolution's output block [input chan][pol][2 * ndat]
        self._cap = parts_per_block

    def _front(self, raw, layout, scale, npart):
        ncv = npart * self.P
        ndat = ncv * self.conv.nkeep
        if self.dedispersed is None or self.dedispersed.shape[2] < 2 * ndat:
            n = max(ndat, self._cap * self.P * self.conv.nkeep)
            self.dedispersed = self.torch.empty((self.in_nchan, self.npol, 2 * n), dtype=self.torch.float32,
                                                device="cuda:%d" % self.ctx.device)
        self.conv.perform_raw(raw, layout, scale, self.dedispersed, ncv)
        return self.dedispersed, ndat // self.nsub

    def perform_raw(self, raw, layout, scale, out, npart, out_step=None):
        x, nfb = self._front(raw, layout, scale, npart)
        self.back.perform(x, out, nfb, 2 * self.nsub, out_step or 2)

    def perform_detect(self, det, npart, state=_lib.COHERENCE, ndim=4, raw=None, layout=_lib.RAW_GENERIC, scale=1.0):
        x, nfb = self._front(raw, layout, scale, npart)
        self.back.perform_detect(det, nfb, state, ndim, inp=x, in_step=2 * self.nsub)

    def perform_fold(self, fold, npart, state=_lib.COHERENCE, raw=None, layout=_lib.RAW_GENERIC, scale=1.0):
        x, nfb = self._front(raw, layout, scale, npart)
        self.back.perform_fold(fold, nfb, state, inp=x, in_step=2 * self.nsub)

    def fold_is_fused(self):
        return 0

    def npass(self, raw_input=True):
        return self.conv.npass(raw_input) + 1

    def finish(self):
        self.ctx.synchronize()

    def close(self):
        self.conv.close()
        self.back.close()
        self.dedispersed = None


class LoadToFold:
    """One pipeline instance = one GPU = one stream (SingleThread).  `raw` blocks are int8 torch
    tensors already resident on the device (the PCIe copy is the caller's, as TransferCUDA is a
    separate Operation in the reference)."""

    def __init__(self, cfg: Config, info: InputInfo, device: int = 0, stream: int | None = None,
                 polyco: Polyco | None = None, reference_phase: float = 0.0, subband: int | None = None,
                 dump_before=(), dump_dir="."):
        """subband = g: this instance is rank g of a sub-band sharded run.  `info`/`cfg` still describe the
        WHOLE band (info.nchan input channels, cfg.nchan output channels); the instance processes input channel g only:
        its block holds that channel's bytes alone ([t][pol][dim], what a rank reads from the NCHAN-interleaved file),
        its kernel is the g-th slice of the ONE full-band kernel (Filterbank.C:563, FilterbankCUDA.cu:249-251) with the
        COMMON nfilt_pos/neg, so all ranks stay sample aligned and fold with identical hits."""
        import torch
        self.torch = torch
        self.cfg, self.info, self.polyco = cfg, info, polyco
        self.reference_phase = reference_phase
        self.subband = subband
        if subband is not None and not 0 <= subband < info.nchan:
            raise DspsrAmdError("dspsr_amd.LoadToFold: subband=%d outside the %d input channels" % (subband, info.nchan))
        if cfg.folding_period <= 0 and polyco is None:
            raise DspsrAmdError("dsp::Fold::fold no polynomial and no period specified")   # Fold.C:638-640
        if info.npol != 2:
            raise DspsrAmdError("dsp::Detection::polarimetry Cannot detect polarization when npol != 2")
        if cfg.convolve_when not in ("during", "after", "before", "never"):
            raise DspsrAmdError("dspsr_amd.LoadToFold: convolve_when=%r is not one of during / after / before / never" % (cfg.convolve_when,))
        if cfg.nchan % info.nchan:
            raise DspsrAmdError("dsp::Filterbank::make_preparations output nchan=%d not a multiple of input nchan=%d"
                                % (cfg.nchan, info.nchan))
        if cfg.convolve_when != "during":
            return self._init_after(device, stream, subband, dump_before)
        self.ctx = Context(device, stream)
        # kernel (host) --------------------------------------------------------------------
        self.response = Dedispersion(info.centre_frequency, info.bandwidth, cfg.dispersion_measure,
                                     input_nchan=info.nchan, ndim=info.ndim,
                                     fractional_delay=cfg.interchan_dedispersion)       # LoadToFold1.C:620-621
        if cfg.freq_res:
            self.response.set_frequency_resolution(cfg.freq_res)
        self.response.match(cfg.nchan)
        r = self.response
        # engines --------------------------------------------------------------------------
        nsub = cfg.nchan // info.nchan
        kernel = r.kernel
        if subband is not None:
            kernel = kernel[subband * nsub * r.ndat:(subband + 1) * nsub * r.ndat]
        self.in_nchan = 1 if subband is not None else info.nchan         # input channels in this instance's blocks
        self.nchan_out = nsub * self.in_nchan                              # output channels this instance produces
        self.fb = FilterbankEngine(self.ctx).setup(nsub, r.ndat, r.impulse_pos, r.impulse_neg,
                                                   self.in_nchan, info.npol, info.ndim == 1, kernel,
                                                   max_parts=cfg.max_parts, force_four_pass=0 if cfg.two_pass else 2,
                                                   fused_fold=(_lib.FUSED_NEVER if not cfg.fused_fold else
                                                               _lib.FUSED_ALWAYS if cfg.force_fused else _lib.FUSED_AUTO))
        self.nkeep, self.nsamp_step, self.nsamp_overlap = self.fb.nkeep, self.fb.nsamp_step, self.fb.nsamp_overlap
        self.npol_out = 4 // cfg.ndim
        self.fold = FoldEngine(self.ctx)
        self.fold.set_shape(self.nchan_out, self.npol_out, cfg.ndim, cfg.nbin)
        self.scale8 = eight_bit_scale()
        self.layout = _lib.RAW_CASPSR if info.machine == "CASPSR" else _lib.RAW_GENERIC
        # output observation (Filterbank::prepare_output, Filterbank.C:265-379)
        n_fft = (cfg.nchan // info.nchan) * r.ndat
        nsamp_fft = 2 * n_fft if info.ndim == 1 else n_fft
        self.out_rate = info.rate * (float(r.ndat) / float(nsamp_fft))
        self.out_start = info.start_seconds + r.impulse_pos / self.out_rate
        self.scalefac = float(n_fft) * float(r.ndat)
        # -K: dsp::SampleDelay on the filterbank output.  Detection acts sample by sample, so delaying the detected rows
        # gives the same numbers as delaying the complex rows first (the reference's order) and keeps the fused
        # filterbank+detect launch group.  The last `total_delay` samples of a block are re-presented in front of the
        # next one (InputBuffering, SampleDelay.C:117,146): they live in the head room in front of `detected`.
        self.sample_delay, self.sd_carried, head = None, 0, 0
        if cfg.interchan_dedispersion:
            dual = info.ndim == 2                              # Observation.C:80-87; Filterbank.C:358-364
            delays = dedispersion_sample_delays(info.centre_frequency, info.bandwidth, cfg.dispersion_measure, cfg.nchan,
                                                self.out_rate, swap=dual and info.nchan == 1,
                                                nsub_swap=info.nchan if dual and info.nchan > 1 else 0)
            self.sd_short = 0
            if subband is None:
                self.sample_delay = SampleDelay(self.ctx, delays, self.npol_out)
                head, zero = self.sample_delay.total_delay, self.sample_delay.zero_delay
            else:
                # a sub-band rank applies ITS channels' delays relative to the zero of the WHOLE band and gives up the
                # whole band's total delay per block, so that every rank emits the same samples with the same start time
                # (SampleDelay.C:75-99 on the full band; the rank's slice goes in as absolute delays)
                d = np.asarray(delays, np.int64)
                zero = int(d.max())
                applied = zero - d
                head = int(applied.max())
                mine = applied[subband * nsub:(subband + 1) * nsub]
                self.sample_delay = SampleDelay(self.ctx, mine, self.npol_out, absolute=True)
                self.sd_short = head - self.sample_delay.total_delay       # samples this rank must NOT emit
            if head > cfg.parts_per_block * self.nkeep:
                raise DspsrAmdError("dspsr_amd.LoadToFold: inter-channel delay of %d samples exceeds the block of %d"
                                    % (head, cfg.parts_per_block * self.nkeep))
            self.out_start += zero / self.out_rate                               # SampleDelay.C:159
        self.sd_head = head
        self.detected = torch.empty((self.nchan_out, self.npol_out, (head + cfg.parts_per_block * self.nkeep) * cfg.ndim),
                                    dtype=torch.float32, device="cuda:%d" % device)
        # fused filterbank+detect+fold (no detected time series in HBM): ndim 4, three-pass geometries
        # (the library decides whether fusing pays for this geometry; when it does not, this driver keeps the
        # separate Detection and Fold operations on its own `detected` block)
        self.fused_mode = self.fb.fold_is_fused() if (cfg.fused_fold and cfg.ndim == 4 and self.sample_delay is None) else 0
        self.fused_fold = self.fused_mode != 0      # fused_mode 2: sums re-associated per run of parts (engine.fold_is_fused)
        # stage capture (dspsr --dump <Operation>, SingleThread.C:315-346): pre_Detection.dump / pre_Fold.dump ------------
        self.optime = {}                    # record_time: operation name -> [seconds, calls]
        self.dumps, self._dump_cplx = {}, None
        for name in dump_before:
            if name not in ("Detection", "Fold"):
                raise DspsrAmdError("dsp::SingleThread::insert_dump_point no operation named '%s' on this path "
                                    "(Detection, Fold)" % name)
            if name == "Detection" and self.sample_delay is not None:
                raise DspsrAmdError("dspsr_amd.LoadToFold: the pre_Detection tap is not built for -K (the delay is applied "
                                    "to the detected rows here); tap pre_Fold instead")
            from . import dada
            chbw = info.bandwidth / info.nchan
            fc = info.centre_frequency if subband is None else info.centre_frequency - 0.5 * info.bandwidth + (subband + 0.5) * chbw
            bw = info.bandwidth if subband is None else chbw
            det = name == "Fold"
            path = os.path.join(dump_dir, "pre_%s%s.dump" % (name, "" if subband is None else ".%d" % subband))
            self.dumps[name] = dada.Dump(
                path, centre_frequency=fc, bandwidth=bw, nchan=self.nchan_out, npol=self.npol_out if det else 2,
                ndim=cfg.ndim if det else 2, nbit=32, rate=self.out_rate, mjd_day=info.mjd_day,
                mjd_sec=info.mjd_sec + self.out_start,
                state=("Stokes" if cfg.stokes else "Coherence") if det else "Analytic",
                source=getattr(info, "source", "unknown"), telescope=getattr(info, "telescope", "unknown"))
        if "Fold" in self.dumps:
            self.fused_mode, self.fused_fold = 0, False          # the detected samples must exist in HBM to be dumped
        # fold bookkeeping (PhaseSeries) ---------------------------------------------------
        self.hits = np.zeros(cfg.nbin, dtype=np.uint32)
        self.integration_length = 0.0
        self.ndat_total = 0
        self.nsamples_in = 0            # unique input samples consumed (per pol)
        self.ndat_out = 0               # output samples produced so far
        self.subints = []               # completed sub-integrations (host copies) on the writer rank

    def _init_after(self, device, stream, subband, dump_before):
        """`dspsr -F N` (Filterbank::Config::After) and the filterbank without coherent dedispersion (Config::Never): the non-convolving
        filterbank (freq_res = 1) followed by dsp::Convolution on its output channels (LoadToFold1.C:296-380).  The response is matched
        to the FILTERBANK'S OUTPUT observation (Convolution.C:105-130 -> Dedispersion::match(input)): cfg.nchan channels, complex,
        dual sideband and DC centred (Filterbank.C:341-348: freq_res = 1), band swapped when the input was single-channel dual
        sideband (:358-364; a multi-channel dual-sideband input sets nsub_swap, which Response::match does not read, Response.C:132-181)."""
        torch, cfg, info = self.torch, self.cfg, self.info
        if subband is not None:
            raise DspsrAmdError("dspsr_amd.LoadToFold: sub-band sharded runs are built for -F N:D (convolve_when = during)")
        if cfg.interchan_dedispersion or dump_before:
            raise DspsrAmdError("dspsr_amd.LoadToFold: -K and the dump taps are built for -F N:D (convolve_when = during)")
        self.ctx = Context(device, stream)
        r = None
        if cfg.convolve_when == "after":
            r = Dedispersion(info.centre_frequency, info.bandwidth, cfg.dispersion_measure, input_nchan=cfg.nchan, ndim=2,
                             dual_sideband=1, dc_centred=True, swap=(info.ndim == 2 and info.nchan == 1))
            if cfg.freq_res:
                r.set_frequency_resolution(cfg.freq_res)
            r.match(cfg.nchan)
        if cfg.convolve_when == "before":
            # Convolution::prepare -> Dedispersion::match(input): ONE response per input channel, over its whole width
            r = Dedispersion(info.centre_frequency, info.bandwidth, cfg.dispersion_measure, input_nchan=info.nchan, ndim=info.ndim)
            if cfg.freq_res:
                r.set_frequency_resolution(cfg.freq_res)
            r.match(info.nchan)
        self.response = r
        nsub = cfg.nchan // info.nchan
        self.in_nchan, self.nchan_out = info.nchan, cfg.nchan
        if cfg.convolve_when == "before":
            self._init_before(device, r, nsub)
            return
        self.fb = FilterbankThenConvolution(self.ctx, nsub, info.nchan, info.npol, info.ndim == 1, r, max_parts=cfg.max_parts,
                                            parts_per_block=cfg.parts_per_block,
                                            fused_fold=(_lib.FUSED_NEVER if not cfg.fused_fold else
                                                        _lib.FUSED_ALWAYS if cfg.force_fused else _lib.FUSED_AUTO))
        self.nkeep, self.nsamp_step, self.nsamp_overlap = self.fb.nkeep, self.fb.nsamp_step, self.fb.nsamp_overlap
        self.npol_out = 4 // cfg.ndim
        self.fold = FoldEngine(self.ctx)
        self.fold.set_shape(self.nchan_out, self.npol_out, cfg.ndim, cfg.nbin)
        self.scale8 = eight_bit_scale()
        self.layout = _lib.RAW_CASPSR if info.machine == "CASPSR" else _lib.RAW_GENERIC
        self.out_rate = info.rate / float(self.fb.nsamp_fft_front)                       # Filterbank.C:338-339: freq_res / nsamp_fft
        ndat = r.ndat if r is not None else 1
        self.out_start = info.start_seconds + (r.impulse_pos if r is not None else 0) / self.out_rate   # Convolution.C:300
        self.scalefac = float(nsub) * (float(ndat) * float(ndat) if r is not None else 1.0)            # Filterbank.C:124-125, Convolution.C:305
        self.sample_delay, self.sd_carried, self.sd_head, self.sd_short = None, 0, 0, 0
        self.detected = torch.empty((self.nchan_out, self.npol_out, cfg.parts_per_block * self.nkeep * cfg.ndim),
                                    dtype=torch.float32, device="cuda:%d" % device)
        self.fused_mode = self.fb.fold_is_fused() if (cfg.fused_fold and cfg.ndim == 4) else 0
        self.fused_fold = self.fused_mode != 0
        self.optime, self.dumps, self._dump_cplx = {}, {}, None
        self.hits = np.zeros(cfg.nbin, dtype=np.uint32)
        self.integration_length, self.ndat_total = 0.0, 0
        self.nsamples_in, self.ndat_out, self.subints = 0, 0, []

    def _init_before(self, device, r, nsub):
        torch, cfg, info = self.torch, self.cfg, self.info
        self.fb = ConvolutionThenFilterbank(self.ctx, nsub, info.nchan, info.npol, info.ndim == 1, r, max_parts=cfg.max_parts,
                                            parts_per_block=cfg.parts_per_block)
        self.nkeep, self.nsamp_step, self.nsamp_overlap = self.fb.nkeep, self.fb.nsamp_step, self.fb.nsamp_overlap
        self.npol_out = 4 // cfg.ndim
        self.fold = FoldEngine(self.ctx)
        self.fold.set_shape(self.nchan_out, self.npol_out, cfg.ndim, cfg.nbin)
        self.scale8 = eight_bit_scale()
        self.layout = _lib.RAW_CASPSR if info.machine == "CASPSR" else _lib.RAW_GENERIC
        conv_rate = info.rate * (0.5 if info.ndim == 1 else 1.0)                           # Convolution.C:266-267
        self.out_rate = conv_rate / float(nsub)                                            # Filterbank.C:338-339
        self.out_start = info.start_seconds + r.impulse_pos / conv_rate                   # Convolution.C:300
        nsamp_fft = 2 * r.ndat if info.ndim == 1 else r.ndat
        self.scalefac = float(nsamp_fft) * float(r.ndat) * float(nsub)                    # Convolution.C:305, Filterbank.C:124-125
        self.sample_delay, self.sd_carried, self.sd_head, self.sd_short = None, 0, 0, 0
        self.detected = torch.empty((self.nchan_out, self.npol_out, cfg.parts_per_block * self.nkeep * cfg.ndim),
                                    dtype=torch.float32, device="cuda:%d" % device)
        self.fused_mode, self.fused_fold = 0, False
        self.optime, self.dumps, self._dump_cplx = {}, {}, None
        self.hits = np.zeros(cfg.nbin, dtype=np.uint32)
        self.integration_length, self.ndat_total = 0.0, 0
        self.nsamples_in, self.ndat_out, self.subints = 0, 0, []

    def _op(self, name, fn):
        """Operation::operate with record_time (Operation.C:90-113): wall time of the operation including its stream
        synchronisation, accumulated per name."""
        if not self.cfg.record_time:
            return fn()
        import time as _time
        self.ctx.synchronize()
        t0 = _time.perf_counter()
        r = fn()
        self.ctx.synchronize()
        e = self.optime.setdefault(name, [0.0, 0])
        e[0] += _time.perf_counter() - t0
        e[1] += 1
        return r

    def vitals(self):
        """The lines `dspsr` prints while it prepares (report_vitals, LoadToFold1.C:773-792,874-879): dedispersion filter
        length, what the filterbank requires, the block size."""
        r, cfg = self.response, self.cfg
        if r is None:                                     # Config::Never: no response
            from types import SimpleNamespace
            r = SimpleNamespace(ndat=1, minimum_ndat=0)
        nsamp_fft = self.nsamp_step + self.nsamp_overlap
        nblock = cfg.parts_per_block * self.nsamp_step + self.nsamp_overlap
        return ["dspsr: dedispersion filter length=%d (minimum=%d) complex samples" % (r.ndat, r.minimum_ndat),
                "dspsr: %d channel dedispersing filterbank requires %d samples" % (cfg.nchan, nsamp_fft),
                "dspsr: blocksize=%d samples or %g MB" % (nblock, self.block_bytes() / (1024.0 * 1024.0))]

    def report(self, file=None):
        """Operation::report (Operation.C:168-190): the table `dspsr -r` prints -- name, time spent, discarded weights."""
        import sys
        file = file or sys.stderr
        if not self.optime:
            return
        pad = lambda t: ("%-25s" % t)
        print(pad("Operation") + pad("Time Spent") + pad("Discarded"), file=file)
        for name, (t, _n) in self.optime.items():
            print(pad(name) + pad("%g" % t) + pad("0"), file=file)

    # bytes of one block of `npart` parts
    def block_bytes(self, npart=None):
        npart = npart or self.cfg.parts_per_block
        nsamp = npart * self.nsamp_step + self.nsamp_overlap
        if self.layout == _lib.RAW_CASPSR:
            return ((nsamp + 3) // 4) * 8          # whole 4-sample groups (4 B pol0 | 4 B pol1)
        return nsamp * self.in_nchan * self.info.npol * self.info.ndim

    def seek_block(self, k):
        """Time-slice replicas: the next block is block k of the stream (blocks of parts_per_block parts), not the one
        following the last (MultiThread.C:120-148 hands whole blocks to the threads in turn)."""
        self.ndat_out = k * self.cfg.parts_per_block * self.nkeep

    def _phase(self, t_seconds):
        """Fold::get_phi / get_pfold (Fold.C:943-958)."""
        if self.cfg.folding_period > 0:
            p = self.cfg.folding_period
            return math.fmod(t_seconds, p) / p - self.reference_phase, p
        day, sec = self.info.mjd_day, self.info.mjd_sec + t_seconds
        return self.polyco.phase_frac(day, sec) - self.reference_phase, 1.0 / self.polyco.frequency(day, sec)

    def process_block(self, raw, npart=None, events=None):
        """raw: device int8 tensor holding npart*nsamp_step + nsamp_overlap samples (the InputBuffering
        tail of the previous block already prepended, Filterbank.C:443-444)."""
        cfg = self.cfg
        npart = npart or cfg.parts_per_block
        if raw.numel() < self.block_bytes(npart):
            raise DspsrAmdError("dspsr_amd.LoadToFold.process_block: block holds %d bytes, %d needed"
                                % (raw.numel(), self.block_bytes(npart)))
        ndat = npart * self.nkeep
        state = _lib.STOKES if cfg.stokes else _lib.COHERENCE
        if "Detection" in self.dumps:                        # the filterbank's complex output: one extra pass, taps only
            if self._dump_cplx is None:
                self._dump_cplx = self.torch.empty((self.nchan_out, 2, 2 * cfg.parts_per_block * self.nkeep),
                                                   dtype=self.torch.float32, device="cuda:%d" % self.ctx.device)
            self.fb.perform_raw(raw, self.layout, self.scale8, self._dump_cplx, npart)
            self.dumps["Detection"].write(self._dump_cplx, ndat, 2)
        if self.sample_delay is not None:
            return self._process_block_interchan(raw, npart, ndat, state, events)
        # Subint<Fold>::transformation: fold piece by piece, emitting a sub-integration at every boundary
        pieces = self._pieces(ndat)
        if self.fused_fold and len(pieces) == 1 and pieces[0][:2] == (0, ndat):
            # one fold call covers the block: filterbank, detection and fold in one launch group.  Same sums in the
            # same order as the unfused chain below (blocks holding a sub-integration boundary take that chain).
            idat_start, ndat_fold, _division, complete = pieces[0]
            folded = self._set_plan(idat_start, ndat_fold)
            if events is not None:
                events[0].record()
            self._op("Filterbank+Detection+Fold", lambda: self.fb.perform_fold(self.fold, npart, state, raw=raw, layout=self.layout,
                                                                            scale=self.scale8))
            if events is not None:
                events[1].record()
            self.integration_length += folded / self.out_rate
            self.ndat_total += ndat_fold
            if complete:
                self.finish_subint(*self._subint_comm)
            self.ndat_out += ndat
            self.nsamples_in += npart * self.nsamp_step
            return
        if events is not None:          # HIP events bracketing the FFT+chirp launch group (bench.py roofline)
            events[0].record()
        self._op("Filterbank+Detection", lambda: self.fb.perform_detect(self.detected, npart, state, cfg.ndim, raw=raw,
                                                                        layout=self.layout, scale=self.scale8))
        if events is not None:
            events[1].record()
        if "Fold" in self.dumps:
            self.dumps["Fold"].write(self.detected, ndat, cfg.ndim)
        for idat_start, ndat_fold, _division, complete in pieces:
            self._fold_piece(idat_start, ndat_fold)
            if complete:
                self.finish_subint(*self._subint_comm)
        self.ndat_out += ndat
        self.nsamples_in += npart * self.nsamp_step

    def _process_block_interchan(self, raw, npart, ndat, state, events):
        """-K chain: filterbank+detect -> SampleDelay (in place, LoadToFold1.C:617-618) -> fold."""
        cfg, nd, head = self.cfg, self.cfg.ndim, self.sd_head
        if events is not None:
            events[0].record()
        self._op("Filterbank+Detection", lambda: self.fb.perform_detect(self.detected[:, :, head * nd:], npart, state, nd, raw=raw,
                                                                        layout=self.layout, scale=self.scale8))
        if events is not None:
            events[1].record()
        off, nin = head - self.sd_carried, self.sd_carried + ndat
        rows = self.detected[:, :, off * nd:(off + nin) * nd]
        nuse = nin - self.sd_short              # (sub-band rank: the band's total delay, not just this rank's, is given up)
        nout = self._op("SampleDelay", lambda: self.sample_delay.transform(rows[:, :, :nuse * nd].unflatten(2, (nuse, nd)))) \
            if nuse > 0 else 0
        if nout and "Fold" in self.dumps:
            self.dumps["Fold"].write(rows, nout, nd)
        if nout:
            for idat_start, ndat_fold, _division, complete in self._pieces(nout):
                folded = self._set_plan(idat_start, ndat_fold)
                self._op("Fold", lambda: self.fold.fold(rows))
                self.integration_length += folded / self.out_rate
                self.ndat_total += ndat_fold
                if complete:
                    self.finish_subint(*self._subint_comm)
        # InputBuffering::set_next_start(output_ndat): the unshifted tail goes in front of the next block
        carry = nin - nout
        if carry:
            copy_data_fpt(self.ctx, self.detected[:, :, (head - carry) * nd:head * nd], rows[:, :, nout * nd:])
        self.sd_carried = carry
        self.ndat_out += nout
        self.nsamples_in += npart * self.nsamp_step

    def _pieces(self, ndat):
        """Subint<Fold>::transformation (Subint.h:234-309): the pieces of the next `ndat` output samples, one per
        sub-integration they touch -- seconds mode (-L), turns mode (-s / -turns) or one piece."""
        cfg = self.cfg
        if cfg.subint_seconds > 0:
            return subint_pieces(self.ndat_out, ndat, cfg.subint_seconds, self.out_rate)
        if cfg.subint_turns > 0:
            if self._turns is None:
                if cfg.folding_period > 0:
                    p = cfg.folding_period
                    phase = lambda t: (int(math.floor(t / p)), t / p - math.floor(t / p))
                    iphase = lambda ph, guess: (ph[0] + ph[1]) * p
                    pg = p
                else:
                    day, s0 = self.info.mjd_day, self.info.mjd_sec
                    phase = lambda t: self.polyco.phase(day, s0 + t)
                    iphase = lambda ph, guess: self.polyco.iphase(ph, day, s0 + guess) - s0
                    pg = 1.0 / self.polyco.frequency(day, s0 + self.out_start)
                self._turns = TurnsDivider(phase, iphase, pg, self.out_start, self.out_rate, cfg.subint_turns,
                                           self.reference_phase, cfg.fractional_pulses)
            return self._turns.pieces(self.ndat_out, ndat)
        return [(0, ndat, 0, False)]

    _turns = None
    _subint_comm = (None, 0, 1, None)     # (dist, rank, world, gather_buffer[, replicas]) used at sub-integration dumps

    def set_communicator(self, dist, rank, world, gather_buffer=None, replicas=False):
        self._subint_comm = (dist, rank, world, gather_buffer, replicas)

    def _set_plan(self, idat_start, ndat_fold):
        """The host plan loop of Fold::fold (Fold.C:650-657,718-787): phase of the first sample, then the bins."""
        cfg = self.cfg
        t0 = self.out_start + (self.ndat_out + idat_start + 0.5) / self.out_rate     # midpoint of first sample
        phi, pfold = self._phase(t0)
        self.fold.set_nbin(cfg.nbin)
        self.fold.set_ndat(ndat_fold, idat_start)
        return self.fold.set_bins(phi, (1.0 / self.out_rate) / pfold, ndat_fold, idat_start, self.hits)

    def _fold_piece(self, idat_start, ndat_fold):
        """Fold::fold (Fold.C:650-657,718-803) on detected[idat_start : idat_start+ndat_fold]."""
        folded = self._set_plan(idat_start, ndat_fold)
        self._op("Fold", lambda: self.fold.fold(self.detected))                  # (no head room without -K)
        self.integration_length += folded / self.out_rate
        self.ndat_total += ndat_fold

    def profiles_tensor(self):
        """Zero-copy torch view of the device-resident PhaseSeries (Fold::Engine::get_profiles)."""
        torch = self.torch
        n = self.nchan_out * self.npol_out * self.cfg.nbin * self.cfg.ndim
        ptr = self.fold.get_profiles_ptr()

        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2,
                                      "strides": None}
        return torch.as_tensor(h, device="cuda:%d" % self.ctx.device)

    comm = None                 # dspsr_amd.Communicator (RCCL behind the C-ABI): the product's exchange
    _comm_pending = None
    copy_subints = True          # False: `subints` profiles stay views of the communicator's pinned buffer (see collect_subint)

    def set_rccl_communicator(self, comm):
        """The exchange of a multi-GPU run: a dspsr_amd.Communicator (dspsr_amd_comm_*, csrc/comm.hip -- the same C entry
        points DSPSR's host calls).  Without one, finish_subint falls back to the torch.distributed calls below, which
        exist for the gloo CPU tests and for rehearsing several ranks on one device (RCCL needs one GPU per rank)."""
        self.comm = comm

    def collect_subint(self, copy=None):
        """Wait for the exchange finish_subint(wait=False) started and append its result to `subints` (root only).
        copy=False (default: self.copy_subints, True): the entry's profile is a VIEW of the communicator's pinned buffer, valid
        until the next dump starts -- for a writer that consumes each sub-integration as it arrives."""
        if self._comm_pending is None:
            return
        if copy is None:
            copy = self.copy_subints
        check = self._comm_pending
        self._comm_pending = None
        prof, hits, length, ndat_total, same = self.comm.finish(copy=copy)
        if check and not same:
            raise DspsrAmdError("sub-band ranks disagree on hits[]: the shards are not sample aligned "
                                "(different nfilt_pos/neg, start time or rate)")
        if prof is not None:
            self.subints.append({"hits": hits, "integration_length": length, "ndat_total": ndat_total, "profile": prof})

    def finish_subint(self, dist=None, rank=0, world=1, gather_buffer=None, replicas=False, check_hits=True, wait=True):
        """Subint<Fold>: emit the finished sub-integration and zero the profile (Subint.h:291-303).  With world > 1 this
        is the one exchange of the path: sub-band shards deliver their slice of the full band to rank 0, hits /
        integration_length taken from rank 0 (identical everywhere; check_hits asserts it); time-slice replicas
        (replicas=True) SUM profiles, hits, integration_length and ndat_total.
        With an RCCL communicator (set_rccl_communicator) the exchange is dspsr_amd_reduce_profiles_start/finish: a
        snapshot on the compute stream, the collective on the communicator's stream -- the next block's kernels overlap
        it; wait=False leaves it in flight until collect_subint() or the next dump.
        check_hits (default True) adds a MIN/MAX all-reduce of hits[] to the sub-band exchange (two small collectives in the
        same group; nothing for replicas): pass False inside a timed loop that has checked once."""
        if self.comm is not None:
            # (one exchange in flight per communicator.  The result of the previous one is COPIED here even with
            #  copy_subints False: start() below may grow the pinned buffer a view would point into -- a view is only handed
            #  out when the caller collects the sub-integration itself)
            self.collect_subint(copy=True)
            n = self.npol_out * self.cfg.nbin * self.cfg.ndim           # floats per channel: rows are packed
            self.comm.start(self.comm.SUM if replicas else self.comm.GATHER, self.fold.get_profiles_ptr(), n, self.nchan_out, n,
                            self.hits, self.integration_length, self.ndat_total, root=0,
                            check_hits=check_hits and not replicas)
            self._comm_pending = bool(check_hits and not replicas)
            self.fold.zero()                                            # stream ordered behind the snapshot
            self.hits[:] = 0
            self.integration_length = 0.0
            self.ndat_total = 0
            if wait:
                self.collect_subint()
            return
        prof = self.profiles_tensor()
        if replicas:
            res = reduce_replicas(prof, self.hits, self.integration_length, self.ndat_total, dist, rank, world)
            if rank == 0:
                self.subints.append({"hits": res[1].copy(), "integration_length": res[2], "ndat_total": res[3],
                                     "profile_dev": res[0].clone()})
        else:
            if check_hits:
                check_identical_hits(self.hits, dist, rank, world)
            result = reduce_subbands(prof, dist, rank, world, gather_buffer)
            if rank == 0:
                self.subints.append({"hits": self.hits.copy(), "integration_length": self.integration_length,
                                     "ndat_total": self.ndat_total, "profile_dev": result.clone()})
        self.fold.zero()
        self.hits[:] = 0
        self.integration_length = 0.0
        self.ndat_total = 0

    def process_host_blocks(self, blocks, npart=None):
        """Host hand-over (the role of dsp::TransferCUDA / TransferBitSeriesCUDA, Signal/General/TransferCUDA.C:24-85):
        `blocks` yields one block each -- a pinned host int8 tensor (copied from where it lies; the caller must not touch
        it again before the call returns), or any other host int8 tensor / numpy array (staged through two pinned buffers
        owned by this call), optionally as a pair (block, npart) for a ragged last block.  Block i+1 is copied to the
        device on a second stream while block i is processed (two device buffers, events both ways).  With the data
        coming over PCIe Gen5 x16 the link, not the GPU, sets the rate.  Returns the number of
        blocks processed."""
        torch = self.torch
        main = torch.cuda.current_stream()
        copy = torch.cuda.Stream()
        bufs, pins = [None, None], [None, None]
        ready, done = [torch.cuda.Event(), torch.cuda.Event()], [torch.cuda.Event(), torch.cuda.Event()]
        used = [False, False]
        it = iter(blocks)

        def split(item):
            if item is None:
                return None, None
            if isinstance(item, tuple):
                return item[0], item[1]
            return item, npart

        def upload(k, host):
            if not (torch.is_tensor(host) and host.is_pinned()):
                src = (host.numpy() if torch.is_tensor(host) else np.asarray(host)).reshape(-1).view(np.int8)
                if used[k]:
                    ready[k].synchronize()                   # host wait: the previous copy out of this pinned buffer is over
                if pins[k] is None or pins[k].numel() < src.size:
                    pins[k] = torch.empty(src.size, dtype=torch.int8).pin_memory()
                host = pins[k][:src.size]
                host.numpy()[:] = src                        # (a read-only memory map is fine as a source)
            with torch.cuda.stream(copy):
                if bufs[k] is None or bufs[k].numel() < host.numel():
                    if bufs[k] is not None:
                        bufs[k].record_stream(main)          # still being read by kernels on the main stream
                    bufs[k] = torch.empty(host.numel(), dtype=torch.int8, device="cuda:%d" % self.ctx.device)
                    bufs[k].record_stream(main)
                elif used[k]:
                    copy.wait_event(done[k])                 # the kernels that read this buffer have finished
                bufs[k][:host.numel()].copy_(host, non_blocking=True)
                ready[k].record(copy)
            used[k] = True
            return host.numel()
        nxt, nxt_parts = split(next(it, None))
        if nxt is None:
            return 0
        sizes = [upload(0, nxt), 0]
        parts = [nxt_parts, None]
        n = 0
        while True:
            k = n & 1
            nxt, nxt_parts = split(next(it, None))
            if nxt is not None:
                sizes[k ^ 1], parts[k ^ 1] = upload(k ^ 1, nxt), nxt_parts
            main.wait_event(ready[k])
            self.process_block(bufs[k][:sizes[k]], parts[k])
            done[k].record(main)
            n += 1
            if nxt is None:
                break
        main.synchronize()                                   # the staging buffers go out of scope with this call
        return n

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        if self._comm_pending is not None:
            self.collect_subint()
        for d in self.dumps.values():
            d.close()
        if self.sample_delay is not None:
            self.sample_delay.close()
        self.fb.close()
        self.fold.close()
        self.ctx.close()


# ---- search mode: digifil (Signal/General/LoadToFil.C) ------------------------------------------------------------

@dataclass
class SearchConfig:
    """The digifil options on this path (LoadToFil.C:60-100 defaults, digifil.C)."""
    nchan: int = 4096                 # -F nchan  (non-convolving TFPFilterbank)
    tscrunch: int = 16                # -t
    nbit: int = 2                     # -b  (digifil default 2; 1, 2, 4, 8, 16, -32)
    rescale_seconds: float = 10.0     # -I  (0 disables Rescale and the digitizer's own scales, LoadToFil.C:318,360)
    rescale_constant: bool = False    # -c
    scale_fac: float = 1.0            # -s
    parts_per_block: int = 4096       # FFT blocks (2*nchan samples each) per call; a multiple of tscrunch
    # the convolving branch, `-F nchan:D` (LoadToFil.C:176-222): LoadToFilCoherent
    dispersion_measure: float = 0.0   # -D  (with -F N:D: coherent dedispersion inside the filterbank)
    freq_res: int = 0                 # -x  (0 => optimal, as dsp::Dedispersion chooses)
    fscrunch: int = 0                 # -f
    npol: int = 1                     # -d  output polarisations: 1 Intensity, 2 PPQQ, 4 Coherence (LoadToFil.C:262-277)
    dedisperse: bool = False          # -K  remove the inter-channel dispersion delays (dsp::SampleDelay, LoadToFil.C:236-247)
    max_parts: int = 32               # parts per launch group
    fused: bool = True                # detection + time scrunch inside the inverse pass where the geometry allows it


def write_sigproc_header(f, *, source_name="unknown", rawdatafile="unknown", machine_id=0, telescope_id=0, src_raj=0.0, src_dej=0.0,
                         fch1, foff, nchans, nbits, tstart_mjd, tsamp, nifs=1):
    """The SIGPROC filterbank header digifil writes (Kernel/Formats/sigproc/filterbank_header.c:42-105, send_stuff.c,
    values as SigProcObservation::unload fills them, SigProcObservation.C:228-270): length-prefixed keywords, native
    (little-endian) int32 / float64 values."""
    import struct

    def send_string(sv):
        b = sv.encode("ascii")
        f.write(struct.pack("<i", len(b)) + b)

    def send_int(name, v):
        send_string(name)
        f.write(struct.pack("<i", int(v)))

    def send_double(name, v):
        send_string(name)
        f.write(struct.pack("<d", float(v)))

    send_string("HEADER_START")
    if rawdatafile:
        send_string("rawdatafile")
        send_string(rawdatafile)
    if source_name:
        send_string("source_name")
        send_string(source_name)
    send_int("machine_id", machine_id)
    send_int("telescope_id", telescope_id)
    for name, v in (("src_raj", src_raj), ("src_dej", src_dej), ("az_start", 0.0), ("za_start", 0.0)):   # send_coords
        send_double(name, v)
    send_int("data_type", 1)
    send_double("fch1", fch1)
    send_double("foff", foff)
    send_int("nchans", nchans)
    send_int("nbeams", 0)
    send_int("ibeam", 0)
    send_int("nbits", nbits)
    send_double("tstart", tstart_mjd)
    send_double("tsamp", tsamp)
    send_int("nifs", nifs)
    send_string("HEADER_END")


class LoadToFil:
    """digifil's chain for 8-bit real dual-polarisation input and one output polarisation, every stage on the device
    (LoadToFil.C:196-362): TFPFilterbank (PPQQ) + TScrunch [one kernel] -> Rescale (per pol and channel, in place)
    -> PScrunch -> SigProcDigitizer.  process_block returns the packed block (device uint8, [time][chan] n-bit)."""

    fused_output = True          # Rescale + PScrunch + digitiser as one pass (False: the three operations one after the other)

    def __init__(self, cfg: SearchConfig, info: InputInfo, device: int = 0, stream: int | None = None):
        import torch
        self.torch = torch
        self.cfg, self.info = cfg, info
        if info.npol != 2 or info.ndim != 1 or info.nchan != 1:
            raise DspsrAmdError("dspsr_amd.LoadToFil: 8-bit real dual-polarisation single-channel input only")
        if cfg.parts_per_block % cfg.tscrunch:
            raise DspsrAmdError("dspsr_amd.LoadToFil: parts_per_block=%d is not a multiple of tscrunch=%d"
                                % (cfg.parts_per_block, cfg.tscrunch))
        self.ctx = Context(device, stream)
        self.scale8 = eight_bit_scale()
        self.layout = _lib.RAW_CASPSR if info.machine == "CASPSR" else _lib.RAW_GENERIC
        self.out_rate = info.rate / (2.0 * cfg.nchan) / cfg.tscrunch           # TFPFilterbank + TScrunch
        nout = cfg.parts_per_block // cfg.tscrunch
        dev = "cuda:%d" % device
        self.detected = torch.empty((nout, cfg.nchan, 2), dtype=torch.float32, device=dev)      # PPQQ, TFP order
        self.intensity = torch.empty((nout, cfg.nchan, 1), dtype=torch.float32, device=dev)
        self.rescale = None
        if cfg.rescale_seconds:
            interval = int(cfg.rescale_seconds * self.out_rate)                   # Rescale::init, Rescale.C:102-103
            if not interval:
                raise DspsrAmdError("dsp::Rescale::init nsample == 0")
            self.rescale = Rescale(self.ctx, cfg.nchan, 2, interval, cfg.rescale_constant)
        nbits = 32 if cfg.nbit == -32 else cfg.nbit
        self.bytes_per_sample = cfg.nchan * nbits // 8
        self.packed = torch.empty(nout * self.bytes_per_sample, dtype=torch.uint8, device=dev)
        self.ndat_out = 0

    def block_bytes(self, npart=None):
        return (npart or self.cfg.parts_per_block) * 2 * self.cfg.nchan * 2

    def process_block(self, raw, npart=None):
        cfg = self.cfg
        npart = npart or cfg.parts_per_block
        if npart % cfg.tscrunch or npart > cfg.parts_per_block:
            raise DspsrAmdError("dspsr_amd.LoadToFil.process_block: npart=%d must be a multiple of tscrunch=%d and <= %d"
                                % (npart, cfg.tscrunch, cfg.parts_per_block))
        if raw.numel() < self.block_bytes(npart):
            raise DspsrAmdError("dspsr_amd.LoadToFil.process_block: block holds %d bytes, %d needed"
                                % (raw.numel(), self.block_bytes(npart)))
        nout = npart // cfg.tscrunch
        det = self.detected[:nout]
        tfp_filterbank(self.ctx, raw, cfg.nchan, npart, det, False, cfg.tscrunch, self.layout, self.scale8)
        packed = self.packed[:nout * self.bytes_per_sample]
        if self.rescale is not None and self.fused_output and cfg.nbit in (1, 2, 4, 8, 16):
            # Rescale -> PScrunch -> digitiser in one pass over the PPQQ block (identical bytes, LoadToFil.C:318-362)
            self.rescale.pscrunch_digitize(det, packed, cfg.nbit, cfg.scale_fac, flip_band=self.info.bandwidth > 0, swap_band=False)
            self.ndat_out += nout
            return packed
        if self.rescale is not None:
            self.rescale.transform(det)                                            # in place, LoadToFil.C:325-326
        inten = self.intensity[:nout]
        pscrunch_tfp(self.ctx, det, inten, cfg.nchan, 2)                            # LoadToFil.C:333-343
        # after Rescale the input scale is 1 (Rescale.C:204); without it the TFP filterbank leaves scale 1 as well
        sigproc_digitize(self.ctx, inten, packed, cfg.nchan, 1, cfg.nbit, use_digi_scales=self.rescale is not None,
                         input_scale=1.0, scale_fac=cfg.scale_fac, flip_band=self.info.bandwidth > 0, swap_band=False)
        self.ndat_out += nout
        return packed

    def header_values(self):
        """fch1/foff/tsamp/tstart as SigProcObservation::unload derives them from the digitizer's output observation
        (bandwidth forced negative, SigProcDigitizer.C:83-85; channel 0 centred half a channel inside the band edge)."""
        nchan, bw = self.cfg.nchan, -abs(self.info.bandwidth)
        fch1 = self.info.centre_frequency - 0.5 * bw + 0.5 * bw / nchan
        nbits = 32 if self.cfg.nbit == -32 else self.cfg.nbit
        return dict(fch1=fch1, foff=bw / nchan, nchans=nchan, nbits=nbits, tsamp=1.0 / self.out_rate,
                    tstart_mjd=self.info.mjd_day + (self.info.mjd_sec + self.info.start_seconds) / 86400.0, nifs=1)

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        if self.rescale is not None:
            self.rescale.close()
        self.ctx.close()


class LoadToFilCoherent:
    """digifil with the convolving filterbank, `digifil -F N:D [-x M] [-f F] -t T -b nbit` (Signal/General/LoadToFil.C:176-222,250-362):
    dsp::Filterbank with the Dedispersion response -> Detection::square_law (Intensity / PPQQ: no PScrunch behind it, :281) ->
    [FScrunch] -> TScrunch -> Rescale -> SigProcDigitizer, FPT order throughout, every stage on the device.  One launch group runs
    filterbank, detection and time scrunch (FilterbankEngine.perform_search): the detected rows never exist at the filterbank's
    output rate.  process_block(raw) takes the 8-bit block of LoadToFold (npart * nsamp_step + nsamp_overlap samples, the overlap
    re-presented by the caller) and returns the packed bytes [time][pol][chan] of the output samples completed by it."""

    def __init__(self, cfg: SearchConfig, info: InputInfo, device: int = 0, stream: int | None = None):
        import torch
        self.torch = torch
        self.cfg, self.info = cfg, info
        if cfg.npol not in (1, 2, 4):
            raise DspsrAmdError("dspsr_amd.LoadToFilCoherent: npol=%d (Intensity 1 / PPQQ 2 / Coherence 4 are built; NthPower 3 is not)" % cfg.npol)
        if cfg.npol >= 2 and info.npol != 2:
            raise DspsrAmdError("dspsr_amd.LoadToFilCoherent: PPQQ / Coherence need two input polarisations")
        if cfg.nchan % info.nchan:
            raise DspsrAmdError("dsp::Filterbank::make_preparations output nchan=%d not a multiple of input nchan=%d" % (cfg.nchan, info.nchan))
        if cfg.dispersion_measure == 0.0 and not cfg.freq_res and cfg.npol <= 2:
            raise DspsrAmdError("dspsr_amd.LoadToFilCoherent: -F N:D with a dispersion measure, -F N -x M, or -d 4 (with none of them "
                                "digifil takes the TFPFilterbank: dspsr_amd.LoadToFil)")
        self.ctx = Context(device, stream)
        nsub = cfg.nchan // info.nchan
        if cfg.dispersion_measure != 0.0:
            r = Dedispersion(info.centre_frequency, info.bandwidth, cfg.dispersion_measure, input_nchan=info.nchan, ndim=info.ndim)
            if cfg.freq_res:
                r.set_frequency_resolution(cfg.freq_res)                               # LoadToFil.C:190-191
            r.match(cfg.nchan)
            kernel = r.kernel
        else:
            # -F N -x M without :D (LoadToFil.C:199-216): the convolving filterbank with no response -- nothing is discarded
            # (Filterbank.C:139-155: freq_res as set, nfilt 0)
            # -F N -d 4 with neither (LoadToFil.C:205-207: `npol > 2` takes the Filterbank too): its default freq_res = 1, the
            # non-convolving filterbank (Filterbank.C:614-623)
            from types import SimpleNamespace
            r, kernel = SimpleNamespace(ndat=cfg.freq_res or 1, impulse_pos=0, impulse_neg=0, kernel=None), None
        self.response = r
        self.fb = FilterbankEngine(self.ctx).setup(nsub, r.ndat, r.impulse_pos, r.impulse_neg, info.nchan, info.npol, info.ndim == 1,
                                                   kernel, max_parts=cfg.max_parts)
        self.nkeep, self.nsamp_step, self.nsamp_overlap = self.fb.nkeep, self.fb.nsamp_step, self.fb.nsamp_overlap
        n_fft = nsub * r.ndat
        nsamp_fft = 2 * n_fft if info.ndim == 1 else n_fft
        self.fb_rate = info.rate * (float(r.ndat) / float(nsamp_fft))
        ts = max(1, cfg.tscrunch)
        self.out_rate = self.fb_rate / ts
        self.out_start = info.start_seconds + r.impulse_pos / self.fb_rate
        self.scale8 = eight_bit_scale()
        self.layout = _lib.RAW_CASPSR if info.machine == "CASPSR" else _lib.RAW_GENERIC
        self.state = {1: _lib.INTENSITY, 2: _lib.PPQQ, 4: _lib.COHERENCE}[cfg.npol]
        self.nchan_out = cfg.nchan // cfg.fscrunch if cfg.fscrunch else cfg.nchan
        if cfg.fscrunch and cfg.nchan % cfg.fscrunch:
            raise DspsrAmdError("dspsr_amd.LoadToFilCoherent: nchan=%d is not a multiple of fscrunch=%d" % (cfg.nchan, cfg.fscrunch))
        dev = "cuda:%d" % device
        nmax = (ts - 1 + cfg.parts_per_block * self.nkeep) // ts
        # -K: dsp::SampleDelay sits between the filterbank and Detection (LoadToFil.C:236-247).  Detection acts sample by sample,
        # so delaying the detected rows gives the reference's numbers; the last total_delay samples of a block are re-presented in
        # front of the next one (InputBuffering, SampleDelay.C:117,146): they live in the head room in front of `detected`.
        self.sample_delay, self.sd_carried, head = None, 0, 0
        if cfg.dedisperse:
            dual = info.ndim == 2                              # Observation.C:80-87; Filterbank.C:358-364
            delays = dedispersion_sample_delays(info.centre_frequency, info.bandwidth, cfg.dispersion_measure, cfg.nchan,
                                                self.fb_rate, swap=dual and info.nchan == 1,
                                                nsub_swap=info.nchan if dual and info.nchan > 1 else 0)
            self.sample_delay = SampleDelay(self.ctx, delays, cfg.npol)
            head = self.sample_delay.total_delay
            if head > cfg.parts_per_block * self.nkeep:
                raise DspsrAmdError("dspsr_amd.LoadToFilCoherent: inter-channel delay of %d samples exceeds the block of %d"
                                    % (head, cfg.parts_per_block * self.nkeep))
            self.out_start += self.sample_delay.zero_delay / self.fb_rate               # SampleDelay.C:159
        self.sd_head = head
        # FScrunch sits between Detection and TScrunch (LoadToFil.C:286-304) and SampleDelay in front of Detection: the fused launch
        # group holds neither, nor the four Coherence products, so with -f, -K or -d 4 the operations run one after the other
        # (perform_search at tscrunch 1 = filterbank + detection; Coherence: perform_detect with ndim 1)
        self.fused = cfg.fused and not cfg.fscrunch and not cfg.dedisperse and cfg.npol != 4
        nmax = (ts - 1 + head + cfg.parts_per_block * self.nkeep) // ts
        self.scrunched = torch.empty((self.nchan_out, cfg.npol, max(1, nmax)), dtype=torch.float32, device=dev)
        self.carry = torch.zeros((self.nchan_out, cfg.npol), dtype=torch.float32, device=dev)
        self.carry_count = 0
        self.detected = None
        if not self.fused:
            nd = head + cfg.parts_per_block * self.nkeep
            self.detected = torch.empty((cfg.nchan, cfg.npol, nd), dtype=torch.float32, device=dev)
            self.det_carry = torch.zeros((cfg.nchan, cfg.npol), dtype=torch.float32, device=dev)
            self.fscr = torch.empty((self.nchan_out, cfg.npol, nd), dtype=torch.float32, device=dev) if cfg.fscrunch else None
        self.rescale = None
        if cfg.rescale_seconds:
            interval = int(cfg.rescale_seconds * self.out_rate)                          # Rescale::init, Rescale.C:102-103
            if not interval:
                raise DspsrAmdError("dsp::Rescale::init nsample == 0")
            self.rescale = Rescale(self.ctx, self.nchan_out, cfg.npol, interval, cfg.rescale_constant)
        nbits = 32 if cfg.nbit == -32 else cfg.nbit
        self.bytes_per_sample = self.nchan_out * cfg.npol * nbits // 8
        self.packed = torch.empty(max(1, nmax) * self.bytes_per_sample, dtype=torch.uint8, device=dev)
        self.ndat_out = 0
        # Filterbank.C:124-128 leaves scale = n_fft * freq_res on its output; FScrunch / TScrunch multiply it by their factors
        # (TimeSeries::rescale, TScrunch.C:126, FScrunch.C:103); without Rescale the digitiser divides by it (SigProcDigitizer.C:158)
        self.input_scale = float(n_fft) * float(r.ndat) * ts * (cfg.fscrunch or 1)

    def block_bytes(self, npart=None):
        npart = npart or self.cfg.parts_per_block
        nsamp = npart * self.nsamp_step + self.nsamp_overlap
        if self.layout == _lib.RAW_CASPSR:
            return ((nsamp + 3) // 4) * 8
        return nsamp * self.info.nchan * self.info.npol * self.info.ndim

    def detect_scrunch(self, raw, npart=None):
        """Filterbank -> Detection -> [FScrunch] -> TScrunch of one block: the scrunched rows [nchan_out][npol][nout] (a view)."""
        cfg = self.cfg
        npart = npart or cfg.parts_per_block
        if raw.numel() < self.block_bytes(npart):
            raise DspsrAmdError("dspsr_amd.LoadToFilCoherent.process_block: block holds %d bytes, %d needed" % (raw.numel(), self.block_bytes(npart)))
        ts = max(1, cfg.tscrunch)
        if self.fused:
            nout, self.carry_count = self.fb.perform_search(self.scrunched, self.carry, self.carry_count, npart, ts, self.state, raw=raw,
                                                            layout=self.layout, scale=self.scale8)
            return self.scrunched[:, :, :nout]
        nd, head = npart * self.nkeep, self.sd_head
        det = self.detected[:, :, head:head + nd]
        if cfg.npol == 4:
            self.fb.perform_detect(det, npart, self.state, 1, raw=raw, layout=self.layout, scale=self.scale8)
        else:
            self.fb.perform_search(det, self.det_carry, 0, npart, 1, self.state, raw=raw, layout=self.layout, scale=self.scale8)
        rows = carry = None
        if self.sample_delay is not None:
            off, nin = head - self.sd_carried, self.sd_carried + nd
            rows = self.detected[:, :, off:off + nin]
            nd = self.sample_delay.transform(rows)                                   # in place (LoadToFil.C:240-241)
            carry = nin - nd                 # InputBuffering::set_next_start: the unshifted tail goes in front of the next block
            det = rows[:, :, :nd]
        nout = 0
        if nd:
            if cfg.fscrunch:
                det = fscrunch_fpt(self.ctx, det, self.fscr[:, :, :nd], cfg.fscrunch)
            nout, self.carry_count = tscrunch_fpt(self.ctx, det, self.scrunched, ts, self.carry, self.carry_count)
        if rows is not None:
            if carry:
                copy_data_fpt(self.ctx, self.detected[:, :, head - carry:head], rows[:, :, nd:])
            self.sd_carried = carry
        return self.scrunched[:, :, :nout]

    def process_block(self, raw, npart=None):
        cfg = self.cfg
        scr = self.detect_scrunch(raw, npart)
        nout = scr.shape[2]
        packed = self.packed[:nout * self.bytes_per_sample]
        flip = self.info.bandwidth > 0
        if nout:
            if self.rescale is not None and cfg.nbit != -32:
                self.rescale.digitize_fpt(scr, packed, cfg.nbit, cfg.scale_fac, flip_band=flip)        # Rescale + digitiser, one pass
            else:
                if self.rescale is not None:
                    self.rescale.transform_fpt(scr)                                                     # in place, LoadToFil.C:312-313
                sigproc_digitize_fpt(self.ctx, scr, packed, cfg.nbit, use_digi_scales=self.rescale is not None,
                                     input_scale=1.0 if self.rescale is not None else self.input_scale, scale_fac=cfg.scale_fac,
                                     flip_band=flip)
        self.ndat_out += nout
        return packed

    def header_values(self):
        nchan, bw = self.nchan_out, -abs(self.info.bandwidth)
        fch1 = self.info.centre_frequency - 0.5 * bw + 0.5 * bw / nchan
        nbits = 32 if self.cfg.nbit == -32 else self.cfg.nbit
        return dict(fch1=fch1, foff=bw / nchan, nchans=nchan, nbits=nbits, tsamp=1.0 / self.out_rate,
                    tstart_mjd=self.info.mjd_day + (self.info.mjd_sec + self.out_start) / 86400.0, nifs=self.cfg.npol)

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        if self.rescale is not None:
            self.rescale.close()
        self.fb.close()
        self.ctx.close()
