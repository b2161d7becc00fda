"""Host-side mirror of the reference's Engine plug-in interfaces, over the C-ABI.

Names, argument meaning and error behaviour follow the reference classes:
  dsp::Filterbank::Engine  (Signal/General/dsp/FilterbankEngine.h:15-44): setup / perform / finish
  dsp::Detection::Engine   (Signal/General/dsp/Detection.h:98-106): polarimetry / square_law
  dsp::Fold::Engine        (Signal/Pulsar/dsp/Fold.h:249-312): set_nbin / set_ndat / set_bin / fold / synch / zero
  dsp::Dedispersion        (Signal/General/Dedispersion.C): prepare / build / match (host side)
Errors the reference throws as `Error` surface as DspsrAmdError with the same message text.

torch is used only to own device memory and streams; every computation goes through
libdspsr_amd.so (hand-written HIP).  TimeSeries are torch tensors in FPT order:
  voltages   float32 [nchan][npol][ndat*ndim]
  detected   float32 [nchan][npol_out][ndat*ndim_out]
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import lib


class DspsrAmdError(RuntimeError):
    """Counterpart of the reference's `Error` exception."""


def _check(ctx_handle, code, where):
    if code != 0:
        msg = lib.dspsr_amd_last_error(ctx_handle).decode() if ctx_handle else ""
        raise DspsrAmdError("%s failed (%d): %s" % (where, code, msg))


class Context:
    """One per pipeline thread/GPU, bound to one HIP stream (SingleThread.C:213-290)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        # stream: a hipStream_t handle as int (0 = the default stream torch uses), None = context-owned stream
        sp = C.c_void_p(-1 & (2 ** 64 - 1)) if stream is None else C.c_void_p(int(stream))
        code = lib.dspsr_amd_ctx_create(int(device), sp, C.byref(h))
        if code != 0:
            raise DspsrAmdError("dspsr_amd_ctx_create(device=%d) failed (%d): no usable HIP device" % (device, code))
        self.handle = h
        self.device = device

    def synchronize(self):
        _check(self.handle, lib.dspsr_amd_stream_sync(self.handle), "dspsr_amd_stream_sync")

    def close(self):
        if self.handle:
            lib.dspsr_amd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fold_binplan_runs(phi: float, phase_per_sample: float, nbin: int, ndat: int, cap: int = 0):
    """The plan of fold_binplan as runs (first sample, bin, samples), found run by run (csrc/host_prep.cpp fold_plan_run); hits[nbin]."""
    cap = cap or max(1, ndat)
    off, rb, rh = np.empty(cap, dtype=np.uint64), np.empty(cap, dtype=np.uint32), np.empty(cap, dtype=np.uint64)
    hits = np.zeros(nbin, dtype=np.uint32)
    n = C.c_uint64()
    code = lib.dspsr_amd_fold_binplan_runs(phi, phase_per_sample, nbin, ndat, off.ctypes.data_as(C.c_void_p), rb.ctypes.data_as(C.c_void_p),
                                           rh.ctypes.data_as(C.c_void_p), cap, C.byref(n), hits.ctypes.data_as(C.c_void_p))
    if code != 0:
        raise DspsrAmdError("dspsr_amd_fold_binplan_runs failed (%d)" % code)
    k = min(n.value, cap)
    return off[:k].copy(), rb[:k].copy(), rh[:k].copy(), hits, n.value


# ----------------------------------------------------------------------------------------------
# host-side preparation
# ----------------------------------------------------------------------------------------------

JA98_SPACING_8BIT = 0.02957   # JenetAnderson98::get_optimal_spacing(8) (PSRCHIVE, ext)


def eight_bit_scale(spacing: float = JA98_SPACING_8BIT) -> float:
    return float(np.float32(lib.dspsr_amd_eight_bit_scale(spacing)))


def optimal_fft_length(nbadperfft: int, nfft_max: int = 0) -> int:
    n = lib.dspsr_amd_optimal_fft_length(nbadperfft, nfft_max)
    return -1 if n == 2 ** 64 - 1 else int(n)


class Dedispersion:
    """dsp::Dedispersion: smearing -> impulse_pos/neg, frequency resolution, chirp (host, double->float)."""

    def __init__(self, centre_frequency, bandwidth, dispersion_measure, input_nchan=1, ndim=1,
                 dual_sideband=-1, dc_centred=False, swap=False, fractional_delay=False):
        self.cfg = _lib.DedispersionConfig(centre_frequency, bandwidth, dispersion_measure, input_nchan, input_nchan,
                                           ndim, dual_sideband, int(dc_centred), int(swap), 0, 0, int(fractional_delay))
        self.impulse_pos = self.impulse_neg = self.ndat = self.minimum_ndat = 0
        self.nchan = input_nchan
        self.kernel = None

    def set_frequency_resolution(self, nfft: int):
        self.cfg.freq_res = int(nfft)

    def set_maximum_ndat(self, n: int):
        self.cfg.ndat_max = int(n)

    def match(self, nchan: int):
        """Dedispersion::match(input, channels): prepare + build + Response::match ordering."""
        self.cfg.nchan = int(nchan)
        self.nchan = int(nchan)
        info = _lib.DedispersionInfo()
        err = C.create_string_buffer(256)
        code = lib.dspsr_amd_dedispersion_prepare(C.byref(self.cfg), C.byref(info), err, 256)
        if code != 0:
            raise DspsrAmdError(err.value.decode() or "dspsr_amd_dedispersion_prepare failed (%d)" % code)
        self.impulse_pos, self.impulse_neg = info.impulse_pos, info.impulse_neg
        self.minimum_ndat, self.ndat = info.minimum_ndat, info.ndat
        k = np.empty(self.nchan * self.ndat, dtype=np.complex64)
        code = lib.dspsr_amd_dedispersion_build(C.byref(self.cfg), self.ndat, k.ctypes.data_as(C.c_void_p))
        if code != 0:
            raise DspsrAmdError("dspsr_amd_dedispersion_build failed (%d)" % code)
        self.kernel = k
        return self


def fold_binplan(phi: float, phase_per_sample: float, nbin: int, ndat: int):
    plan = np.empty(ndat, dtype=np.uint32)
    hits = np.zeros(nbin, dtype=np.uint32)
    code = lib.dspsr_amd_fold_binplan(phi, phase_per_sample, nbin, ndat, plan.ctypes.data_as(C.c_void_p),
                                      hits.ctypes.data_as(C.c_void_p))
    if code != 0:
        raise DspsrAmdError("dspsr_amd_fold_binplan failed (%d)" % code)
    return plan, hits


# ----------------------------------------------------------------------------------------------
# engines
# ----------------------------------------------------------------------------------------------

def _strides3(t):
    """(chan_stride, pol_stride) in elements of a [nchan][npol][n] tensor with contiguous last dim."""
    assert t.dim() == 3 and t.stride(2) == 1, "TimeSeries rows must be contiguous"
    return t.stride(0), t.stride(1)


class FilterbankEngine:
    """dsp::Filterbank::Engine.  setup() takes what CUDA::FilterbankEngine::setup reads from the
    Filterbank (FilterbankCUDA.cu:73-168)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.handle = None

    def setup(self, nchan_subband, freq_res, nfilt_pos, nfilt_neg, input_nchan=1, npol=2, real_input=True,
              kernel: np.ndarray | None = None, max_parts: int = 1, force_four_pass: bool | int = False,
              fused_fold: int = _lib.FUSED_AUTO):
        """force_four_pass: False / 0 = passes chosen from the geometry, True / 1 = two-pass inverse everywhere, 2 = never the
        two-pass path of short responses (dspsr_amd_filterbank_config::force_four_pass)."""
        self.close()
        cfg = _lib.FilterbankConfig(nchan_subband, freq_res, nfilt_pos, nfilt_neg, input_nchan, npol,
                                    1 if real_input else 0, max_parts, int(force_four_pass), fused_fold)
        h = C.c_void_p()
        _check(self.ctx.handle, lib.dspsr_amd_filterbank_create(self.ctx.handle, C.byref(cfg), C.byref(h)),
               "dspsr_amd_filterbank_create")
        self.handle = h
        self.cfg = cfg
        if kernel is not None:
            k = np.ascontiguousarray(kernel, dtype=np.complex64)
            _check(self.ctx.handle, lib.dspsr_amd_filterbank_set_kernel(h, k.ctypes.data_as(C.c_void_p), k.size),
                   "dspsr_amd_filterbank_set_kernel")
        else:
            _check(self.ctx.handle, lib.dspsr_amd_filterbank_set_kernel(h, None, 0), "dspsr_amd_filterbank_set_kernel")
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        lib.dspsr_amd_filterbank_sizes(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        self.nsamp_fft, self.nsamp_overlap, self.nsamp_step, self.nkeep = a.value, b.value, c.value, d.value
        return self

    def _need(self, what, have, need):
        if have < need:
            raise DspsrAmdError("dspsr_amd.FilterbankEngine: %s holds %d elements per row, %d needed" % (what, have, need))

    def _raw_bytes(self, npart, layout=_lib.RAW_GENERIC):
        c = self.cfg
        nsamp = npart * self.nsamp_step + self.nsamp_overlap
        if layout == _lib.RAW_CASPSR:
            # 4 samples of pol0 then 4 of pol1: the block must hold WHOLE 8-byte groups (its last samples' pol1 bytes lie
            # up to 4 bytes behind their pol0 bytes; CASPSRUnpacker.C:132-187 unpacks group by group)
            return ((nsamp + 3) // 4) * 8
        return nsamp * c.input_nchan * c.npol * (1 if c.real_input else 2)

    def perform(self, inp, out, npart, in_step, out_step):
        ics, ips = _strides3(inp)
        if npart:
            ndim = 1 if self.cfg.real_input else 2
            self._need("input", inp.shape[2], (npart - 1) * in_step + self.nsamp_fft * ndim)
            if out is not None:
                self._need("output", out.shape[2], (npart - 1) * out_step + 2 * self.nkeep)
        if out is not None:
            ocs, ops = _strides3(out)
            optr = out.data_ptr()
        else:
            ocs = ops = 0
            optr = None
        _check(self.ctx.handle,
               lib.dspsr_amd_filterbank_perform(self.handle, inp.data_ptr(), ics, ips, optr, ocs, ops, npart, in_step,
                                                out_step), "dspsr_amd_filterbank_perform")

    def perform_raw(self, raw, layout, scale, out, npart, out_step=None):
        ocs, ops = _strides3(out) if out is not None else (0, 0)
        if out_step is None:
            out_step = 2 * self.nkeep
        if npart:
            if layout != _lib.RAW_UWB16:
                self._need("raw block", raw.numel(), self._raw_bytes(npart, layout))
            if out is not None:
                self._need("output", out.shape[2], (npart - 1) * out_step + 2 * self.nkeep)
        _check(self.ctx.handle,
               lib.dspsr_amd_filterbank_perform_raw(self.handle, raw.data_ptr(), layout, scale,
                                                    out.data_ptr() if out is not None else None, ocs, ops, npart,
                                                    out_step), "dspsr_amd_filterbank_perform_raw")

    def perform_detect(self, det, npart, state=_lib.COHERENCE, ndim=4, inp=None, in_step=0, raw=None,
                       layout=_lib.RAW_GENERIC, scale=1.0):
        dcs, dps = _strides3(det)
        if npart:
            self._need("detected block", det.shape[2], npart * self.nkeep * ndim)
            if det.shape[1] != 4 // ndim:
                raise DspsrAmdError("dspsr_amd.FilterbankEngine.perform_detect: ndim=%d needs %d planes, the block has %d"
                                    % (ndim, 4 // ndim, det.shape[1]))
            if raw is not None and layout != _lib.RAW_UWB16:
                self._need("raw block", raw.numel(), self._raw_bytes(npart, layout))
            if inp is not None:
                self._need("input", inp.shape[2], (npart - 1) * in_step + self.nsamp_fft * (1 if self.cfg.real_input else 2))
        if inp is not None:
            ics, ips = _strides3(inp)
            iptr = inp.data_ptr()
        else:
            ics = ips = 0
            iptr = None
        _check(self.ctx.handle,
               lib.dspsr_amd_filterbank_perform_detect(self.handle, iptr, ics, ips, in_step,
                                                       raw.data_ptr() if raw is not None else None, layout, scale,
                                                       state, ndim, det.data_ptr(), dcs, dps, npart),
               "dspsr_amd_filterbank_perform_detect")

    def perform_search(self, out, carry, carry_count, npart, tscrunch, state=_lib.INTENSITY, inp=None, in_step=0, raw=None,
                       layout=_lib.RAW_GENERIC, scale=1.0):
        """digifil's convolving branch in one launch group (LoadToFil.C:185-222,250-304): Filterbank -> Detection::square_law
        (Intensity / PPQQ) -> TScrunch on the detected stream.  out: device float32 rows [nchan][npol_out][>= nout]; carry:
        device float32 [nchan][npol_out] (the open output sample's partial sums); carry_count: samples already in it.
        Returns (nout, carry_count_after)."""
        ocs, ops = _strides3(out)
        npo = 2 if state == _lib.PPQQ else 1
        if out.shape[1] != npo or carry.numel() < out.shape[0] * npo:
            raise DspsrAmdError("dspsr_amd.FilterbankEngine.perform_search: out needs %d polarisation rows per channel and carry "
                                "[nchan][%d] floats" % (npo, npo))
        if npart:
            self._need("scrunched block", out.shape[2], (carry_count + npart * self.nkeep) // max(1, tscrunch))
            if raw is not None and layout != _lib.RAW_UWB16:
                self._need("raw block", raw.numel(), self._raw_bytes(npart, layout))
        if inp is not None:
            ics, ips = _strides3(inp)
            iptr = inp.data_ptr()
        else:
            ics = ips = 0
            iptr = None
        cc, nout = C.c_uint32(carry_count), C.c_uint64(0)
        _check(self.ctx.handle,
               lib.dspsr_amd_filterbank_perform_search(self.handle, iptr, ics, ips, in_step, raw.data_ptr() if raw is not None else None,
                                                       layout, scale, state, tscrunch, out.data_ptr(), ocs, ops, carry.data_ptr(),
                                                       C.byref(cc), npart, C.byref(nout)), "dspsr_amd_filterbank_perform_search")
        return int(nout.value), int(cc.value)

    def search_is_fused(self) -> bool:
        """perform_search runs detection and the time scrunch inside the inverse pass (else: separate launches, same numbers)."""
        return bool(lib.dspsr_amd_filterbank_search_is_fused(self.handle))

    def npass(self, raw_input: bool = True) -> int:
        """Transform passes of a call: 2 (short responses on the 8-bit block), 3, or 4 (two-pass inverse)."""
        return int(lib.dspsr_amd_filterbank_npass(self.handle, 1 if raw_input else 0))

    def fold_is_fused(self) -> int:
        """0: perform_fold runs Detection + Fold launches; 1: it folds inside the last filterbank pass with exact time-order
        sums (one workgroup per channel tile); 2: it folds inside the last pass with the parts of a launch cut into runs
        (geometries with fewer tiles than compute units): sums re-associated per run, equal to float rounding."""
        return int(lib.dspsr_amd_filterbank_fold_is_fused(self.handle))

    def perform_fold(self, fold, npart, state=_lib.COHERENCE, inp=None, in_step=0, raw=None,
                     layout=_lib.RAW_GENERIC, scale=1.0):
        """Fused filterbank -> detection (ndim 4) -> fold into `fold`'s device profile; the bin plan of the
        npart*nkeep output samples must already have been given to `fold` (set_nbin/set_ndat/set_bins)."""
        if npart:
            if raw is not None and layout != _lib.RAW_UWB16:
                self._need("raw block", raw.numel(), self._raw_bytes(npart, layout))
            if inp is not None:
                self._need("input", inp.shape[2], (npart - 1) * in_step + self.nsamp_fft * (1 if self.cfg.real_input else 2))
        if inp is not None:
            ics, ips = _strides3(inp)
            iptr = inp.data_ptr()
        else:
            ics = ips = 0
            iptr = None
        _check(self.ctx.handle,
               lib.dspsr_amd_filterbank_perform_fold(self.handle, iptr, ics, ips, in_step,
                                                     raw.data_ptr() if raw is not None else None, layout, scale,
                                                     state, fold.handle, npart),
               "dspsr_amd_filterbank_perform_fold")

    def finish(self):
        self.ctx.synchronize()

    def close(self):
        if self.handle and self.ctx.handle:          # (a context closed first has already taken the device objects with it)
            lib.dspsr_amd_filterbank_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConvolutionEngine(FilterbankEngine):
    """Mirror of dsp::Convolution::Engine (Signal/General/dsp/Convolution.h:158-167; CUDA twin
    ConvolutionCUDA.cu:202-800): coherent dedispersion of already-channelised (or single-channel) voltages.
    One forward FFT, response multiply and backward FFT of `ndat` points per (channel, pol, part) -- the
    filterbank object with nchan_subband = 1."""

    def prepare(self, ndat, nfilt_pos, nfilt_neg, nchan=1, npol=2, real_input=False, kernel=None, max_parts=1,
                force_four_pass=False):
        return self.setup(1, ndat, nfilt_pos, nfilt_neg, nchan, npol, real_input, kernel, max_parts=max_parts,
                          force_four_pass=force_four_pass)


def tfp_filterbank(ctx: Context, raw, nchan, npart, out, pscrunch=False, tscrunch=1, layout=_lib.RAW_GENERIC,
                   scale=1.0):
    """digifil front end: dsp::TFPFilterbank (+pscrunch) + dsp::TScrunch fused (TFPFilterbank.C:27-101,
    TScrunch.C:180-206).  raw: device int8 block (real, 2 pols); out: device float32
    [npart // tscrunch][nchan][1 or 2]."""
    cfg = _lib.TfpConfig(nchan, 2, int(bool(pscrunch)), int(tscrunch))
    _check(ctx.handle, lib.dspsr_amd_tfp_filterbank(ctx.handle, C.byref(cfg), raw.data_ptr(), layout, scale,
                                                    out.data_ptr(), npart), "dspsr_amd_tfp_filterbank")


def sigproc_digitize_fpt(ctx: Context, inp, out, nbit=8, use_digi_scales=True, input_scale=1.0, scale_fac=1.0, flip_band=False,
                         swap_band=False):
    """dsp::SigProcDigitizer::pack on FPT rows [nchan][npol][ndat] (float32, device) -> packed n-bit device bytes in TPF order."""
    nchan, npol, ndat = inp.shape
    ics, ips = _strides3(inp)
    _check(ctx.handle, lib.dspsr_amd_sigproc_digitize_fpt(ctx.handle, inp.data_ptr(), ics, ips, ndat, nchan, npol, nbit, int(use_digi_scales),
                                                          input_scale, scale_fac, int(flip_band), int(swap_band), out.data_ptr()),
           "dspsr_amd_sigproc_digitize_fpt")
    return out


def tscrunch_fpt(ctx: Context, inp, out, sfactor, carry, carry_count=0, ndim=1):
    """dsp::TScrunch::fpt_tscrunch on device rows [nchan][npol][ndat * ndim] as a stream (carry: device [nchan][npol][ndim]
    floats).  Returns (nout, carry_count_after)."""
    nchan, npol, nfloat = inp.shape
    ics, ips = _strides3(inp)
    ocs, ops = _strides3(out)
    cc, nout = C.c_uint32(carry_count), C.c_uint64(0)
    _check(ctx.handle, lib.dspsr_amd_tscrunch_fpt(ctx.handle, inp.data_ptr(), ics, ips, out.data_ptr(), ocs, ops, nchan, npol, ndim,
                                                  nfloat // ndim, sfactor, carry.data_ptr(), C.byref(cc), C.byref(nout)),
           "dspsr_amd_tscrunch_fpt")
    return int(nout.value), int(cc.value)


def fscrunch_fpt(ctx: Context, inp, out, sfactor):
    """dsp::FScrunch::fpt_fscrunch on device rows [nchan][npol][nfloat] -> [nchan / sfactor][npol][nfloat]."""
    nchan, npol, nfloat = inp.shape
    ics, ips = _strides3(inp)
    ocs, ops = _strides3(out)
    _check(ctx.handle, lib.dspsr_amd_fscrunch_fpt(ctx.handle, inp.data_ptr(), ics, ips, out.data_ptr(), ocs, ops, nchan, npol, nfloat, sfactor),
           "dspsr_amd_fscrunch_fpt")
    return out


def copy_data_fpt(ctx: Context, to, frm):
    """dsp::TimeSeries::Engine::copy_data_fpt: device rows [nchan][npol][nfloat] (strided views allowed)."""
    nchan, npol, nfloat = frm.shape
    _check(ctx.handle, lib.dspsr_amd_copy_fpt(ctx.handle, to.data_ptr(), to.stride(0), to.stride(1), frm.data_ptr(), frm.stride(0),
                                             frm.stride(1), nchan, npol, nfloat), "dspsr_amd_copy_fpt")


def add_fpt(ctx: Context, to, frm):
    """dsp::TimeSeries::operator += on device rows [nchan][npol][nfloat] (PhaseSeries::combine): to += frm."""
    nchan, npol, nfloat = frm.shape
    _check(ctx.handle, lib.dspsr_amd_add_fpt(ctx.handle, to.data_ptr(), to.stride(0), to.stride(1), frm.data_ptr(), frm.stride(0),
                                            frm.stride(1), nchan, npol, nfloat), "dspsr_amd_add_fpt")


def dedispersion_sample_delays(centre_frequency, bandwidth, dispersion_measure, nchan, rate_hz, swap=False, nsub_swap=0,
                               dc_centred=False):
    """Dedispersion::SampleDelay::match (DedispersionSampleDelay.C:24-75): int64 delay of each channel in samples."""
    d = np.zeros(nchan, np.int64)
    rc = lib.dspsr_amd_dedispersion_sample_delays(centre_frequency, bandwidth, dispersion_measure, nchan, rate_hz, int(swap),
                                                  nsub_swap, int(dc_centred), d.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise DspsrAmdError("dsp::Dedispersion::SampleDelay::match invalid input")
    return d


class SampleDelay:
    """Mirror of dsp::SampleDelay (Signal/General/SampleDelay.C): delays[ichan][ipol] from a SampleDelayFunction."""

    def __init__(self, ctx: Context, delays, npol=1, absolute=False):
        self.ctx = ctx
        d = np.ascontiguousarray(np.asarray(delays, np.int64))
        if d.ndim == 1:                                       # per channel, same for every polarisation
            d = np.ascontiguousarray(np.repeat(d[:, None], npol, axis=1))
        self.nchan, self.npol = d.shape
        h = C.c_void_p()
        _check(ctx.handle, lib.dspsr_amd_sample_delay_create(ctx.handle, self.nchan, self.npol, d.ctypes.data_as(C.c_void_p),
                                                             int(absolute), C.byref(h)), "dspsr_amd_sample_delay_create")
        self.handle = h
        self.zero_delay = lib.dspsr_amd_sample_delay_zero_delay(h)
        self.total_delay = lib.dspsr_amd_sample_delay_total_delay(h)

    def transform(self, inp, out=None):
        """inp/out: float32 device tensors [nchan][npol][ndat][ndim] (the last axis may be absent); returns ndat_out."""
        out = inp if out is None else out
        ndim = inp.shape[3] if inp.dim() == 4 else 1
        n = C.c_uint64(0)
        _check(self.ctx.handle, lib.dspsr_amd_sample_delay_transform(
            self.handle, inp.data_ptr(), inp.stride(0), inp.stride(1), out.data_ptr(), out.stride(0), out.stride(1), ndim,
            inp.shape[2], C.byref(n)), "dspsr_amd_sample_delay_transform")
        return n.value

    def close(self):
        if self.handle and self.ctx.handle:
            lib.dspsr_amd_sample_delay_destroy(self.handle)
        self.handle = None


class Rescale:
    """Mirror of dsp::Rescale for TFP-ordered detected data (Signal/General/Rescale.C): offset/scale per (pol, chan)."""

    def __init__(self, ctx: Context, nchan, npol=1, interval_samples=0, constant=False):
        self.ctx = ctx
        self.nchan, self.npol = nchan, npol
        h = C.c_void_p()
        _check(ctx.handle, lib.dspsr_amd_rescale_create(ctx.handle, nchan, npol, interval_samples, int(constant), C.byref(h)),
               "dspsr_amd_rescale_create")
        self.handle = h

    def transform(self, inp, out=None):
        """inp/out: float32 device tensors [ndat][nchan][npol] (out defaults to in place)."""
        out = inp if out is None else out
        ndat = inp.numel() // (self.nchan * self.npol)
        _check(self.ctx.handle, lib.dspsr_amd_rescale_transform(self.handle, inp.data_ptr(), out.data_ptr(), ndat),
               "dspsr_amd_rescale_transform")
        return out

    def transform_fpt(self, inp, out=None):
        """FPT rows: inp/out float32 device tensors [nchan][npol][ndat] (strided views allowed; out defaults to in place)."""
        out = inp if out is None else out
        ics, ips = _strides3(inp)
        ocs, ops = _strides3(out)
        _check(self.ctx.handle, lib.dspsr_amd_rescale_transform_fpt(self.handle, inp.data_ptr(), ics, ips, out.data_ptr(), ocs, ops,
                                                                    inp.shape[2]), "dspsr_amd_rescale_transform_fpt")
        return out

    def digitize_fpt(self, inp, out, nbit=8, scale_fac=1.0, flip_band=False, swap_band=False):
        """Rescale -> SigProcDigitizer of FPT rows [nchan][npol][ndat] in one pass (the bytes of transform_fpt() +
        sigproc_digitize_fpt(), without the rescaled block)."""
        ics, ips = _strides3(inp)
        ndat = inp.shape[2]
        need = ndat * self.nchan * self.npol * nbit // 8
        if out.numel() * out.element_size() < need:
            raise DspsrAmdError("dspsr_amd.Rescale.digitize_fpt: out holds %d bytes, %d needed" % (out.numel() * out.element_size(), need))
        _check(self.ctx.handle, lib.dspsr_amd_rescale_digitize_fpt(self.handle, inp.data_ptr(), ics, ips, ndat, nbit, scale_fac,
                                                                   int(flip_band), int(swap_band), out.data_ptr()),
               "dspsr_amd_rescale_digitize_fpt")
        return out

    def pscrunch_digitize(self, inp, out, nbit=8, scale_fac=1.0, flip_band=False, swap_band=False):
        """Rescale -> PScrunch -> SigProcDigitizer of a PPQQ block [ndat][nchan][2] in one pass (the bytes of transform() +
        pscrunch_tfp() + sigproc_digitize(), without the two intermediate blocks)."""
        ndat = inp.numel() // (self.nchan * self.npol)
        need = ndat * self.nchan * nbit // 8
        if out.numel() * out.element_size() < need:
            raise DspsrAmdError("dspsr_amd.Rescale.pscrunch_digitize: out holds %d bytes, %d needed" % (out.numel() * out.element_size(), need))
        _check(self.ctx.handle,
               lib.dspsr_amd_rescale_pscrunch_digitize(self.handle, inp.data_ptr(), ndat, nbit, scale_fac, int(flip_band), int(swap_band),
                                                       out.data_ptr()), "dspsr_amd_rescale_pscrunch_digitize")
        return out

    def get(self):
        off = np.empty(self.nchan * self.npol, np.float32)
        sc = np.empty(self.nchan * self.npol, np.float32)
        _check(self.ctx.handle, lib.dspsr_amd_rescale_get(self.handle, off.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)),
               "dspsr_amd_rescale_get")
        return off.reshape(self.nchan, self.npol), sc.reshape(self.nchan, self.npol)

    def close(self):
        if self.handle and self.ctx.handle:
            lib.dspsr_amd_rescale_destroy(self.handle)
        self.handle = None


def pscrunch_tfp(ctx: Context, inp, out, nchan, npol=2):
    """dsp::PScrunch on TFP-ordered device data: out[t][c] = (p0 + p1) * float(1/sqrt(2)); out of place."""
    ndat = inp.numel() // (nchan * npol)
    _check(ctx.handle, lib.dspsr_amd_pscrunch_tfp(ctx.handle, inp.data_ptr(), out.data_ptr(), ndat, nchan, npol),
           "dspsr_amd_pscrunch_tfp")
    return out


def sigproc_digitize(ctx: Context, inp, out, nchan, npol=1, nbit=8, use_digi_scales=True, input_scale=1.0, scale_fac=1.0,
                     flip_band=False, swap_band=False):
    """dsp::SigProcDigitizer::pack on TFP-ordered float32 device data -> packed n-bit device bytes."""
    ndat = inp.numel() // (nchan * npol)
    _check(ctx.handle, lib.dspsr_amd_sigproc_digitize(ctx.handle, inp.data_ptr(), ndat, nchan, npol, nbit, int(use_digi_scales),
                                                      input_scale, scale_fac, int(flip_band), int(swap_band), out.data_ptr()),
           "dspsr_amd_sigproc_digitize")
    return out


class DetectionEngine:
    """dsp::Detection::Engine."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def polarimetry(self, ndim, inp, out, state=_lib.COHERENCE):
        ics, ips = _strides3(inp)
        ocs, ops = _strides3(out)
        nchan = inp.shape[0]
        ndat = inp.shape[2] // 2
        _check(self.ctx.handle,
               lib.dspsr_amd_detect_polarimetry(self.ctx.handle, state, ndim, inp.data_ptr(), ics, ips,
                                                out.data_ptr(), ocs, ops, nchan, ndat),
               "dspsr_amd_detect_polarimetry")

    def square_law(self, inp, out, intensity=False):
        ics, ips = _strides3(inp)
        ocs, ops = _strides3(out)
        _check(self.ctx.handle,
               lib.dspsr_amd_detect_square_law(self.ctx.handle, int(intensity), inp.data_ptr(), ics, ips,
                                               out.data_ptr(), ocs, ops, inp.shape[0], inp.shape[1],
                                               inp.shape[2] // 2), "dspsr_amd_detect_square_law")


class FoldEngine:
    """dsp::Fold::Engine; owns the device-resident profiles (get_profiles)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        h = C.c_void_p()
        _check(ctx.handle, lib.dspsr_amd_fold_create(ctx.handle, C.byref(h)), "dspsr_amd_fold_create")
        self.handle = h
        self.shape = None

    def set_shape(self, nchan, npol, ndim, nbin):
        _check(self.ctx.handle, lib.dspsr_amd_fold_set_shape(self.handle, nchan, npol, ndim, nbin),
               "dspsr_amd_fold_set_shape")
        self.shape = (nchan, npol, nbin, ndim)

    def bind_profile(self, prof, nchan, npol, ndim, nbin):
        """Fold::Engine::setup: fold into the caller's device PhaseSeries buffer; prof: float32 device tensor
        [nchan*npol][span >= nbin*ndim] (rows may be padded), None = back to a library-owned profile."""
        if prof is None:
            _check(self.ctx.handle, lib.dspsr_amd_fold_bind_profile(self.handle, None, 0, nchan, npol, ndim, nbin),
                   "dspsr_amd_fold_bind_profile")
        else:
            assert prof.dim() == 2 and prof.shape[0] == nchan * npol and prof.stride(1) == 1
            _check(self.ctx.handle, lib.dspsr_amd_fold_bind_profile(self.handle, prof.data_ptr(), prof.stride(0), nchan, npol,
                                                                    ndim, nbin), "dspsr_amd_fold_bind_profile")
        self.shape = (nchan, npol, nbin, ndim)

    def set_nbin(self, nbin):
        _check(self.ctx.handle, lib.dspsr_amd_fold_set_nbin(self.handle, nbin), "dspsr_amd_fold_set_nbin")

    def set_ndat(self, ndat, idat_start):
        _check(self.ctx.handle, lib.dspsr_amd_fold_set_ndat(self.handle, ndat, idat_start), "dspsr_amd_fold_set_ndat")

    def set_bin(self, idat, ibin, bins_per_samp=0.0):
        _check(self.ctx.handle, lib.dspsr_amd_fold_set_bin(self.handle, idat, ibin, bins_per_samp),
               "dspsr_amd_fold_set_bin")

    def set_bins(self, phi, phase_per_sample, ndat, idat_start, hits: np.ndarray | None = None, weights=None,
                 ndatperweight=0, weight_idat=0):
        """weights (uint32, one per ndatperweight samples; sample idat belongs to weight (idat + weight_idat) //
        ndatperweight): samples of a zero weight are not folded (Fold.C:686-716,746-763)."""
        n = C.c_uint64()
        hp = hits.ctypes.data_as(C.c_void_p) if hits is not None else None
        if weights is None:
            _check(self.ctx.handle,
                   lib.dspsr_amd_fold_set_bins(self.handle, phi, phase_per_sample, ndat, idat_start, hp, C.byref(n)),
                   "dspsr_amd_fold_set_bins")
        else:
            w = np.ascontiguousarray(weights, dtype=np.uint32)
            _check(self.ctx.handle,
                   lib.dspsr_amd_fold_set_bins_weighted(self.handle, phi, phase_per_sample, ndat, idat_start,
                                                        w.ctypes.data_as(C.c_void_p), w.size, ndatperweight, weight_idat, hp,
                                                        C.byref(n)), "dspsr_amd_fold_set_bins_weighted")
        return n.value

    def get_ndat_folded(self):
        return lib.dspsr_amd_fold_get_ndat_folded(self.handle)

    def fold(self, inp):
        cs, ps = _strides3(inp)
        _check(self.ctx.handle, lib.dspsr_amd_fold_fold(self.handle, inp.data_ptr(), cs, ps), "dspsr_amd_fold_fold")

    def fold_zeroed(self, inp, hits_dev):
        """fold() of an input with zeroed samples: hits_dev (uint32 device tensor [nchan][nbin]) counts, per channel, the
        planned samples of polarisation 0 whose first float is not zero (Fold.C:853-866)."""
        cs, ps = _strides3(inp)
        assert hits_dev.is_contiguous() and hits_dev.numel() == self.shape[0] * self.shape[2]
        _check(self.ctx.handle, lib.dspsr_amd_fold_fold_zeroed(self.handle, inp.data_ptr(), cs, ps, hits_dev.data_ptr()),
               "dspsr_amd_fold_fold_zeroed")

    def get_profiles_ptr(self):
        return lib.dspsr_amd_fold_profiles_dev(self.handle)

    def zero(self):
        _check(self.ctx.handle, lib.dspsr_amd_fold_zero(self.handle), "dspsr_amd_fold_zero")

    def synch(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float32)
        _check(self.ctx.handle, lib.dspsr_amd_fold_synch(self.handle, out.ctypes.data_as(C.c_void_p)),
               "dspsr_amd_fold_synch")
        return out

    def close(self):
        if self.handle and self.ctx.handle:
            lib.dspsr_amd_fold_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Communicator:
    """The sub-integration exchange over RCCL / xGMI behind the C-ABI (dspsr_amd_comm_*, csrc/comm.hip): the same entry
    points DSPSR's C++ host calls.  One per pipeline context.  `unique_id` = the 128 bytes of `Communicator.unique_id()`
    made on rank 0 and handed to every rank by the host (bench.py: torch.distributed's store; DSPSR: its MPI transport).
    The Python host shares PyTorch's ROCm runtime (see _lib.load), so the RCCL opened is the one PyTorch ships."""

    SUM, GATHER = _lib.REDUCE_SUM, _lib.REDUCE_GATHER

    @staticmethod
    def _use_torch_rccl():
        try:
            import torch
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(path):
                lib.dspsr_amd_comm_set_library(path.encode())
        except ImportError:
            pass

    @staticmethod
    def unique_id() -> bytes:
        Communicator._use_torch_rccl()
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        if lib.dspsr_amd_comm_unique_id(buf) != 0:
            raise DspsrAmdError("dspsr_amd_comm_unique_id: RCCL not available")
        return buf.raw

    def __init__(self, ctx: Context, nranks: int, rank: int, unique_id: bytes):
        Communicator._use_torch_rccl()
        assert len(unique_id) == _lib.UNIQUE_ID_BYTES
        self.ctx = ctx
        h = C.c_void_p()
        _check(ctx.handle, lib.dspsr_amd_comm_create(ctx.handle, nranks, rank, unique_id, C.byref(h)), "dspsr_amd_comm_create")
        self.handle = h
        self.nranks, self.rank = nranks, rank
        self._shape = None

    def start(self, mode, profile_ptr, span, nrow, row_floats, hits, integration_length, ndat_total, root=0, check_hits=False):
        """Snapshot + collective, asynchronous: the caller may zero the profile and go on at once."""
        hits = np.ascontiguousarray(hits, dtype=np.uint32)
        _check(self.ctx.handle,
               lib.dspsr_amd_reduce_profiles_start(self.handle, mode, root, profile_ptr, span, nrow, row_floats,
                                                   hits.ctypes.data_as(C.c_void_p), hits.size, float(integration_length),
                                                   int(ndat_total), 1 if check_hits else 0), "dspsr_amd_reduce_profiles_start")
        self._shape = (mode, root, nrow * row_floats, hits.size)

    def finish(self, copy=True):
        """Waits.  Returns (profile, hits, integration_length, ndat_total, hits_identical): numpy arrays on the root
        (profile flat: the SUM, or the nranks slices in rank order), None for the first four elsewhere.
        copy=False: `profile` is a view of the communicator's pinned host buffer (valid until the next start) -- what a writer
        that consumes the sub-integration at once wants."""
        mode, root, n, nbin = self._shape
        same = C.c_int(1)
        if self.rank != root:
            _check(self.ctx.handle, lib.dspsr_amd_reduce_profiles_finish(self.handle, None, None, None, None, C.byref(same)),
                   "dspsr_amd_reduce_profiles_finish")
            return None, None, None, None, bool(same.value)
        hits = np.empty(nbin, np.uint32)
        length, ndat = C.c_double(), C.c_uint64()
        _check(self.ctx.handle,
               lib.dspsr_amd_reduce_profiles_finish(self.handle, None, hits.ctypes.data_as(C.c_void_p),
                                                    C.byref(length), C.byref(ndat), C.byref(same)), "dspsr_amd_reduce_profiles_finish")
        nf = C.c_uint64()
        ptr = lib.dspsr_amd_reduce_profiles_result(self.handle, C.byref(nf))
        prof = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(nf.value,))
        return (prof.copy() if copy else prof), hits, length.value, ndat.value, bool(same.value)

    def close(self):
        if self.handle and self.ctx.handle:
            lib.dspsr_amd_comm_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
