"""ctypes binding of libdspsr_amd.so (the C-ABI declared in include/dspsr_amd.h).

The library is the product: there is NO fallback.  If it is missing or a symbol is absent the
import fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSPSR_AMD_LIB: alternative build of the same library (kernel experiments); there is still no fallback
LIB_PATH = os.environ.get("DSPSR_AMD_LIB") or os.path.join(_HERE, "libdspsr_amd.so")

OK, EINVAL, EHIP, ENOMEM, ESTATE = 0, -1, -2, -3, -4
H2D, D2H, D2D = 1, 2, 3
RAW_GENERIC, RAW_CASPSR, RAW_UWB16 = 0, 1, 2
COHERENCE, STOKES, INTENSITY, PPQQ = 0, 1, 2, 3
FUSED_AUTO, FUSED_ALWAYS, FUSED_NEVER = 0, 1, 2
REDUCE_SUM, REDUCE_GATHER = 0, 1
UNIQUE_ID_BYTES = 128


class FilterbankConfig(C.Structure):
    _fields_ = [("nchan_subband", C.c_uint32), ("freq_res", C.c_uint32), ("nfilt_pos", C.c_uint32),
                ("nfilt_neg", C.c_uint32), ("input_nchan", C.c_uint32), ("npol", C.c_uint32),
                ("real_input", C.c_uint32), ("max_parts", C.c_uint32), ("force_four_pass", C.c_uint32),
                ("fused_fold", C.c_uint32)]


class TfpConfig(C.Structure):
    _fields_ = [("nchan", C.c_uint32), ("npol", C.c_uint32), ("pscrunch", C.c_uint32), ("tscrunch", C.c_uint32)]


class DedispersionConfig(C.Structure):
    _fields_ = [("centre_frequency", C.c_double), ("bandwidth", C.c_double), ("dispersion_measure", C.c_double),
                ("input_nchan", C.c_uint32), ("nchan", C.c_uint32), ("ndim", C.c_uint32),
                ("dual_sideband", C.c_int32), ("dc_centred", C.c_uint32), ("swap", C.c_uint32),
                ("freq_res", C.c_uint32), ("ndat_max", C.c_uint32), ("fractional_delay", C.c_uint32)]


class DedispersionInfo(C.Structure):
    _fields_ = [("impulse_pos", C.c_uint32), ("impulse_neg", C.c_uint32), ("minimum_ndat", C.c_uint32),
                ("ndat", C.c_uint32)]


_vp, _u32, _u64, _i, _f, _d, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float, C.c_double, C.c_size_t
_pp = C.POINTER(C.c_void_p)

# every symbol include/dspsr_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dspsr_amd_ctx_create": (_i, [_i, _vp, _pp]),
    "dspsr_amd_ctx_destroy": (None, [_vp]),
    "dspsr_amd_last_error": (C.c_char_p, [_vp]),
    "dspsr_amd_stream_sync": (_i, [_vp]),
    "dspsr_amd_version": (C.c_char_p, []),
    "dspsr_amd_build_id": (C.c_char_p, []),
    "dspsr_amd_malloc": (_i, [_vp, _sz, _pp]),
    "dspsr_amd_free": (_i, [_vp, _vp]),
    "dspsr_amd_zero": (_i, [_vp, _vp, _sz]),
    "dspsr_amd_copy": (_i, [_vp, _vp, _vp, _sz, _i]),
    "dspsr_amd_copy_fpt": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _u32, _u32, _u64]),
    "dspsr_amd_add_fpt": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _u32, _u32, _u64]),
    "dspsr_amd_filterbank_create": (_i, [_vp, C.POINTER(FilterbankConfig), _pp]),
    "dspsr_amd_filterbank_destroy": (None, [_vp]),
    "dspsr_amd_filterbank_set_kernel": (_i, [_vp, _vp, _u64]),
    "dspsr_amd_filterbank_sizes": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u32)]),
    "dspsr_amd_filterbank_perform": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _u64, _u64, _u64]),
    "dspsr_amd_filterbank_perform_raw": (_i, [_vp, _vp, _i, _f, _vp, _u64, _u64, _u64, _u64]),
    "dspsr_amd_filterbank_perform_detect": (_i, [_vp, _vp, _u64, _u64, _u64, _vp, _i, _f, _i, _u32, _vp, _u64, _u64,
                                                 _u64]),
    "dspsr_amd_filterbank_fold_is_fused": (_i, [_vp]),
    "dspsr_amd_filterbank_npass": (_i, [_vp, _i]),
    "dspsr_amd_filterbank_perform_fold": (_i, [_vp, _vp, _u64, _u64, _u64, _vp, _i, _f, _i, _vp, _u64]),
    "dspsr_amd_filterbank_perform_search": (_i, [_vp, _vp, _u64, _u64, _u64, _vp, _i, _f, _i, _u32, _vp, _u64, _u64, _vp,
                                                 C.POINTER(_u32), _u64, C.POINTER(_u64)]),
    "dspsr_amd_filterbank_search_is_fused": (_i, [_vp]),
    "dspsr_amd_tscrunch_fpt": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _u32, _u32, _u32, _u64, _u32, _vp, C.POINTER(_u32),
                                    C.POINTER(_u64)]),
    "dspsr_amd_fscrunch_fpt": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _u32, _u32, _u64, _u32]),
    "dspsr_amd_sample_delay_create": (_i, [_vp, _u32, _u32, _vp, _i, _pp]),
    "dspsr_amd_sample_delay_destroy": (None, [_vp]),
    "dspsr_amd_sample_delay_zero_delay": (C.c_int64, [_vp]),
    "dspsr_amd_sample_delay_total_delay": (_u64, [_vp]),
    "dspsr_amd_sample_delay_transform": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _u32, _u64, C.POINTER(_u64)]),
    "dspsr_amd_dedispersion_sample_delays": (_i, [_d, _d, _d, _u32, _d, _i, _u32, _i, _vp]),
    "dspsr_amd_pscrunch_tfp": (_i, [_vp, _vp, _vp, _u64, _u32, _u32]),
    "dspsr_amd_rescale_create": (_i, [_vp, _u32, _u32, _u64, _i, _pp]),
    "dspsr_amd_rescale_destroy": (None, [_vp]),
    "dspsr_amd_rescale_transform": (_i, [_vp, _vp, _vp, _u64]),
    "dspsr_amd_rescale_get": (_i, [_vp, _vp, _vp]),
    "dspsr_amd_rescale_pscrunch_digitize": (_i, [_vp, _vp, _u64, _i, _f, _i, _i, _vp]),
    "dspsr_amd_sigproc_digitize": (_i, [_vp, _vp, _u64, _u32, _u32, _i, _i, _d, _f, _i, _i, _vp]),
    "dspsr_amd_rescale_transform_fpt": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _u64]),
    "dspsr_amd_sigproc_digitize_fpt": (_i, [_vp, _vp, _u64, _u64, _u64, _u32, _u32, _i, _i, _d, _f, _i, _i, _vp]),
    "dspsr_amd_rescale_digitize_fpt": (_i, [_vp, _vp, _u64, _u64, _u64, _i, _f, _i, _i, _vp]),
    "dspsr_amd_detect_polarimetry": (_i, [_vp, _i, _u32, _vp, _u64, _u64, _vp, _u64, _u64, _u32, _u64]),
    "dspsr_amd_detect_square_law": (_i, [_vp, _i, _vp, _u64, _u64, _vp, _u64, _u64, _u32, _u32, _u64]),
    "dspsr_amd_tfp_filterbank": (_i, [_vp, C.POINTER(TfpConfig), _vp, _i, _f, _vp, _u64]),
    "dspsr_amd_fold_create": (_i, [_vp, _pp]),
    "dspsr_amd_fold_destroy": (None, [_vp]),
    "dspsr_amd_fold_set_shape": (_i, [_vp, _u32, _u32, _u32, _u32]),
    "dspsr_amd_fold_bind_profile": (_i, [_vp, _vp, _u64, _u32, _u32, _u32, _u32]),
    "dspsr_amd_fold_set_nbin": (_i, [_vp, _u32]),
    "dspsr_amd_fold_set_ndat": (_i, [_vp, _u64, _u64]),
    "dspsr_amd_fold_set_bin": (_i, [_vp, _u64, _d, _d]),
    "dspsr_amd_fold_set_bins": (_i, [_vp, _d, _d, _u64, _u64, _vp, C.POINTER(_u64)]),
    "dspsr_amd_fold_set_bins_weighted": (_i, [_vp, _d, _d, _u64, _u64, _vp, _u64, _u64, _u64, _vp, C.POINTER(_u64)]),
    "dspsr_amd_fold_fold": (_i, [_vp, _vp, _u64, _u64]),
    "dspsr_amd_fold_fold_zeroed": (_i, [_vp, _vp, _u64, _u64, _vp]),
    "dspsr_amd_fold_profiles_dev": (_vp, [_vp]),
    "dspsr_amd_fold_get_ndat_folded": (_u64, [_vp]),
    "dspsr_amd_fold_zero": (_i, [_vp]),
    "dspsr_amd_fold_synch": (_i, [_vp, _vp]),
    "dspsr_amd_comm_set_library": (_i, [C.c_char_p]),
    "dspsr_amd_comm_unique_id": (_i, [_vp]),
    "dspsr_amd_comm_create": (_i, [_vp, _i, _i, _vp, _pp]),
    "dspsr_amd_comm_destroy": (None, [_vp]),
    "dspsr_amd_comm_rank": (_i, [_vp]),
    "dspsr_amd_comm_size": (_i, [_vp]),
    "dspsr_amd_reduce_profiles_start": (_i, [_vp, _i, _i, _vp, _u64, _u64, _u64, _vp, _u32, _d, _u64, _i]),
    "dspsr_amd_reduce_profiles_result": (_vp, [_vp, C.POINTER(_u64)]),
    "dspsr_amd_reduce_profiles_finish": (_i, [_vp, _vp, _vp, C.POINTER(_d), C.POINTER(_u64), C.POINTER(_i)]),
    "dspsr_amd_dedispersion_prepare": (_i, [C.POINTER(DedispersionConfig), C.POINTER(DedispersionInfo), C.c_char_p,
                                            _sz]),
    "dspsr_amd_dedispersion_build": (_i, [C.POINTER(DedispersionConfig), _u32, _vp]),
    "dspsr_amd_optimal_fft_length": (_u64, [_u64, _u64]),
    "dspsr_amd_eight_bit_scale": (_d, [_d]),
    "dspsr_amd_fold_binplan": (_i, [_d, _d, _u32, _u64, _vp, _vp]),
    "dspsr_amd_fold_binplan_runs": (_i, [_d, _d, _u32, _u64, _vp, _vp, _vp, _u64, _vp, _vp]),
}


def load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "dspsr_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process.  torch ships its own libamdhip64.so.7 / libhsa-runtime64 and the Python host side
    # shares streams and device memory with it, so torch's copy must be the one the dynamic linker binds: import torch
    # BEFORE the library (loading libdspsr_amd.so first binds /opt/rocm's runtime, and the context then finds no device
    # once torch has initialised its own).  A C/C++ host such as DSPSR has no torch in the process and links /opt/rocm.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
