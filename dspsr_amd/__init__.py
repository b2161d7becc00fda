"""dspsr_amd: MI355X-native coherent-dedispersion + detection + fold engine (DSPSR hot path).

Importing the package loads libdspsr_amd.so (hand-written HIP for gfx950); it raises if the
library has not been built -- there is no CPU fallback."""
from ._lib import (lib, LIB_PATH, RAW_GENERIC, RAW_CASPSR, RAW_UWB16, COHERENCE, STOKES, INTENSITY, PPQQ,  # noqa: F401
                   FUSED_AUTO, FUSED_ALWAYS, FUSED_NEVER)
from .engine import (Communicator, Context, ConvolutionEngine, Dedispersion, DetectionEngine, DspsrAmdError, FilterbankEngine, FoldEngine, Rescale, SampleDelay, add_fpt, copy_data_fpt, dedispersion_sample_delays, fscrunch_fpt, pscrunch_tfp, sigproc_digitize, sigproc_digitize_fpt, tscrunch_fpt,  # noqa: F401
                     eight_bit_scale, fold_binplan, fold_binplan_runs, optimal_fft_length, tfp_filterbank)

__all__ = ["Communicator", "Context", "ConvolutionEngine", "Dedispersion", "DetectionEngine", "DspsrAmdError", "FilterbankEngine", "FoldEngine", "Rescale", "SampleDelay", "add_fpt", "copy_data_fpt", "dedispersion_sample_delays", "fscrunch_fpt", "pscrunch_tfp", "sigproc_digitize", "sigproc_digitize_fpt", "tscrunch_fpt",
           "eight_bit_scale", "fold_binplan", "fold_binplan_runs", "optimal_fft_length", "tfp_filterbank", "lib", "LIB_PATH", "build_id"]


def build_id() -> str:
    """sha256 (12 hex digits) of the sources the loaded library was built from (csrc/Makefile: BUILD_ID)."""
    return lib.dspsr_amd_build_id().decode()
