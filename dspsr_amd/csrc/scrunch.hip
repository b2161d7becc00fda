// dsp::TScrunch and dsp::FScrunch on FPT-ordered detected rows (digifil's convolving branch, Signal/General/LoadToFil.C:286-304).
//   TScrunch::fpt_tscrunch (Signal/General/TScrunch.C:148-178): out[o] = in[o*sf]; out[o] += in[o*sf + 1]; ... sequentially
//   FScrunch::fpt_fscrunch (Signal/General/FScrunch.C:117-145): out row c = in row c*sf; += in rows c*sf + 1 ... in order
// These are the stand-alone forms: the search-mode launch group of the filterbank (dspsr_amd_filterbank_perform_search) runs the
// time scrunch inside its inverse pass and gives the same numbers (tests/test_gpu_search.py).
#include "engine_internal.h"

namespace dspsr_amd {

// One thread per output sample; the rows are a STREAM (see dspsr_amd.h): the first c0 samples of output 0 are already in `carry`,
// the samples behind the last complete output go to `carry` (summed by the thread of output 0, see below).  blockIdx.y (+ k gridDim.y) = row (chan * npol + pol).
__global__ __launch_bounds__(256) void k_tscrunch_fpt(const float* __restrict__ in, const uint64_t ics, const uint64_t ips,
                                                      float* __restrict__ out, const uint64_t ocs, const uint64_t ops, const uint32_t npol,
                                                      const uint32_t ndim, const uint64_t ndat_in, const uint32_t sf, const uint32_t c0,
                                                      float* __restrict__ carry, const uint64_t nout, const uint32_t rem, const uint32_t nrow)
{
  // The thread of output 0 reads the carry and the partial group behind the last complete output REPLACES it: both belong to the same
  // thread (read first, then written).  With the partial group on a thread of its own -- another workgroup when the outputs of a row
  // need more than one -- the new carry could land before output 0 had read the old one (tests/fuzz_search.py 300 702, case 246: sample
  // 91 of 1729 rows wrong with 2048 channels, four-pass kernels, a call that begins and ends inside an output sample).
  const uint64_t nown = nout ? nout : 1;
  for (uint32_t row = blockIdx.y; row < nrow; row += gridDim.y) {
  const uint32_t chan = row / npol, pol = row % npol;
  const float* __restrict__ x = in + chan * ics + pol * ips;
  float* __restrict__ y = out + chan * ocs + pol * ops;
  // one thread per (output sample, dimension): consecutive threads = the dimensions of a sample, then the next sample
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nown * ndim; w += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t o = w / ndim;
    const uint32_t d = (uint32_t)(w - o * ndim);
    if (o < nout) {
      // stream samples [o*sf, (o+1)*sf) = input samples [o*sf - c0, ...)
      const uint64_t s0 = o * sf;
      uint64_t i = s0 < c0 ? 0 : s0 - c0;
      const uint64_t i1 = s0 + sf - c0;
      float acc;
      if (o == 0 && c0) acc = carry[(uint64_t)row * ndim + d];
      else { acc = x[i * ndim + d]; i++; }
      for (; i < i1; i++) acc = __fadd_rn(acc, x[i * ndim + d]);
      y[o * ndim + d] = acc;
    }
    if (o == 0 && rem) {
      // the open group: stream samples [nout*sf, nout*sf + rem); it starts from the carry when no output was completed
      const uint64_t s0 = nout * sf;
      uint64_t i = s0 < c0 ? 0 : s0 - c0;
      const uint64_t i1 = s0 + rem - c0;
      float acc;
      if (nout == 0 && c0) acc = carry[(uint64_t)row * ndim + d];
      else { acc = x[i * ndim + d]; i++; }
      for (; i < i1; i++) acc = __fadd_rn(acc, x[i * ndim + d]);
      carry[(uint64_t)row * ndim + d] = acc;
    }
  }
  }
}

__global__ __launch_bounds__(256) void k_fscrunch_fpt(const float* __restrict__ in, const uint64_t ics, const uint64_t ips,
                                                      float* __restrict__ out, const uint64_t ocs, const uint64_t ops, const uint32_t npol,
                                                      const uint64_t nfloat, const uint32_t sf, const uint32_t nrow)
{
  for (uint32_t row = blockIdx.y; row < nrow; row += gridDim.y) {
    const uint32_t chan = row / npol, pol = row % npol;
    const float* __restrict__ x = in + (uint64_t)chan * sf * ics + pol * ips;
    float* __restrict__ y = out + chan * ocs + pol * ops;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nfloat; i += (uint64_t)gridDim.x * blockDim.x) {
      float acc = x[i];
      for (uint32_t f = 1; f < sf; f++) acc = __fadd_rn(acc, x[f * ics + i]);
      y[i] = acc;
    }
  }
}

}  // namespace dspsr_amd

using namespace dspsr_amd;

extern "C" int dspsr_amd_tscrunch_fpt(dspsr_amd_ctx* ctx, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                      float* out_dev, uint64_t out_chan_stride, uint64_t out_pol_stride, uint32_t nchan, uint32_t npol,
                                      uint32_t ndim, uint64_t ndat_in, uint32_t sfactor, float* carry_dev, uint32_t* carry_count, uint64_t* nout)
{
  if (!ctx || !carry_count || !nout || !ndim) return DSPSR_AMD_EINVAL;
  if (!sfactor) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::TScrunch::get_factor scrunch factor not set");        // TScrunch.C:88-90
  if (*carry_count >= sfactor)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tscrunch_fpt: carry_count=%u must be < sfactor=%u", *carry_count, sfactor);
  const uint64_t total = (uint64_t)*carry_count + ndat_in;
  *nout = total / sfactor;
  const uint32_t rem = (uint32_t)(total % sfactor);
  if (!nchan || !npol || !ndat_in) return DSPSR_AMD_OK;
  if (!in_dev || (!out_dev && *nout) || !carry_dev) return DSPSR_AMD_EINVAL;
  if (in_dev == out_dev)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tscrunch_fpt: in place is not supported on the device (use a second block)");
  const uint64_t rows = (uint64_t)nchan * npol;
  if (rows > 0xffffffffull) return DSPSR_AMD_EINVAL;
  const uint64_t ngroup = *nout + (rem ? 1 : 0);
  uint64_t bx = (ngroup * ndim + 255) / 256;
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(k_tscrunch_fpt, dim3((uint32_t)bx, (uint32_t)(rows > 65535 ? 65535 : rows)), dim3(256), 0, ctx->stream, in_dev, in_chan_stride,
                     in_pol_stride, out_dev, out_chan_stride, out_pol_stride, npol, ndim, ndat_in, sfactor, *carry_count, carry_dev, *nout, rem, (uint32_t)rows);
  *carry_count = rem;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_tscrunch_fpt: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fscrunch_fpt(dspsr_amd_ctx* ctx, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                      float* out_dev, uint64_t out_chan_stride, uint64_t out_pol_stride, uint32_t nchan_in, uint32_t npol,
                                      uint64_t nfloat, uint32_t sfactor)
{
  if (!ctx) return DSPSR_AMD_EINVAL;
  if (!sfactor) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::FScrunch::get_factor scrunch factor not set");        // FScrunch.C:72-74
  if (nchan_in % sfactor)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_fscrunch_fpt: nchan=%u is not a multiple of the scrunch factor %u", nchan_in, sfactor);
  if (!nchan_in || !npol || !nfloat) return DSPSR_AMD_OK;
  if (!in_dev || !out_dev) return DSPSR_AMD_EINVAL;
  if (in_dev == out_dev)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_fscrunch_fpt: in place is not supported on the device (use a second block)");
  const uint64_t rows = (uint64_t)(nchan_in / sfactor) * npol;
  if (rows > 0xffffffffull) return DSPSR_AMD_EINVAL;
  uint64_t bx = (nfloat + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(k_fscrunch_fpt, dim3((uint32_t)bx, (uint32_t)(rows > 65535 ? 65535 : rows)), dim3(256), 0, ctx->stream, in_dev, in_chan_stride,
                     in_pol_stride, out_dev, out_chan_stride, out_pol_stride, npol, nfloat, sfactor, (uint32_t)rows);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_fscrunch_fpt: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}
