// Stand-alone dsp::Detection::Engine kernels (Signal/General/dsp/Detection.h:98-106).
// Reference: Detection.C:218-474, cross_detect.ic:23-43, stokes_detect.ic:21-44; CUDA twin
// DetectionCUDA.cu:97-119 (coherence2), :180-213 (sqld).  Elementwise, HBM-bound: one pass,
// 8-byte coalesced loads of both polarisations, widest store the layout allows.
// (The dspsr pipeline path uses the detection fused into filterbank pass 3 instead.)
#include "engine_internal.h"

namespace dspsr_amd {

// `in` and `out` may be the same buffer (ndim 2, LoadToFold1.C:545-546): every thread reads its two samples before it
// writes the same elements back, so the pointers carry no __restrict__.
__global__ void k_polarimetry(const int state, const uint32_t ndim, const float* in,
                              const uint64_t in_chan_stride, const uint64_t in_pol_stride, float* out,
                              const uint64_t out_chan_stride, const uint64_t out_pol_stride, const uint64_t ndat)
{
  const uint32_t chan = blockIdx.y;
  const float2* p = (const float2*)(in + chan * in_chan_stride);
  const float2* q = (const float2*)(in + chan * in_chan_stride + in_pol_stride);
  float* row = out + chan * out_chan_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndat; i += (uint64_t)gridDim.x * blockDim.x) {
    const float2 a = p[i], b = q[i];
    const float pp = a.x * a.x + a.y * a.y;
    const float qq = b.x * b.x + b.y * b.y;
    const float re = a.x * b.x + a.y * b.y;
    const float im = a.x * b.y - a.y * b.x;
    float r0 = pp, r1 = qq, r2 = re, r3 = im;
    if (state == DSPSR_AMD_STOKES) { r0 = pp + qq; r1 = pp - qq; r2 = 2.0f * re; r3 = 2.0f * im; }
    if (ndim == 4) {
      ((float4*)row)[i] = make_float4(r0, r1, r2, r3);
    } else if (ndim == 2) {
      ((float2*)row)[i] = make_float2(r0, r1);
      ((float2*)(row + out_pol_stride))[i] = make_float2(r2, r3);
    } else {
      row[i] = r0;
      row[out_pol_stride + i] = r1;
      row[2 * out_pol_stride + i] = r2;
      row[3 * out_pol_stride + i] = r3;
    }
  }
}

// ndim 2, two samples per thread: 16-byte loads of both polarisations, 16-byte stores of both planes (the layout the GPU
// pipeline uses, in place, LoadToFold1.C:545-546,1105-1109).  Per-lane access width matters on this chip (8 B per lane streams
// at 5.6 TB/s, 16 B at 7.1: tools/load_width_probe.hip).  Same expressions as k_polarimetry: identical results.
__global__ __launch_bounds__(256) void k_polarimetry2x(const int state, const float* in, const uint64_t in_chan_stride,
                                                       const uint64_t in_pol_stride, float* out, const uint64_t out_chan_stride,
                                                       const uint64_t out_pol_stride, const uint64_t npair)
{
  const uint32_t chan = blockIdx.y;
  const float4* p = (const float4*)(in + chan * in_chan_stride);
  const float4* q = (const float4*)(in + chan * in_chan_stride + in_pol_stride);
  float4* o0 = (float4*)(out + chan * out_chan_stride);
  float4* o1 = (float4*)(out + chan * out_chan_stride + out_pol_stride);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npair; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 a = p[i], b = q[i];
    float r[2][4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float ax = h ? a.z : a.x, ay = h ? a.w : a.y, bx = h ? b.z : b.x, by = h ? b.w : b.y;
      const float pp = ax * ax + ay * ay;
      const float qq = bx * bx + by * by;
      const float re = ax * bx + ay * by;
      const float im = ax * by - ay * bx;
      r[h][0] = pp; r[h][1] = qq; r[h][2] = re; r[h][3] = im;
      if (state == DSPSR_AMD_STOKES) { r[h][0] = pp + qq; r[h][1] = pp - qq; r[h][2] = 2.0f * re; r[h][3] = 2.0f * im; }
    }
    o0[i] = make_float4(r[0][0], r[0][1], r[1][0], r[1][1]);
    o1[i] = make_float4(r[0][2], r[0][3], r[1][2], r[1][3]);
  }
}

__global__ void k_square_law(const int intensity, const uint32_t npol, const float* __restrict__ in,
                             const uint64_t in_chan_stride, const uint64_t in_pol_stride, float* __restrict__ out,
                             const uint64_t out_chan_stride, const uint64_t out_pol_stride, const uint64_t ndat)
{
  const uint32_t chan = blockIdx.y;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndat; i += (uint64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (uint32_t ipol = 0; ipol < npol; ipol++) {
      const float2 a = ((const float2*)(in + chan * in_chan_stride + ipol * in_pol_stride))[i];
      // Detection.C:273-279: `*out = re*re; *out += im*im;` -- two roundings, as the reference's x86 host code (no fused
      // multiply-add), and the same expression as the search-mode epilogue of the filterbank (fb_common.h sqld)
      const float v = __fadd_rn(__fmul_rn(a.x, a.x), __fmul_rn(a.y, a.y));
      if (intensity) acc = ipol ? __fadd_rn(acc, v) : v;          // Detection.C:285-300: *p0 += *p1
      else out[chan * out_chan_stride + ipol * out_pol_stride + i] = v;
    }
    if (intensity) out[chan * out_chan_stride + i] = acc;
  }
}

}  // namespace dspsr_amd

using namespace dspsr_amd;

extern "C" int dspsr_amd_detect_polarimetry(dspsr_amd_ctx* ctx, int state, uint32_t ndim, const float* in_dev,
                                            uint64_t in_chan_stride, uint64_t in_pol_stride, float* out_dev,
                                            uint64_t out_chan_stride, uint64_t out_pol_stride, uint32_t nchan,
                                            uint64_t ndat)
{
  if (!ctx || !in_dev || !out_dev) return DSPSR_AMD_EINVAL;
  if (ndim != 1 && ndim != 2 && ndim != 4)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::Detection::get_result_pointers invalid ndim=%u", ndim);
  if (state != DSPSR_AMD_COHERENCE && state != DSPSR_AMD_STOKES)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_detect_polarimetry: invalid state=%d", state);
  if (in_dev == out_dev && ndim != 2)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL,
                    "dspsr_amd_detect_polarimetry: in-place only when ndim==2 (Detection.C:358-366)");
  if (ndat == 0 || nchan == 0) return DSPSR_AMD_OK;
  const uint32_t threads = 256;
  const bool vec = ndim == 2 && (ndat % 2) == 0 && ((uintptr_t)in_dev % 16) == 0 && ((uintptr_t)out_dev % 16) == 0 &&
                   (in_chan_stride % 4) == 0 && (in_pol_stride % 4) == 0 && (out_chan_stride % 4) == 0 && (out_pol_stride % 4) == 0;
  if (vec) {
    uint64_t bx = (ndat / 2 + threads - 1) / threads;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(k_polarimetry2x, dim3((uint32_t)bx, nchan), dim3(threads), 0, ctx->stream, state, in_dev, in_chan_stride,
                       in_pol_stride, out_dev, out_chan_stride, out_pol_stride, ndat / 2);
  } else {
    uint64_t bx = (ndat + threads - 1) / threads;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(k_polarimetry, dim3((uint32_t)bx, nchan), dim3(threads), 0, ctx->stream, state, ndim, in_dev,
                       in_chan_stride, in_pol_stride, out_dev, out_chan_stride, out_pol_stride, ndat);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_detect_polarimetry: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_detect_square_law(dspsr_amd_ctx* ctx, int intensity, const float* in_dev,
                                           uint64_t in_chan_stride, uint64_t in_pol_stride, float* out_dev,
                                           uint64_t out_chan_stride, uint64_t out_pol_stride, uint32_t nchan,
                                           uint32_t npol, uint64_t ndat)
{
  if (!ctx || !in_dev || !out_dev) return DSPSR_AMD_EINVAL;
  if (npol != 1 && npol != 2) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_detect_square_law: npol=%u", npol);
  if (in_dev == out_dev)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_detect_square_law: in-place not supported");
  if (ndat == 0 || nchan == 0) return DSPSR_AMD_OK;
  const uint32_t threads = 256;
  uint64_t bx = (ndat + threads - 1) / threads;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(k_square_law, dim3((uint32_t)bx, nchan), dim3(threads), 0, ctx->stream, intensity, npol, in_dev,
                     in_chan_stride, in_pol_stride, out_dev, out_chan_stride, out_pol_stride, ndat);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_detect_square_law: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}
