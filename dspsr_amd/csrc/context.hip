// Context (one per pipeline thread / GPU, one stream: SingleThread.C:213-290) and the
// dsp::Memory operations (Kernel/Classes/dsp/Memory.h:18-34, CUDA impl MemoryCUDA.C:47-106).
#include <math.h>

#include <vector>

#include "engine_internal.h"

using namespace dspsr_amd;

#ifndef DSPSR_AMD_BUILD_ID
#define DSPSR_AMD_BUILD_ID "unknown"     // (the Makefile passes the sha256 of the sources, see there)
#endif
extern "C" const char* dspsr_amd_version(void) { return "dspsr_amd 0.5 (gfx950) build " DSPSR_AMD_BUILD_ID; }
extern "C" const char* dspsr_amd_build_id(void) { return DSPSR_AMD_BUILD_ID; }

extern "C" int dspsr_amd_ctx_create(int device, void* hip_stream, dspsr_amd_ctx** out)
{
  if (!out) return DSPSR_AMD_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return DSPSR_AMD_EHIP;
  if (device < 0 || device >= ndev) return DSPSR_AMD_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return DSPSR_AMD_EHIP;
  dspsr_amd_ctx* ctx = new dspsr_amd_ctx;
  ctx->device = device;
  ctx->error[0] = 0;
  hipDeviceProp_t prop;
  ctx->ncu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                 ? (uint32_t)prop.multiProcessorCount : 256u;
  ctx->own_stream = (hip_stream == DSPSR_AMD_NEW_STREAM);
  ctx->stream = ctx->own_stream ? nullptr : (hipStream_t)hip_stream;
  if (ctx->own_stream && hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return DSPSR_AMD_EHIP;
  }
  std::vector<cf> tw(TWN);
  for (int j = 0; j < TWN; j++) {
    const double a = -2.0 * M_PI * (double)j / (double)TWN;
    tw[j] = make_float2((float)cos(a), (float)sin(a));
  }
  if (hipMalloc((void**)&ctx->tw, TWN * sizeof(cf)) != hipSuccess ||
      hipMemcpy(ctx->tw, tw.data(), TWN * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess) {
    delete ctx;
    return DSPSR_AMD_ENOMEM;
  }
  *out = ctx;
  return DSPSR_AMD_OK;
}

extern "C" void dspsr_amd_ctx_destroy(dspsr_amd_ctx* ctx)
{
  if (!ctx) return;
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->tw) (void)hipFree(ctx->tw);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* dspsr_amd_last_error(const dspsr_amd_ctx* ctx) { return ctx ? ctx->error : "null context"; }

extern "C" int dspsr_amd_stream_sync(dspsr_amd_ctx* ctx)
{
  if (!ctx) return DSPSR_AMD_EINVAL;
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_stream_sync: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_malloc(dspsr_amd_ctx* ctx, size_t nbytes, void** ptr)
{
  if (!ctx || !ptr) return DSPSR_AMD_EINVAL;
  *ptr = nullptr;
  if (nbytes == 0) return DSPSR_AMD_OK;
  hipError_t e = hipMalloc(ptr, nbytes);
  if (e != hipSuccess)
    return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_malloc: hipMalloc(%zu): %s", nbytes, hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_free(dspsr_amd_ctx* ctx, void* ptr)
{
  if (!ctx) return DSPSR_AMD_EINVAL;
  if (!ptr) return DSPSR_AMD_OK;
  hipError_t e = hipFree(ptr);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_free: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_zero(dspsr_amd_ctx* ctx, void* ptr, size_t nbytes)
{
  if (!ctx || (!ptr && nbytes)) return DSPSR_AMD_EINVAL;
  if (!nbytes) return DSPSR_AMD_OK;
  hipError_t e = hipMemsetAsync(ptr, 0, nbytes, ctx->stream);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_zero: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_copy(dspsr_amd_ctx* ctx, void* dst, const void* src, size_t nbytes, int kind)
{
  if (!ctx || ((!dst || !src) && nbytes)) return DSPSR_AMD_EINVAL;
  if (!nbytes) return DSPSR_AMD_OK;
  hipMemcpyKind k;
  switch (kind) {
    case DSPSR_AMD_H2D: k = hipMemcpyHostToDevice; break;
    case DSPSR_AMD_D2H: k = hipMemcpyDeviceToHost; break;
    case DSPSR_AMD_D2D: k = hipMemcpyDeviceToDevice; break;
    default: return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_copy: invalid kind %d", kind);
  }
  hipError_t e = hipMemcpyAsync(dst, src, nbytes, k, ctx->stream);
  if (e == hipSuccess && kind == DSPSR_AMD_D2H) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_copy: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

// dsp::TimeSeries::Engine::copy_data_fpt: one (chan, pol) row per blockIdx.y/z, grid-stride over the row
__global__ __launch_bounds__(256) void k_copy_fpt(float* __restrict__ to, const uint64_t tcs, const uint64_t tps,
                                                  const float* __restrict__ from, const uint64_t fcs, const uint64_t fps,
                                                  const uint64_t nfloat)
{
  float* __restrict__ t = to + blockIdx.z * tcs + blockIdx.y * tps;
  const float* __restrict__ f = from + blockIdx.z * fcs + blockIdx.y * fps;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (((((uintptr_t)t) | ((uintptr_t)f)) & 15) == 0) {          // aligned rows: 16 bytes per lane
    const uint64_t n4 = nfloat >> 2;
    for (uint64_t j = i; j < n4; j += stride) ((float4*)t)[j] = ((const float4*)f)[j];
    for (uint64_t j = (n4 << 2) + i; j < nfloat; j += stride) t[j] = f[j];
  } else {
    for (; i < nfloat; i += stride) t[i] = f[i];
  }
}

extern "C" int dspsr_amd_copy_fpt(dspsr_amd_ctx* ctx, float* to_dev, uint64_t to_chan_stride, uint64_t to_pol_stride,
                                  const float* from_dev, uint64_t from_chan_stride, uint64_t from_pol_stride,
                                  uint32_t nchan, uint32_t npol, uint64_t nfloat)
{
  if (!ctx || ((!to_dev || !from_dev) && nfloat)) return DSPSR_AMD_EINVAL;
  if (!nfloat || !nchan || !npol) return DSPSR_AMD_OK;
  if (npol > 65535 || nchan > 65535)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_copy_fpt: nchan=%u npol=%u exceed the grid limits", nchan, npol);
  uint64_t bx = (nfloat / 4 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_copy_fpt, dim3((uint32_t)bx, npol, nchan), dim3(256), 0, ctx->stream, to_dev, to_chan_stride,
                     to_pol_stride, from_dev, from_chan_stride, from_pol_stride, nfloat);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_copy_fpt: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

// dsp::TimeSeries::operator += on device rows (PhaseSeries::combine, PhaseSeries.C:442-484): one (chan, pol) row per
// blockIdx.y/z, grid-stride over the row; plain float adds (the same sum the host loop of TimeSeries.C makes)
__global__ __launch_bounds__(256) void k_add_fpt(float* __restrict__ to, const uint64_t tcs, const uint64_t tps,
                                                 const float* __restrict__ from, const uint64_t fcs, const uint64_t fps,
                                                 const uint64_t nfloat)
{
  float* __restrict__ t = to + blockIdx.z * tcs + blockIdx.y * tps;
  const float* __restrict__ f = from + blockIdx.z * fcs + blockIdx.y * fps;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (((((uintptr_t)t) | ((uintptr_t)f)) & 15) == 0) {
    const uint64_t n4 = nfloat >> 2;
    for (uint64_t j = i; j < n4; j += stride) {
      float4 a = ((float4*)t)[j];
      const float4 b = ((const float4*)f)[j];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      ((float4*)t)[j] = a;
    }
    for (uint64_t j = (n4 << 2) + i; j < nfloat; j += stride) t[j] += f[j];
  } else {
    for (; i < nfloat; i += stride) t[i] += f[i];
  }
}

extern "C" int dspsr_amd_add_fpt(dspsr_amd_ctx* ctx, float* to_dev, uint64_t to_chan_stride, uint64_t to_pol_stride,
                                 const float* from_dev, uint64_t from_chan_stride, uint64_t from_pol_stride,
                                 uint32_t nchan, uint32_t npol, uint64_t nfloat)
{
  if (!ctx || ((!to_dev || !from_dev) && nfloat)) return DSPSR_AMD_EINVAL;
  if (!nfloat || !nchan || !npol) return DSPSR_AMD_OK;
  if (npol > 65535 || nchan > 65535)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_add_fpt: nchan=%u npol=%u exceed the grid limits", nchan, npol);
  if (to_dev == from_dev)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_add_fpt: a PhaseSeries cannot be combined with itself");
  uint64_t bx = (nfloat / 4 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_add_fpt, dim3((uint32_t)bx, npol, nchan), dim3(256), 0, ctx->stream, to_dev, to_chan_stride,
                     to_pol_stride, from_dev, from_chan_stride, from_pol_stride, nfloat);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_add_fpt: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}
