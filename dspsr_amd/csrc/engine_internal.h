// Shared internals of the C-ABI implementation (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dspsr_amd.h"
#include "wgfft.h"

struct dspsr_amd_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  dspsr_amd::cf* tw;  // exp(-2*pi*i*j/TWN), j < TWN (built in double on the host)
  uint32_t ncu;       // compute units of the device (queried once, at context creation)
  char error[512];
};

// The dynamic-LDS limit is a per-function, process-wide attribute: it is only ever raised (several objects may share a
// kernel with different tile sizes), once per new maximum, outside the per-block calls.
#include <map>
#include <mutex>
inline hipError_t dspsr_amd_allow_lds(const void* kern, size_t bytes)
{
  static std::mutex mtx;
  static std::map<const void*, size_t> limit;
  std::lock_guard<std::mutex> lock(mtx);
  size_t& cur = limit[kern];
  if (bytes <= cur) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) cur = bytes;
  return e;
}

static inline void ctx_set_error_v(dspsr_amd_ctx* ctx, const char* fmt, va_list ap)
{
  if (ctx) vsnprintf(ctx->error, sizeof(ctx->error), fmt, ap);
}

static inline int ctx_fail(dspsr_amd_ctx* ctx, int code, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  ctx_set_error_v(ctx, fmt, ap);
  va_end(ap);
  return code;
}
