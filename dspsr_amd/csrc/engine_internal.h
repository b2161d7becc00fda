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
  char error[512];
};

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
