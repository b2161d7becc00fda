// Search-mode output stage for gfx950: dsp::Rescale followed by dsp::SigProcDigitizer on
// time-major (TFP) detected data, as digifil wires them after the TFP filterbank and the scrunches
// (Signal/General/LoadToFil.C:318-362).
//
//   dsp::Rescale::transformation  (Signal/General/Rescale.C:157-388): per (pol, chan) running sums of x and x^2
//       (double) over intervals of `nsample` samples; at the end of an interval -- and right after the very first
//       call -- compute_various (:390-420) sets offset = -mean, scale = 1/sqrt(variance) (1 when the variance is 0),
//       unless `constant` freezes the first estimate; every sample leaves as (x + offset) * scale.
//   dsp::SigProcDigitizer::pack, TFP branch (Kernel/Formats/sigproc/SigProcDigitizer.C:80-236): nbit 1/2/4/8/16,
//       result = int(x * digi_scale + digi_mean + 0.5) clipped to [0, 2^nbit - 1], digi_scale = digi_mean/6 (8-bit:
//       127.5/6) divided by input_scale * scale_fac; output channel k takes input channel ChannelSort(k) (:38-66);
//       sub-byte samples are packed LSB first; -32 (pack_float, :309-342) writes TPF floats divided by the scale.
//
// The reference runs both on the host after a device-to-host transfer; here the block stays in HBM.  The sums are
// tree reductions in double (the reference adds sample by sample in double): means agree to ~1e-16 relative, so a
// digitised value can differ by one level only when x*scale falls within that distance of a rounding boundary.
#include <math.h>

#include "engine_internal.h"

namespace dspsr_amd {

// partial sums over a slice of time samples: block (x: chan-pol tile of 256 columns, y: time slice)
__global__ __launch_bounds__(256) void k_rescale_sums(const float* __restrict__ in, const uint64_t ndat, const uint32_t ncol,
                                                      const uint32_t rows_per_block, double* __restrict__ part_sum,
                                                      double* __restrict__ part_sq)
{
  const uint32_t col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ncol) return;
  const uint64_t r0 = (uint64_t)blockIdx.y * rows_per_block;
  const uint64_t r1 = r0 + rows_per_block < ndat ? r0 + rows_per_block : ndat;
  double s = 0.0, q = 0.0;
  uint64_t r = r0;
  // eight loads in flight, added in row order (one load per iteration left the pass at one memory round trip per row:
  // 33 us per 67 MB block)
  for (; r + 8 <= r1; r += 8) {                      // consecutive threads read consecutive floats of a row
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = in[(r + u) * ncol + col];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      s += (double)v[u];
      q += (double)__fmul_rn(v[u], v[u]);            // the reference squares in float and accumulates in double (:243-244)
    }
  }
  for (; r < r1; r++) {
    const float v = in[r * ncol + col];
    s += (double)v;
    q += (double)__fmul_rn(v, v);
  }
  part_sum[(uint64_t)blockIdx.y * ncol + col] = s;
  part_sq[(uint64_t)blockIdx.y * ncol + col] = q;
}

// total += sum over slices (slice order fixed => deterministic)
__global__ __launch_bounds__(256) void k_rescale_accumulate(const double* __restrict__ part_sum, const double* __restrict__ part_sq,
                                                            const uint32_t nslice, const uint32_t ncol,
                                                            double* __restrict__ total, double* __restrict__ totalsq)
{
  const uint32_t col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ncol) return;
  double s = total[col], q = totalsq[col];
  uint32_t i = 0;
  for (; i + 8 <= nslice; i += 8) {                  // (loads ahead, sums in slice order)
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { a[u] = part_sum[(uint64_t)(i + u) * ncol + col]; b[u] = part_sq[(uint64_t)(i + u) * ncol + col]; }
#pragma unroll
    for (int u = 0; u < 8; u++) { s += a[u]; q += b[u]; }
  }
  for (; i < nslice; i++) { s += part_sum[(uint64_t)i * ncol + col]; q += part_sq[(uint64_t)i * ncol + col]; }
  total[col] = s;
  totalsq[col] = q;
}

// Rescale::compute_various + zeroing of the running sums (Rescale.C:390-420, :318-325)
__global__ __launch_bounds__(256) void k_rescale_update(double* __restrict__ total, double* __restrict__ totalsq, const uint32_t ncol,
                                                        const double isample, const int set_scale, float* __restrict__ offset,
                                                        float* __restrict__ scale)
{
  const uint32_t col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ncol) return;
  const double mean = total[col] / isample, meansq = totalsq[col] / isample;
  const double variance = __dsub_rn(meansq, __dmul_rn(mean, mean));   // no fma contraction: the host code rounds the product
  if (set_scale) {
    offset[col] = (float)(-mean);
    scale[col] = variance == 0.0 ? 1.0f : (float)(1.0 / sqrt(variance));
  }
  total[col] = 0.0;
  totalsq[col] = 0.0;
}

__global__ __launch_bounds__(256) void k_rescale_apply(const float* in, float* out /* may alias in */, const uint64_t n,
                                                       const uint32_t ncol, const float* __restrict__ offset,
                                                       const float* __restrict__ scale)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t col = (uint32_t)(i % ncol);
    out[i] = __fmul_rn(__fadd_rn(in[i], offset[col]), scale[col]);      // Rescale.C:352
  }
}

// SigProcDigitizer::pack, TFP order.  One thread per output byte (8-bit: one sample; sub-byte: 8/nbit samples of
// consecutive output channels; 16-bit: one thread per sample).
__global__ __launch_bounds__(256) void k_sigproc_digitize(const float* __restrict__ in, uint8_t* __restrict__ out, const uint64_t ndat,
                                                          const uint32_t nchan, const uint32_t npol, const int nbit,
                                                          const float digi_scale, const float digi_mean, const float xpol_offset,
                                                          const int digi_max, const int flip_band, const int swap_band)
{
  const uint32_t spb = nbit >= 8 ? 1 : 8 / nbit;                 // samples per byte
  const uint64_t units_per_row = (uint64_t)npol * (nchan / spb);  // bytes (16-bit: samples) per time sample
  const uint64_t total = ndat * units_per_row;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += stride) {
    const uint64_t idat = u / units_per_row;
    const uint32_t w = (uint32_t)(u % units_per_row);
    const uint32_t ipol = w / (nchan / spb), k0 = (w % (nchan / spb)) * spb;
    const float mean = digi_mean + (ipol > 1 ? xpol_offset : 0.0f);
    uint32_t byte = 0;
    for (uint32_t j = 0; j < spb; j++) {
      uint32_t ic = k0 + j;                                       // ChannelSort, SigProcDigitizer.C:52-65
      if (swap_band) ic = (ic + nchan / 2) % nchan;
      if (flip_band) ic = nchan - ic - 1;
      const float x = in[(idat * nchan + ic) * npol + ipol];
      // :198  float multiply, float add, then + 0.5 in double (the literal is a double), truncation
      const double d = (double)__fadd_rn(__fmul_rn(x, digi_scale), mean) + 0.5;
      // out-of-range and NaN conversions yield INT_MIN on the reference's x86 hosts (cvttsd2si), i.e. clip to 0
      int r = (d >= 2147483648.0 || d <= -2147483649.0 || d != d) ? (int)0x80000000 : (int)d;
      r = r < 0 ? 0 : r;
      r = r > digi_max ? digi_max : r;
      byte |= (uint32_t)r << (j * (nbit >= 8 ? 0 : nbit));
    }
    if (nbit == 16) ((uint16_t*)out)[u] = (uint16_t)byte;
    else out[u] = (uint8_t)byte;
  }
}

// SigProcDigitizer::pack_float, TFP branch (:325-342): out[idat][ipol][k] = in[idat][ChannelSort(k)][ipol] / scale
__global__ __launch_bounds__(256) void k_sigproc_float(const float* __restrict__ in, float* __restrict__ out, const uint64_t ndat,
                                                       const uint32_t nchan, const uint32_t npol, const float scale,
                                                       const int flip_band, const int swap_band)
{
  const uint64_t row = (uint64_t)nchan * npol, total = ndat * row;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += stride) {
    const uint64_t idat = u / row;
    const uint32_t w = (uint32_t)(u % row), ipol = w / nchan;
    uint32_t ic = w % nchan;
    if (swap_band) ic = (ic + nchan / 2) % nchan;
    if (flip_band) ic = nchan - ic - 1;
    out[u] = __fdiv_rn(in[(idat * nchan + ic) * npol + ipol], scale);
  }
}

// dsp::PScrunch::transformation, TFP branch (Signal/General/PScrunch.C:52,72-90): out[t][c] = (p0 + p1) * float(1/sqrt(2))
__global__ __launch_bounds__(256) void k_pscrunch_tfp(const float* __restrict__ in, float* __restrict__ out, const uint64_t n,
                                                      const uint32_t npol, const float scale)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = __fmul_rn(__fadd_rn(in[i * npol], in[i * npol + 1]), scale);
}


// ---- FPT order: rows [chan][pol] of ndat floats (Rescale.C:232-262,330-347; SigProcDigitizer.C:238-290,346-358) ----------------------
// partial sums of one time slice of one row: block (x: slice, y: row = chan * npol + pol); fixed-order tree in LDS (deterministic)
__global__ __launch_bounds__(256) void k_rescale_sums_fpt(const float* __restrict__ in, const uint64_t ics, const uint64_t ips, const uint32_t npol,
                                                          const uint64_t ndat, const uint32_t per_block, const uint32_t ncol,
                                                          double* __restrict__ part_sum, double* __restrict__ part_sq)
{
  __shared__ double ss[256], sq[256];
  const uint32_t col = blockIdx.y, chan = col / npol, pol = col % npol;
  const float* __restrict__ x = in + chan * ics + pol * ips;
  const uint64_t i0 = (uint64_t)blockIdx.x * per_block, i1 = i0 + per_block < ndat ? i0 + per_block : ndat;
  double s = 0.0, q = 0.0;
  for (uint64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float v = x[i];
    s += (double)v;
    q += (double)__fmul_rn(v, v);                      // the reference squares in float and accumulates in double (:243-244)
  }
  ss[threadIdx.x] = s; sq[threadIdx.x] = q;
  __syncthreads();
  for (uint32_t w = 128; w; w >>= 1) {
    if (threadIdx.x < w) { ss[threadIdx.x] += ss[threadIdx.x + w]; sq[threadIdx.x] += sq[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_sum[(uint64_t)blockIdx.x * ncol + col] = ss[0]; part_sq[(uint64_t)blockIdx.x * ncol + col] = sq[0]; }
}

__global__ __launch_bounds__(256) void k_rescale_apply_fpt(const float* in, const uint64_t ics, const uint64_t ips, float* out /* may alias in */,
                                                           const uint64_t ocs, const uint64_t ops, const uint32_t npol, const uint64_t n,
                                                           const float* __restrict__ offset, const float* __restrict__ scale)
{
  const uint32_t col = blockIdx.y, chan = col / npol, pol = col % npol;
  const float o = offset[col], sc = scale[col];
  const float* x = in + chan * ics + pol * ips;
  float* y = out + chan * ocs + pol * ops;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    y[i] = __fmul_rn(__fadd_rn(x[i], o), sc);          // Rescale.C:343
}

// SigProcDigitizer::pack, FPT branch, optionally behind Rescale's apply (RESC): a block transposes 64 time samples x 64 output
// channels of one polarisation through LDS -- rows are read along time (coalesced), bytes leave along the channels, in the TPF
// order of the file (outidx = idat*nchan*npol + ipol*nchan + ichan).  nbit -32: floats divided by the input scale (pack_float).
template <bool RESC>
__global__ __launch_bounds__(256) void k_digitize_fpt(const float* __restrict__ in, const uint64_t ics, const uint64_t ips, uint8_t* __restrict__ out,
                                                      const uint64_t ndat, const uint32_t nchan, const uint32_t npol, const int nbit,
                                                      const float* __restrict__ offset, const float* __restrict__ scale,
                                                      const float digi_scale, const float digi_mean, const float xpol_offset,
                                                      const int digi_max, const float fscale, const int flip_band, const int swap_band)
{
  __shared__ float tile[64][65];
  const uint32_t k0 = blockIdx.x * 64, ipol = blockIdx.z;
  const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // read: tx = time, ty + 4 j = channel of the tile
  // (time tiles beyond the grid's y limit: the block walks on; the trip count is the same for every thread of a block)
  for (uint64_t t0 = (uint64_t)blockIdx.y * 64; t0 < ndat; t0 += (uint64_t)gridDim.y * 64) {
  for (uint32_t j = 0; j < 16; j++) {
    const uint32_t k = k0 + ty + 4 * j;
    float v = 0.f;
    if (k < nchan && t0 + tx < ndat) {
      uint32_t ic = k;                                               // ChannelSort, SigProcDigitizer.C:52-65
      if (swap_band) ic = (ic + nchan / 2) % nchan;
      if (flip_band) ic = nchan - ic - 1;
      v = in[ic * ics + ipol * ips + t0 + tx];
      if (RESC) v = __fmul_rn(__fadd_rn(v, offset[ic * npol + ipol]), scale[ic * npol + ipol]);      // Rescale.C:343
    }
    tile[ty + 4 * j][tx] = v;
  }
  __syncthreads();
  const float mean = digi_mean + (ipol > 1 ? xpol_offset : 0.0f);
  auto level = [&](const float x) {
    // :253  float multiply, float add, then + 0.5 in double (the literal is a double), truncation; out-of-range and NaN
    // conversions yield INT_MIN on the reference's x86 hosts (cvttsd2si), i.e. clip to 0
    const double d = (double)__fadd_rn(__fmul_rn(x, digi_scale), mean) + 0.5;
    int r = (d >= 2147483648.0 || d <= -2147483649.0 || d != d) ? (int)0x80000000 : (int)d;
    r = r < 0 ? 0 : r;
    return r > digi_max ? digi_max : r;
  };
  if (nbit == -32) {
    for (uint32_t j = 0; j < 16; j++) {                              // write: tx = channel, ty + 4 j = time
      const uint32_t tt = ty + 4 * j, k = k0 + tx;
      if (k < nchan && t0 + tt < ndat) ((float*)out)[(t0 + tt) * nchan * npol + (uint64_t)ipol * nchan + k] = __fdiv_rn(tile[tx][tt], fscale);
    }
    __syncthreads();
    continue;
  }
  const uint32_t spb = nbit >= 8 ? 1 : 8 / nbit, units = 64 / spb;  // bytes (16-bit: samples) of the tile per time sample
  for (uint32_t u = threadIdx.x; u < 64 * units; u += 256) {
    const uint32_t tt = u / units, w = u % units, k = k0 + w * spb;
    if (k >= nchan || t0 + tt >= ndat) continue;
    uint32_t byte = 0;
    for (uint32_t j = 0; j < spb; j++) byte |= (uint32_t)level(tile[w * spb + j][tt]) << (j * (nbit >= 8 ? 0 : nbit));
    const uint64_t oidx = ((t0 + tt) * nchan * npol + (uint64_t)ipol * nchan + k) / spb;
    if (nbit == 16) ((uint16_t*)out)[oidx] = (uint16_t)byte;
    else out[oidx] = (uint8_t)byte;
  }
  __syncthreads();                                                   // the tile is free for the next time slice
  }
}

}  // namespace dspsr_amd

using namespace dspsr_amd;

extern "C" int dspsr_amd_pscrunch_tfp(dspsr_amd_ctx* ctx, const float* in_tfp_dev, float* out_tfp_dev, uint64_t ndat,
                                      uint32_t nchan, uint32_t npol)
{
  if (!ctx || ((!in_tfp_dev || !out_tfp_dev) && ndat)) return DSPSR_AMD_EINVAL;
  if (npol == 1) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::PScrunch::transformation invalid npol=%u", npol);   // :36-38
  if (in_tfp_dev == out_tfp_dev)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_pscrunch_tfp: in place is not supported on the device (use a second block)");
  if (!ndat || !nchan) return DSPSR_AMD_OK;
  const uint64_t n = ndat * nchan;
  uint64_t gb = (n + 255) / 256;
  if (gb > 8192) gb = 8192;
  const float scale = (float)(1.0 / sqrt(2.0));                        // PScrunch.C:52
  hipLaunchKernelGGL(k_pscrunch_tfp, dim3((uint32_t)gb), dim3(256), 0, ctx->stream, in_tfp_dev, out_tfp_dev, n, npol, scale);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_pscrunch_tfp: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

struct dspsr_amd_rescale {
  dspsr_amd_ctx* ctx;
  uint32_t nchan, npol, ncol;
  uint64_t nsample;        // interval in samples (0 => the length of the first block, Rescale.C:100-105)
  uint64_t isample = 0;
  bool first_call = true, constant = false;
  double *total = nullptr, *totalsq = nullptr, *part_sum = nullptr, *part_sq = nullptr;
  float *offset = nullptr, *scale = nullptr;
  uint32_t part_cap = 0;
};

extern "C" int dspsr_amd_rescale_create(dspsr_amd_ctx* ctx, uint32_t nchan, uint32_t npol, uint64_t interval_samples,
                                        int constant, dspsr_amd_rescale** out)
{
  if (!ctx || !out || !nchan || !npol) return DSPSR_AMD_EINVAL;
  dspsr_amd_rescale* r = new dspsr_amd_rescale;
  r->ctx = ctx;
  r->nchan = nchan; r->npol = npol; r->ncol = nchan * npol;
  r->nsample = interval_samples;
  r->constant = constant != 0;
  const size_t nd = (size_t)r->ncol * sizeof(double), nf = (size_t)r->ncol * sizeof(float);
  if (hipMalloc((void**)&r->total, nd) != hipSuccess || hipMalloc((void**)&r->totalsq, nd) != hipSuccess ||
      hipMalloc((void**)&r->offset, nf) != hipSuccess || hipMalloc((void**)&r->scale, nf) != hipSuccess) {
    dspsr_amd_rescale_destroy(r);
    return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_rescale_create: hipMalloc failed");
  }
  (void)hipMemsetAsync(r->total, 0, nd, ctx->stream);
  (void)hipMemsetAsync(r->totalsq, 0, nd, ctx->stream);
  *out = r;
  return DSPSR_AMD_OK;
}

extern "C" void dspsr_amd_rescale_destroy(dspsr_amd_rescale* r)
{
  if (!r) return;
  (void)hipStreamSynchronize(r->ctx->stream);
  if (r->total) (void)hipFree(r->total);
  if (r->totalsq) (void)hipFree(r->totalsq);
  if (r->part_sum) (void)hipFree(r->part_sum);
  if (r->part_sq) (void)hipFree(r->part_sq);
  if (r->offset) (void)hipFree(r->offset);
  if (r->scale) (void)hipFree(r->scale);
  delete r;
}

// Rescale::transformation over one block: statistics per interval segment, then `apply(start, n)` for the segment's samples
// with the offset / scale in force for it (the plain apply pass, or the fused apply + PScrunch + digitiser)
// fpt != nullptr: FPT rows (strides in floats) instead of a TFP block
struct FptRows { const float* base; uint64_t cs, ps; };
template <class Apply>
static int rescale_block(dspsr_amd_rescale* r, const float* in_tfp_dev, uint64_t ndat, const char* who, Apply&& apply, const FptRows* fpt = nullptr)
{
  dspsr_amd_ctx* ctx = r->ctx;
  if (!r->nsample) r->nsample = ndat;                       // Rescale::init: nsample = input->get_ndat()
  const uint32_t ncol = r->ncol;
  const uint32_t gx = (ncol + 255) / 256;
  uint64_t start = 0;
  do {                                                       // Rescale.C:217-380 (not `exact`)
    uint64_t end = ndat;
    const uint64_t interval_end = start + r->nsample - r->isample;
    if (interval_end < end) end = interval_end;
    const uint64_t n = end - start;
    // sums over [start, end): slices of time so that the chip is filled, then a fixed-order accumulation
    uint32_t rows_per_block = fpt ? 4096 : 64;
    uint32_t nslice = (uint32_t)((n + rows_per_block - 1) / rows_per_block);
    while (nslice > 4096) { rows_per_block *= 2; nslice = (uint32_t)((n + rows_per_block - 1) / rows_per_block); }
    if (nslice > r->part_cap) {
      if (r->part_sum) (void)hipFree(r->part_sum);
      if (r->part_sq) (void)hipFree(r->part_sq);
      r->part_sum = r->part_sq = nullptr;
      r->part_cap = 0;
      const size_t bytes = (size_t)nslice * ncol * sizeof(double);
      if (hipMalloc((void**)&r->part_sum, bytes) != hipSuccess || hipMalloc((void**)&r->part_sq, bytes) != hipSuccess)
        return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "%s: hipMalloc of the partial sums failed", who);
      r->part_cap = nslice;
    }
    if (fpt) {
      hipLaunchKernelGGL(k_rescale_sums_fpt, dim3(nslice, ncol), dim3(256), 0, ctx->stream, fpt->base + start, fpt->cs, fpt->ps, r->npol, n,
                         rows_per_block, ncol, r->part_sum, r->part_sq);
    } else {
      const float* seg = in_tfp_dev + start * ncol;
      hipLaunchKernelGGL(k_rescale_sums, dim3(gx, nslice), dim3(256), 0, ctx->stream, seg, n, ncol, rows_per_block,
                         r->part_sum, r->part_sq);
    }
    hipLaunchKernelGGL(k_rescale_accumulate, dim3(gx), dim3(256), 0, ctx->stream, r->part_sum, r->part_sq, nslice, ncol,
                       r->total, r->totalsq);
    r->isample += n;
    if (r->isample == r->nsample || r->first_call) {        // :303-326
      hipLaunchKernelGGL(k_rescale_update, dim3(gx), dim3(256), 0, ctx->stream, r->total, r->totalsq, ncol,
                         (double)r->isample, (!r->constant || r->first_call) ? 1 : 0, r->offset, r->scale);
      r->isample = 0;
      r->first_call = false;
    }
    apply(start, n);
    start = end;
  } while (start < ndat);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "%s: %s", who, hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_rescale_transform(dspsr_amd_rescale* r, const float* in_tfp_dev, float* out_tfp_dev, uint64_t ndat)
{
  if (!r || (!in_tfp_dev && ndat) || (!out_tfp_dev && ndat)) return DSPSR_AMD_EINVAL;
  if (!ndat) return DSPSR_AMD_OK;
  const uint32_t ncol = r->ncol;
  return rescale_block(r, in_tfp_dev, ndat, "dspsr_amd_rescale_transform", [&](const uint64_t start, const uint64_t n) {
    const uint64_t nel = n * ncol;
    uint64_t gb = (nel + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(k_rescale_apply, dim3((uint32_t)gb), dim3(256), 0, r->ctx->stream, in_tfp_dev + start * ncol,
                       out_tfp_dev + start * ncol, nel, ncol, r->offset, r->scale);
  });
}

extern "C" int dspsr_amd_rescale_get(dspsr_amd_rescale* r, float* offset_host, float* scale_host)
{
  if (!r) return DSPSR_AMD_EINVAL;
  hipError_t e = hipSuccess;
  if (offset_host) e = hipMemcpyAsync(offset_host, r->offset, r->ncol * sizeof(float), hipMemcpyDeviceToHost, r->ctx->stream);
  if (e == hipSuccess && scale_host)
    e = hipMemcpyAsync(scale_host, r->scale, r->ncol * sizeof(float), hipMemcpyDeviceToHost, r->ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(r->ctx->stream);
  if (e != hipSuccess) return ctx_fail(r->ctx, DSPSR_AMD_EHIP, "dspsr_amd_rescale_get: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

// digitiser constants, SigProcDigitizer.C:112-158
struct DigiParams { float mean, scale, xpol_offset; int max; };
static DigiParams digi_params(int nbit, int use_digi_scales, double input_scale, float scale_fac)
{
  DigiParams d = {0.f, 0.f, 0.f, 0};
  const float digi_sigma = 6.f;
  switch (nbit) {                                            // :112-143
    case 1: d.mean = 0.5f; d.scale = 1.f; d.max = 1; break;
    case 2: d.mean = 1.5f; d.scale = 1.f; d.max = 3; break;
    case 4: d.mean = 7.5f; d.scale = d.mean / digi_sigma; d.max = 15; break;
    case 8: d.mean = 127.5f; d.scale = d.mean / digi_sigma; d.max = 255; break;
    case 16: d.mean = 32768.0f; d.scale = d.mean / digi_sigma; d.max = 65535; break;
  }
  if (!use_digi_scales) { d.xpol_offset = d.mean; d.mean = 0.f; d.scale = 1.f; }          // :148-154
  d.scale = (float)((double)d.scale / (input_scale * (double)scale_fac));                // :158 (get_scale() is a double)
  return d;
}

// digifil's output stage behind the TFP filterbank in ONE pass over the detected block (LoadToFil.C:318-362): Rescale's
// (x + offset) * scale per polarisation (Rescale.C:352), PScrunch's (p0 + p1) * float(1/sqrt 2) (PScrunch.C:52,72-90) and the
// digitiser (SigProcDigitizer.C:160-236), the same float operations in the same order as the three kernels above -- the
// rescaled block and the intensity block are not written (200 of the 343 MB the separate passes move per 67 MB block).
// One thread per output byte (sub-byte: 8/nbit consecutive output channels; 16-bit: per sample); blockIdx.y walks the rows.
template <int SPB>       // samples per output byte (1 for 8 and 16 bits)
__global__ __launch_bounds__(256) void k_rescale_pscrunch_digitize(const float* __restrict__ in, uint8_t* __restrict__ out,
                                                                   const uint32_t ndat, const uint32_t nchan, const int nbit,
                                                                   const float* __restrict__ offset, const float* __restrict__ scale,
                                                                   const float pscale, const float digi_scale, const float digi_mean,
                                                                   const int digi_max, const int flip_band, const int swap_band)
{
  constexpr uint32_t spb = SPB;
  const uint32_t units = nchan / spb;                            // bytes (16-bit: samples) per time sample
  const uint32_t w = blockIdx.x * 256 + threadIdx.x;
  if (w >= units) return;
  // the thread's channels and their offset / scale are the same for every row
  uint32_t ic[SPB];
  float o0[SPB], o1[SPB], s0[SPB], s1[SPB];
#pragma unroll
  for (uint32_t j = 0; j < spb; j++) {
    uint32_t c = w * spb + j;                                    // ChannelSort, SigProcDigitizer.C:52-65
    if (swap_band) c = (c + nchan / 2) % nchan;
    if (flip_band) c = nchan - c - 1;
    ic[j] = c;
    o0[j] = offset[2 * c]; o1[j] = offset[2 * c + 1];
    s0[j] = scale[2 * c]; s1[j] = scale[2 * c + 1];
  }
  for (uint32_t idat = blockIdx.y; idat < ndat; idat += gridDim.y) {
    const float2* __restrict__ row = (const float2*)(in + (uint64_t)idat * nchan * 2);
    uint32_t byte = 0;
#pragma unroll
    for (uint32_t j = 0; j < spb; j++) {
      const float2 p = row[ic[j]];
      const float a = __fmul_rn(__fadd_rn(p.x, o0[j]), s0[j]), b = __fmul_rn(__fadd_rn(p.y, o1[j]), s1[j]);     // Rescale.C:352
      const float x = __fmul_rn(__fadd_rn(a, b), pscale);                                                       // PScrunch.C:84
      const double d = (double)__fadd_rn(__fmul_rn(x, digi_scale), digi_mean) + 0.5;                            // :198
      int r = (d >= 2147483648.0 || d <= -2147483649.0 || d != d) ? (int)0x80000000 : (int)d;
      r = r < 0 ? 0 : r;
      r = r > digi_max ? digi_max : r;
      byte |= (uint32_t)r << (j * (nbit >= 8 ? 0 : nbit));
    }
    if (nbit == 16) ((uint16_t*)out)[(uint64_t)idat * units + w] = (uint16_t)byte;
    else out[(uint64_t)idat * units + w] = (uint8_t)byte;
  }
}

extern "C" int dspsr_amd_rescale_pscrunch_digitize(dspsr_amd_rescale* r, const float* in_tfp_dev, uint64_t ndat, int nbit,
                                                   float scale_fac, int flip_band, int swap_band, void* out_dev)
{
  if (!r || ((!in_tfp_dev || !out_dev) && ndat)) return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = r->ctx;
  if (r->npol != 2)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_rescale_pscrunch_digitize: two polarisations in (PPQQ), npol=%u", r->npol);
  if (nbit != 1 && nbit != 2 && nbit != 4 && nbit != 8 && nbit != 16)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::SigProcDigitizer::set_nbit nbit=%i not understood", nbit);
  if (nbit < 8 && r->nchan % (8 / nbit))
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_rescale_pscrunch_digitize: nchan=%u not a multiple of %d samples per byte",
                    r->nchan, 8 / nbit);
  if (!ndat) return DSPSR_AMD_OK;
  const DigiParams d = digi_params(nbit, 1, 1.0, scale_fac);        // behind Rescale the input scale is 1 (Rescale.C:204)
  const uint32_t nchan = r->nchan, spb = nbit >= 8 ? 1 : 8 / nbit, units = nchan / spb;
  const float pscale = (float)(1.0 / sqrt(2.0));                     // PScrunch.C:52
  const size_t out_row = (size_t)units * (nbit == 16 ? 2 : 1);
  return rescale_block(r, in_tfp_dev, ndat, "dspsr_amd_rescale_pscrunch_digitize", [&](const uint64_t start, const uint64_t n) {
    for (uint64_t s0 = 0; s0 < n; s0 += 1u << 30) {                 // (32-bit row counter in the kernel)
      const uint32_t rows = (uint32_t)(n - s0 < (1u << 30) ? n - s0 : (1u << 30));
      const uint32_t gy = rows < 8192 ? rows : 8192;
      auto k = spb == 1 ? k_rescale_pscrunch_digitize<1> : spb == 2 ? k_rescale_pscrunch_digitize<2>
               : spb == 4 ? k_rescale_pscrunch_digitize<4> : k_rescale_pscrunch_digitize<8>;
      hipLaunchKernelGGL(k, dim3((units + 255) / 256, gy), dim3(256), 0, ctx->stream,
                         in_tfp_dev + (start + s0) * (uint64_t)nchan * 2, (uint8_t*)out_dev + (start + s0) * out_row, rows, nchan, nbit,
                         r->offset, r->scale, pscale, d.scale, d.mean, d.max, flip_band, swap_band);
    }
  });
}

extern "C" int dspsr_amd_sigproc_digitize(dspsr_amd_ctx* ctx, const float* in_tfp_dev, uint64_t ndat, uint32_t nchan,
                                          uint32_t npol, int nbit, int use_digi_scales, double input_scale, float scale_fac,
                                          int flip_band, int swap_band, void* out_dev)
{
  if (!ctx || ((!in_tfp_dev || !out_dev) && ndat)) return DSPSR_AMD_EINVAL;
  if (nbit != 1 && nbit != 2 && nbit != 4 && nbit != 8 && nbit != 16 && nbit != -32)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::SigProcDigitizer::set_nbit nbit=%i not understood", nbit);
  if (!ndat) return DSPSR_AMD_OK;
  if (nbit == -32) {                                         // pack_float: TPF floats divided by the input scale
    uint64_t gb = (ndat * nchan * npol + 255) / 256;
    if (gb > 8192) gb = 8192;
    hipLaunchKernelGGL(k_sigproc_float, dim3((uint32_t)gb), dim3(256), 0, ctx->stream, in_tfp_dev, (float*)out_dev, ndat, nchan,
                       npol, (float)input_scale, flip_band, swap_band);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_sigproc_digitize: %s", hipGetErrorString(e));
    return DSPSR_AMD_OK;
  }
  if (nbit < 8 && nchan % (8 / nbit))
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_sigproc_digitize: nchan=%u not a multiple of %d samples per byte", nchan, 8 / nbit);
  const DigiParams dp = digi_params(nbit, use_digi_scales, input_scale, scale_fac);
  const float digi_mean = dp.mean, digi_scale = dp.scale, xpol_offset = dp.xpol_offset;
  const int digi_max = dp.max;
  const uint32_t spb = nbit >= 8 ? 1 : 8 / nbit;
  const uint64_t units = ndat * npol * (nchan / spb);
  uint64_t gb = (units + 255) / 256;
  if (gb > 8192) gb = 8192;
  hipLaunchKernelGGL(k_sigproc_digitize, dim3((uint32_t)gb), dim3(256), 0, ctx->stream, in_tfp_dev, (uint8_t*)out_dev, ndat, nchan,
                     npol, nbit, digi_scale, digi_mean, xpol_offset, digi_max, flip_band, swap_band);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_sigproc_digitize: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}


static hipError_t launch_digitize_fpt(dspsr_amd_ctx* ctx, bool resc, const float* in, uint64_t ics, uint64_t ips, uint8_t* out, uint64_t ndat,
                                      uint32_t nchan, uint32_t npol, int nbit, const float* offset, const float* scale, const DigiParams& d,
                                      float fscale, int flip_band, int swap_band)
{
  const uint64_t ty = (ndat + 63) / 64;
  const dim3 grid((nchan + 63) / 64, (uint32_t)(ty > 65535 ? 65535 : ty), npol);
  if (resc) hipLaunchKernelGGL(k_digitize_fpt<true>, grid, dim3(256), 0, ctx->stream, in, ics, ips, out, ndat, nchan, npol, nbit, offset, scale,
                               d.scale, d.mean, d.xpol_offset, d.max, fscale, flip_band, swap_band);
  else hipLaunchKernelGGL(k_digitize_fpt<false>, grid, dim3(256), 0, ctx->stream, in, ics, ips, out, ndat, nchan, npol, nbit, offset, scale,
                          d.scale, d.mean, d.xpol_offset, d.max, fscale, flip_band, swap_band);
  return hipGetLastError();
}

extern "C" int dspsr_amd_rescale_transform_fpt(dspsr_amd_rescale* r, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                               float* out_dev, uint64_t out_chan_stride, uint64_t out_pol_stride, uint64_t ndat)
{
  if (!r || ((!in_dev || !out_dev) && ndat)) return DSPSR_AMD_EINVAL;
  if (!ndat) return DSPSR_AMD_OK;
  if (r->ncol > 65535) return ctx_fail(r->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_rescale_transform_fpt: nchan*npol=%u exceeds the grid limit", r->ncol);
  const FptRows rows = {in_dev, in_chan_stride, in_pol_stride};
  return rescale_block(r, nullptr, ndat, "dspsr_amd_rescale_transform_fpt", [&](const uint64_t start, const uint64_t n) {
    uint64_t gb = (n + 255) / 256;
    if (gb > 256) gb = 256;
    hipLaunchKernelGGL(k_rescale_apply_fpt, dim3((uint32_t)gb, r->ncol), dim3(256), 0, r->ctx->stream, in_dev + start, in_chan_stride,
                       in_pol_stride, out_dev + start, out_chan_stride, out_pol_stride, r->npol, n, r->offset, r->scale);
  }, &rows);
}

static int digitize_fpt_check(dspsr_amd_ctx* ctx, const char* who, uint32_t nchan, uint32_t npol, int nbit, uint64_t ndat)
{
  if (nbit != 1 && nbit != 2 && nbit != 4 && nbit != 8 && nbit != 16 && nbit != -32)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::SigProcDigitizer::set_nbit nbit=%i not understood", nbit);
  if (nbit > 0 && nbit < 8 && nchan % (8 / nbit))
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "%s: nchan=%u not a multiple of %d samples per byte", who, nchan, 8 / nbit);
  if (npol > 65535) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "%s: npol=%u exceeds the grid limit", who, npol);
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_sigproc_digitize_fpt(dspsr_amd_ctx* ctx, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                              uint64_t ndat, uint32_t nchan, uint32_t npol, int nbit, int use_digi_scales, double input_scale,
                                              float scale_fac, int flip_band, int swap_band, void* out_dev)
{
  if (!ctx || ((!in_dev || !out_dev) && ndat)) return DSPSR_AMD_EINVAL;
  const int rc = digitize_fpt_check(ctx, "dspsr_amd_sigproc_digitize_fpt", nchan, npol, nbit, ndat);
  if (rc != DSPSR_AMD_OK) return rc;
  if (!ndat || !nchan || !npol) return DSPSR_AMD_OK;
  const DigiParams d = nbit == -32 ? DigiParams{0.f, 0.f, 0.f, 0} : digi_params(nbit, use_digi_scales, input_scale, scale_fac);
  const hipError_t e = launch_digitize_fpt(ctx, false, in_dev, in_chan_stride, in_pol_stride, (uint8_t*)out_dev, ndat, nchan, npol, nbit, nullptr,
                                           nullptr, d, (float)input_scale, flip_band, swap_band);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_sigproc_digitize_fpt: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_rescale_digitize_fpt(dspsr_amd_rescale* r, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                              uint64_t ndat, int nbit, float scale_fac, int flip_band, int swap_band, void* out_dev)
{
  if (!r || ((!in_dev || !out_dev) && ndat)) return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = r->ctx;
  if (nbit == -32) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_rescale_digitize_fpt: nbit -32 takes the separate operations");
  const int rc = digitize_fpt_check(ctx, "dspsr_amd_rescale_digitize_fpt", r->nchan, r->npol, nbit, ndat);
  if (rc != DSPSR_AMD_OK) return rc;
  if (!ndat) return DSPSR_AMD_OK;
  if (r->ncol > 65535) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_rescale_digitize_fpt: nchan*npol=%u exceeds the grid limit", r->ncol);
  const DigiParams d = digi_params(nbit, 1, 1.0, scale_fac);        // behind Rescale the input scale is 1 (Rescale.C:204)
  const size_t out_row = (size_t)r->nchan * r->npol * (nbit == 16 ? 2 : 1) / (nbit >= 8 ? 1 : 8 / nbit);
  const FptRows rows = {in_dev, in_chan_stride, in_pol_stride};
  return rescale_block(r, nullptr, ndat, "dspsr_amd_rescale_digitize_fpt", [&](const uint64_t start, const uint64_t n) {
    (void)launch_digitize_fpt(ctx, true, in_dev + start, in_chan_stride, in_pol_stride, (uint8_t*)out_dev + start * out_row, n, r->nchan, r->npol,
                              nbit, r->offset, r->scale, d, 1.0f, flip_band, swap_band);
  }, &rows);
}
