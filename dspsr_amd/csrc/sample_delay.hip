// dsp::SampleDelay for gfx950: integer-sample inter-channel delay correction (-K), the companion of the
// fractional-delay phase that host_prep.cpp adds to the chirp.
//
//   dsp::SampleDelay::build           (Signal/General/SampleDelay.C:52-102): zero_delay = largest delay of the function,
//       applied delay of a row = zero_delay - delay (relative) or delay itself (absolute); total_delay = largest
//       applied delay = samples lost at the end of every block (re-presented by InputBuffering, :106-118,146).
//   dsp::SampleDelay::transformation  (:123-195): out[chan][pol][i] = in[chan][pol][i + applied_delay], i < ndat - total.
//   Dedispersion::SampleDelay::match  (Signal/General/DedispersionSampleDelay.C:24-75) is host arithmetic and lives in
//       host_prep.cpp (dspsr_amd_dedispersion_sample_delays).
//
// The reference runs in place (LoadToFold1.C:617-618).  A row is shifted towards its start, so one workgroup per row
// walking the row in order, with a barrier between the loads and the stores of a chunk, is safe in place; out of place
// the rows are additionally cut into segments to fill the chip when there are few rows.
#include <vector>

#include "engine_internal.h"

namespace dspsr_amd {

constexpr uint32_t SD_THREADS = 256, SD_PER_THREAD = 8;

__global__ __launch_bounds__(SD_THREADS) void k_sample_delay(const float* in /* may be out */, const uint64_t ics, const uint64_t ips,
                                                             float* out, const uint64_t ocs, const uint64_t ops,
                                                             const uint32_t npol, const uint32_t ndim, const uint64_t nfloat,
                                                             const int64_t* __restrict__ applied, const uint64_t seg_floats)
{
  const uint32_t ichan = blockIdx.z, ipol = blockIdx.y;
  const float* f = in + ichan * ics + ipol * ips + (uint64_t)applied[ichan * npol + ipol] * ndim;
  float* t = out + ichan * ocs + ipol * ops;
  const uint64_t begin = (uint64_t)blockIdx.x * seg_floats;
  uint64_t end = begin + seg_floats;
  if (end > nfloat) end = nfloat;
  constexpr uint32_t CHUNK = SD_THREADS * SD_PER_THREAD;
  for (uint64_t base = begin; base < end; base += CHUNK) {
    float v[SD_PER_THREAD];
#pragma unroll
    for (uint32_t j = 0; j < SD_PER_THREAD; j++) {
      const uint64_t i = base + j * SD_THREADS + threadIdx.x;
      v[j] = i < end ? f[i] : 0.f;
    }
    __syncthreads();                 // in place: every load of this chunk precedes every store of it
#pragma unroll
    for (uint32_t j = 0; j < SD_PER_THREAD; j++) {
      const uint64_t i = base + j * SD_THREADS + threadIdx.x;
      if (i < end) t[i] = v[j];
    }
  }
}

}  // namespace dspsr_amd

using namespace dspsr_amd;

struct dspsr_amd_sample_delay {
  dspsr_amd_ctx* ctx;
  uint32_t nchan, npol;
  int64_t zero_delay = 0;
  uint64_t total_delay = 0;
  int64_t* applied = nullptr;   // device [nchan][npol]
};

extern "C" int dspsr_amd_sample_delay_create(dspsr_amd_ctx* ctx, uint32_t nchan, uint32_t npol, const int64_t* delays_host,
                                             int absolute, dspsr_amd_sample_delay** out)
{
  if (!ctx || !out || !nchan || !npol || !delays_host) return DSPSR_AMD_EINVAL;
  if (npol > 65535 || nchan > 65535)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_sample_delay_create: nchan=%u npol=%u exceed the grid limits", nchan, npol);
  const size_t n = (size_t)nchan * npol;
  std::vector<int64_t> applied(n);
  int64_t zero = 0;
  uint64_t total = 0;
  int64_t maxd = delays_host[0];
  for (size_t i = 0; i < n; i++) if (delays_host[i] > maxd) maxd = delays_host[i];
  if (absolute) {                                           // SampleDelay.C:60-73
    for (size_t i = 0; i < n; i++) {
      if (delays_host[i] < 0) return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::SampleDelay absolute delays must not be negative");
      applied[i] = delays_host[i];
    }
    total = (uint64_t)maxd;
  } else {                                                  // :75-99
    zero = maxd;
    for (size_t i = 0; i < n; i++) {
      applied[i] = zero - delays_host[i];
      if ((uint64_t)applied[i] > total) total = (uint64_t)applied[i];
    }
    // transformation() switches on zero_delay != 0 (:166-172): with zero_delay == 0 it applies the raw delays, which
    // are then all <= 0; only all-zero delays pass its assert.
    if (zero == 0)
      for (size_t i = 0; i < n; i++)
        if (delays_host[i] != 0)
          return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::SampleDelay relative delays with zero_delay == 0 must all be zero");
  }
  dspsr_amd_sample_delay* h = new dspsr_amd_sample_delay;
  h->ctx = ctx; h->nchan = nchan; h->npol = npol; h->zero_delay = zero; h->total_delay = total;
  if (hipMalloc((void**)&h->applied, n * sizeof(int64_t)) != hipSuccess) {
    delete h;
    return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_sample_delay_create: hipMalloc failed");
  }
  hipError_t e = hipMemcpyAsync(h->applied, applied.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // `applied` is a local
  if (e != hipSuccess) {
    (void)hipFree(h->applied);
    delete h;
    return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_sample_delay_create: %s", hipGetErrorString(e));
  }
  *out = h;
  return DSPSR_AMD_OK;
}

extern "C" void dspsr_amd_sample_delay_destroy(dspsr_amd_sample_delay* h)
{
  if (!h) return;
  (void)hipStreamSynchronize(h->ctx->stream);
  if (h->applied) (void)hipFree(h->applied);
  delete h;
}

extern "C" int64_t dspsr_amd_sample_delay_zero_delay(const dspsr_amd_sample_delay* h) { return h ? h->zero_delay : 0; }
extern "C" uint64_t dspsr_amd_sample_delay_total_delay(const dspsr_amd_sample_delay* h) { return h ? h->total_delay : 0; }

extern "C" int dspsr_amd_sample_delay_transform(dspsr_amd_sample_delay* h, const float* in_dev, uint64_t in_chan_stride,
                                                uint64_t in_pol_stride, float* out_dev, uint64_t out_chan_stride,
                                                uint64_t out_pol_stride, uint32_t ndim, uint64_t ndat_in, uint64_t* ndat_out)
{
  if (!h || !ndim) return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = h->ctx;
  const uint64_t nout = ndat_in < h->total_delay ? 0 : ndat_in - h->total_delay;      // SampleDelay.C:137-145
  if (ndat_out) *ndat_out = nout;
  if (!nout) return DSPSR_AMD_OK;
  if (!in_dev || !out_dev) return DSPSR_AMD_EINVAL;
  const uint64_t nfloat = nout * ndim;
  // In place means the SAME rows (LoadToFold1.C:617-618): one workgroup per row then loads every chunk before it stores
  // it.  Buffers that overlap in any other way would race between workgroups: refused.
  const bool inplace = in_dev == out_dev;
  if (inplace && (in_chan_stride != out_chan_stride || in_pol_stride != out_pol_stride))
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_sample_delay_transform: in place needs equal strides");
  if (!inplace) {
    const uint64_t in_span = (h->nchan - 1) * in_chan_stride + (h->npol - 1) * in_pol_stride + ndat_in * ndim;
    const uint64_t out_span = (h->nchan - 1) * out_chan_stride + (h->npol - 1) * out_pol_stride + nfloat;
    if (in_dev < out_dev + out_span && out_dev < in_dev + in_span)
      return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_sample_delay_transform: input and output overlap without being "
                                             "the same buffer");
  }
  constexpr uint64_t CHUNK = (uint64_t)SD_THREADS * SD_PER_THREAD;
  uint64_t nseg = 1;
  if (!inplace) {                                           // out of place: cut rows so that >= ~2048 workgroups exist
    const uint64_t rows = (uint64_t)h->nchan * h->npol;
    nseg = (2048 + rows - 1) / rows;
    const uint64_t max_seg = (nfloat + CHUNK - 1) / CHUNK;
    if (nseg > max_seg) nseg = max_seg;
    if (nseg > 65535) nseg = 65535;
    if (nseg < 1) nseg = 1;
  }
  uint64_t seg_floats = (nfloat + nseg - 1) / nseg;
  seg_floats = (seg_floats + CHUNK - 1) / CHUNK * CHUNK;
  nseg = (nfloat + seg_floats - 1) / seg_floats;
  hipLaunchKernelGGL(k_sample_delay, dim3((uint32_t)nseg, h->npol, h->nchan), dim3(SD_THREADS), 0, ctx->stream, in_dev,
                     in_chan_stride, in_pol_stride, out_dev, out_chan_stride, out_pol_stride, h->npol, ndim, nfloat, h->applied,
                     seg_floats);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_sample_delay_transform: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}
