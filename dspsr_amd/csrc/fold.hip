// dsp::Fold::Engine for gfx950 (Signal/Pulsar/dsp/Fold.h:249-312).
// Reference: host plan loop Fold.C:744-787, CPU accumulate Fold.C:835-891, CUDA twin FoldCUDA.cu
// (set_bin run-length builder :84-113, send_binplan :154-198, fold1bin* kernels :208-576).
//
// Differences by design (DESIGN.md "Fold"): the reference GPU kernels sum each run-length
// interval from zero and atomicAdd it into the profile (order-dependent float sums).  Here
// every (chan, pol, bin) accumulator is owned by exactly one thread that walks the bin's
// intervals in time order and adds sample by sample, i.e. the SAME association order as the
// CPU loop Fold.C:844-852 -- results are deterministic and bit-identical to the CPU fold of
// the same detected samples.
#include <algorithm>
#include <string.h>
#include <math.h>
#include <vector>

#include "engine_internal.h"
#include "fold_internal.h"

namespace dspsr_amd {


// Direct variant: every bin-owner thread reads its samples straight from global memory
// (used when nbin is too large for the chunked kernel).
template <int NDIM>
__global__ void k_fold_direct(const float* __restrict__ in, const uint64_t chan_stride, const uint64_t pol_stride,
                              float* __restrict__ prof, const uint64_t prof_span, const uint32_t nbin,
                              const uint32_t* __restrict__ bin_start, const Interval* __restrict__ iv)
{
  const uint32_t ipol = blockIdx.x, npol = gridDim.x, ichan = blockIdx.y;
  const float* __restrict__ row = in + ichan * chan_stride + ipol * pol_stride;
  float* __restrict__ out = prof + ((uint64_t)ichan * npol + ipol) * prof_span;
  for (uint32_t b = blockIdx.z + gridDim.z * threadIdx.x; b < nbin; b += gridDim.z * blockDim.x) {
    const uint32_t i0 = bin_start[b], i1 = bin_start[b + 1];
    if (i0 == i1) continue;
    float acc[NDIM];
#pragma unroll
    for (int d = 0; d < NDIM; d++) acc[d] = out[b * NDIM + d];
    for (uint32_t i = i0; i < i1; i++) {
      const Interval v = iv[i];
      const float* __restrict__ x = row + v.offset * NDIM;
      for (uint32_t h = 0; h < v.hits; h++)
#pragma unroll
        for (int d = 0; d < NDIM; d++) acc[d] += x[h * NDIM + d];
    }
#pragma unroll
    for (int d = 0; d < NDIM; d++) out[b * NDIM + d] = acc[d];
  }
}

// Chunked variant (the hot one).  One workgroup per (channel, pol) row.  The row is streamed
// through LDS in chunks of FOLD_CHUNK samples with fully coalesced 16-byte loads (the next
// chunk is already in flight in registers while the current one is folded); thread b owns phase
// bins b, b+blockDim, ... and walks each bin's time-ordered interval list with a cursor, adding
// the samples that fall inside the current chunk one by one.  Per (chan, pol, bin, dim) the adds
// therefore happen in time order, as in Fold.C:844-852.  With few (chan, pol) rows the bins of a row are dealt to
// gridDim.z workgroups (each streams the whole row: the re-reads come from L2 / Infinity Cache), so that the chip
// is filled without touching the order of any sum.
constexpr uint32_t FOLD_CHUNK = 2048;   // samples per chunk
constexpr int FOLD_BPT = 4;             // bins per thread (nbin <= FOLD_BPT * blockDim)
// LONG runs.  A sum in strict time order is one dependent chain of float adds per (chan, pol, bin, dim): with phase bins
// hundreds of samples wide only a couple of chains are alive per row and the kernel is bound by the add latency (10 ms per
// 4 M samples and row at -F 64:D, profiles/r02j).  When the plan holds a run of FOLD_LONG_RUN samples or more, the host
// picks the LONG variant for the whole call: every chunk is first reduced, by all threads, to the sums of its aligned
// FOLD_MB-sample micro-blocks (each summed in time order); a bin's owner then adds, in time order, single samples up to
// the first micro-block boundary inside its run, whole micro-block sums, and single samples after the last boundary.
// Deterministic (the association depends on the plan and the chunk grid only), no longer the association of the CPU
// loop: it agrees with it to float rounding, like the reference's own GPU fold (FoldCUDA.cu:208-269 sums each
// run from zero and adds the run sums atomically).  Plans of shorter runs keep the exact time-order kernel.
constexpr uint32_t FOLD_MB = 32;
constexpr uint32_t FOLD_LONG_RUN = FOLD_LONG_RUN_HOST;

// NROW: polarisation rows of one channel folded by the same workgroup (detected data with ndim < 4 lie in 4/ndim planes:
// the walk through the bin plan -- most of the work with 4-byte samples -- then serves all planes; the sums of every
// (chan, pol, bin, dim) keep their order).
template <int NDIM, bool LONG, int NROW>
__global__ __launch_bounds__(1024) void k_fold_chunked(const float* __restrict__ in, const uint64_t chan_stride,
                                                       const uint64_t pol_stride, float* __restrict__ prof,
                                                       const uint64_t prof_span, const uint32_t nbin,
                                                       const uint32_t* __restrict__ bin_start,
                                                       const Interval* __restrict__ iv, const uint64_t first,
                                                       const uint64_t last /* [first,last): sample span of the plan */,
                                                       float* __restrict__ part /* LONG: [seg][row][nbin][NDIM] partial sums */,
                                                       const uint32_t chunks_per_seg /* LONG: chunks per blockIdx.z */)
{
  extern __shared__ __attribute__((aligned(16))) float fold_lds[];   // NROW x FOLD_CHUNK * NDIM floats (+ micro-block sums, LONG)
  const uint32_t ipol = blockIdx.x * NROW, npol = gridDim.x * NROW, ichan = blockIdx.y;
  constexpr uint32_t RS = FOLD_CHUNK * NDIM;             // floats per row of the chunk image
  constexpr uint32_t MS = (FOLD_CHUNK / FOLD_MB) * NDIM;  // micro-block sums per row (LONG)
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const float* __restrict__ row = in + ichan * chan_stride + ipol * pol_stride;          // row r: + r * pol_stride
  // LONG: blockIdx.z is a TIME segment of the row (all bins), summed from zero into this segment's partial profile --
  // every sample of the row is read once; k_fold_combine adds the partials to the profile in time order.
  // exact: blockIdx.z deals the BINS of a row to gridDim.z workgroups, each streams the whole row and owns its sums.
  float* __restrict__ out = LONG ? part + (((uint64_t)blockIdx.z * gridDim.y + ichan) * npol + ipol) * nbin * NDIM
                                 : prof + ((uint64_t)ichan * npol + ipol) * prof_span;
  const uint64_t out_rs = LONG ? (uint64_t)nbin * NDIM : prof_span;                        // row r: + r * out_rs
  const uint32_t bz = LONG ? 0u : blockIdx.z, nz = LONG ? 1u : gridDim.z;
  constexpr uint32_t NF4 = FOLD_CHUNK * NDIM / 4;        // float4 per chunk
  constexpr uint32_t MAXR = NF4 / 256;                    // float4 per thread at the minimum block size (256)

  // cursor into each owned bin's interval list; the current and the following interval are kept in
  // registers (loaded long before they are needed) so the chunk loop never waits on global memory
  uint32_t cur[FOLD_BPT], end[FOLD_BPT];
  Interval v0[FOLD_BPT], v1[FOLD_BPT];
  float acc[FOLD_BPT][NROW][NDIM];
  bool touched[FOLD_BPT];
  // "no interval" = offset ~0.  Always load through the global pointer with a clamped index: selecting
  // between &iv[i] and a local would make the load generic (flat_load + full vmcnt/lgkmcnt drain).
  auto load_iv = [&](const uint32_t i, const bool valid) -> Interval {
    Interval t = iv[valid ? i : 0u];
    if (!valid) { t.offset = ~0ull; t.hits = 0u; }
    return t;
  };
#pragma unroll
  for (int j = 0; j < FOLD_BPT; j++) {
    const uint32_t b = bz + nz * (tid + j * nt);   // bins are dealt to the nz workgroups of a row
    cur[j] = end[j] = 0;
    if (b < nbin) { cur[j] = bin_start[b]; end[j] = bin_start[b + 1]; }
    if constexpr (LONG) {
      // first interval of the bin that reaches into this segment (intervals are time ordered: binary search)
      const uint64_t seg0 = first + (uint64_t)blockIdx.z * chunks_per_seg * FOLD_CHUNK;
      uint32_t lo_i = cur[j], hi_i = end[j];
      while (lo_i < hi_i) {
        const uint32_t mid = (lo_i + hi_i) >> 1;
        const Interval t = iv[mid];
        if (t.offset + t.hits <= seg0) lo_i = mid + 1; else hi_i = mid;
      }
      cur[j] = lo_i;
    }
    touched[j] = LONG ? (b < nbin) : (cur[j] != end[j]);
    v0[j] = load_iv(cur[j], cur[j] < end[j]);
    v1[j] = load_iv(cur[j] + 1, cur[j] + 1 < end[j]);
#pragma unroll
    for (int r = 0; r < NROW; r++)
#pragma unroll
      for (int d = 0; d < NDIM; d++) acc[j][r][d] = (!LONG && b < nbin && touched[j]) ? out[r * out_rs + b * NDIM + d] : 0.f;
  }

  // chunk c covers samples [first + c*FOLD_CHUNK, ...); rows are 16-byte aligned when first*NDIM % 4 == 0,
  // the host guarantees it by rounding `first` down
  const float* __restrict__ src0 = row + first * NDIM;
  const uint64_t nfl_total = (last - first) * NDIM;      // floats in the span
  const uint32_t nchunk_all = (uint32_t)((last - first + FOLD_CHUNK - 1) / FOLD_CHUNK);
  const uint32_t cbeg = LONG ? blockIdx.z * chunks_per_seg : 0u;
  const uint32_t nchunk = LONG ? (cbeg + chunks_per_seg < nchunk_all ? cbeg + chunks_per_seg : nchunk_all) : nchunk_all;
  float4 pre[NROW][MAXR];
  auto fetch = [&](uint32_t c) {
#pragma unroll
    for (int rw = 0; rw < NROW; rw++) {
      const float4* __restrict__ src = (const float4*)(src0 + rw * pol_stride);
#pragma unroll
      for (uint32_t r = 0; r < MAXR; r++) {
        const uint32_t q = tid + r * nt;
        const uint64_t k = (uint64_t)c * NF4 + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < NF4) {
          if (4 * k + 4 <= nfl_total) {
            v = src[k];
          } else if (4 * k < nfl_total) {                  // ragged end of the span: never read past it
            const float* t = (const float*)(src + k);
            const uint32_t n = (uint32_t)(nfl_total - 4 * k);
            v.x = t[0];
            if (n > 1) v.y = t[1];
            if (n > 2) v.z = t[2];
          }
        }
        pre[rw][r] = v;
      }
    }
  };
  if (cbeg < nchunk) fetch(cbeg);
  for (uint32_t c = cbeg; c < nchunk; c++) {
    __syncthreads();                                    // previous chunk fully consumed
#pragma unroll
    for (int rw = 0; rw < NROW; rw++)
#pragma unroll
      for (uint32_t r = 0; r < MAXR; r++)
        if (tid + r * nt < NF4) ((float4*)(fold_lds + rw * RS))[tid + r * nt] = pre[rw][r];
    if (c + 1 < nchunk) fetch(c + 1);
    __syncthreads();
    if constexpr (LONG) {                                 // level 1: sums of the aligned micro-blocks of this chunk
      float* mbs = fold_lds + NROW * RS;
      for (uint32_t q = tid; q < NROW * MS; q += nt) {
        const uint32_t rw = q / MS, qr = q - rw * MS, mb = qr / NDIM, d = qr % NDIM;
        float sacc = 0.f;
#pragma unroll 8
        for (uint32_t h = 0; h < FOLD_MB; h++) sacc += fold_lds[rw * RS + (mb * FOLD_MB + h) * NDIM + d];
        mbs[q] = sacc;
      }
      __syncthreads();
    }
    const uint64_t c0 = first + (uint64_t)c * FOLD_CHUNK, c1 = c0 + FOLD_CHUNK;
#pragma unroll
    for (int j = 0; j < FOLD_BPT; j++) {
      while (v0[j].offset < c1) {                        // `none` has offset ~0 and ends the walk
        const Interval v = v0[j];
        const uint64_t lo = v.offset > c0 ? v.offset : c0;
        const uint64_t hi = v.offset + v.hits < c1 ? v.offset + v.hits : c1;
        // index the __shared__ array directly (a generic pointer would turn these into flat loads
        // whose vmcnt(0) wait also drains the chunk prefetch)
        const uint32_t x0 = (uint32_t)(lo - c0) * NDIM;
        const uint32_t n = (uint32_t)(hi - lo);
        if constexpr (LONG) {
          const float* mbs = fold_lds + NROW * RS;
          const uint32_t s0 = (uint32_t)(lo - c0), s1 = s0 + n;                       // samples [s0, s1) of the chunk
          const uint32_t a0 = (s0 + FOLD_MB - 1) / FOLD_MB * FOLD_MB;                // first micro-block boundary >= s0
          const uint32_t a1 = s1 / FOLD_MB * FOLD_MB;                                // last boundary <= s1
          if (a0 >= a1) {                                                            // no whole micro-block inside the run
#pragma unroll 4
            for (uint32_t h = s0; h < s1; h++)
#pragma unroll
              for (int r = 0; r < NROW; r++)
#pragma unroll
                for (int d = 0; d < NDIM; d++) acc[j][r][d] += fold_lds[r * RS + h * NDIM + d];
          } else {
#pragma unroll 4
            for (uint32_t h = s0; h < a0; h++)
#pragma unroll
              for (int r = 0; r < NROW; r++)
#pragma unroll
                for (int d = 0; d < NDIM; d++) acc[j][r][d] += fold_lds[r * RS + h * NDIM + d];
#pragma unroll 4
            for (uint32_t mb = a0 / FOLD_MB; mb < a1 / FOLD_MB; mb++)
#pragma unroll
              for (int r = 0; r < NROW; r++)
#pragma unroll
                for (int d = 0; d < NDIM; d++) acc[j][r][d] += mbs[r * MS + mb * NDIM + d];
#pragma unroll 4
            for (uint32_t h = a1; h < s1; h++)
#pragma unroll
              for (int r = 0; r < NROW; r++)
#pragma unroll
                for (int d = 0; d < NDIM; d++) acc[j][r][d] += fold_lds[r * RS + h * NDIM + d];
          }
        } else {
#pragma unroll 4
          for (uint32_t h = 0; h < n; h++)
#pragma unroll
            for (int r = 0; r < NROW; r++)
#pragma unroll
              for (int d = 0; d < NDIM; d++) acc[j][r][d] += fold_lds[r * RS + x0 + h * NDIM + d];
        }
        if (v.offset + v.hits > c1) break;               // interval continues in the next chunk (or segment)
        cur[j]++;
        v0[j] = v1[j];
        v1[j] = load_iv(cur[j] + 1, cur[j] + 1 < end[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < FOLD_BPT; j++) {
    const uint32_t b = bz + nz * (tid + j * nt);
    if (b < nbin && touched[j])
#pragma unroll
      for (int r = 0; r < NROW; r++)
#pragma unroll
        for (int d = 0; d < NDIM; d++) out[r * out_rs + b * NDIM + d] = acc[j][r][d];
  }
}


// Dense variant (round 4).  The walk above takes its intervals from global memory; the loads are small, but the per-wave
// vector-memory counter is in order, so every `s_waitcnt` in front of an interval's use also waits for the NEXT chunk's
// prefetch, issued just before -- a workgroup has bytes in flight for part of its time only (0.50 of the HBM peak on the
// reference's fold benchmark).  When no phase bin receives more than ONE run of samples per chunk (always, once the folding
// period exceeds FOLD_CHUNK samples: Benchmark/fold.csh 2794, the headline's detected series 34816) the plan is a dense table
//   tab[chunk][bin] = first sample of the run inside the chunk | samples << 11      (0: none; runs are cut at chunk ends)
// built on the host from the same run-length plan.  A thread's entries for chunk c + 1 travel with the prefetch of chunk c + 1
// and are waited for together with it at the top of the next iteration: the add phase issues no vector-memory instruction and
// no wait, so the prefetch stays in flight across it.  Same bins per thread, same chunk grid, one run per (chunk, bin) in
// chunk order: every (chan, pol, bin, dim) sum has exactly the association of k_fold_chunked<., false, .> and of Fold.C:844-852.
template <int NDIM, int NROW>
__global__ __launch_bounds__(1024) void k_fold_dense(const float* __restrict__ in, const uint64_t chan_stride,
                                                     const uint64_t pol_stride, float* __restrict__ prof,
                                                     const uint64_t prof_span, const uint32_t nbin,
                                                     const uint32_t* __restrict__ tab, const uint64_t first, const uint64_t last)
{
  extern __shared__ __attribute__((aligned(16))) float fold_lds[];   // NROW x FOLD_CHUNK * NDIM floats
  const uint32_t ipol = blockIdx.x * NROW, npol = gridDim.x * NROW, ichan = blockIdx.y;
  constexpr uint32_t RS = FOLD_CHUNK * NDIM;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const float* __restrict__ row = in + ichan * chan_stride + ipol * pol_stride;
  float* __restrict__ out = prof + ((uint64_t)ichan * npol + ipol) * prof_span;
  const uint32_t bz = blockIdx.z, nz = gridDim.z;
  constexpr uint32_t NF4 = FOLD_CHUNK * NDIM / 4;
  constexpr uint32_t MAXR = NF4 / 256;
  uint32_t bin[FOLD_BPT];
  float acc[FOLD_BPT][NROW][NDIM];
#pragma unroll
  for (int j = 0; j < FOLD_BPT; j++) {
    bin[j] = bz + nz * (tid + j * nt);
#pragma unroll
    for (int r = 0; r < NROW; r++)
#pragma unroll
      for (int d = 0; d < NDIM; d++) acc[j][r][d] = bin[j] < nbin ? out[r * prof_span + bin[j] * NDIM + d] : 0.f;
  }
  const float* __restrict__ src0 = row + first * NDIM;
  const uint64_t nfl_total = (last - first) * NDIM;
  const uint32_t nchunk = (uint32_t)((last - first + FOLD_CHUNK - 1) / FOLD_CHUNK);
  float4 pre[NROW][MAXR];
  uint32_t tabn[FOLD_BPT];
  auto fetch = [&](const uint32_t c) {
#pragma unroll
    for (int rw = 0; rw < NROW; rw++) {
      const float4* __restrict__ src = (const float4*)(src0 + rw * pol_stride);
#pragma unroll
      for (uint32_t r = 0; r < MAXR; r++) {
        const uint32_t q = tid + r * nt;
        const uint64_t k = (uint64_t)c * NF4 + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < NF4) {
          if (4 * k + 4 <= nfl_total) {
            v = src[k];
          } else if (4 * k < nfl_total) {                  // ragged end of the span: never read past it
            const float* t = (const float*)(src + k);
            const uint32_t n = (uint32_t)(nfl_total - 4 * k);
            v.x = t[0];
            if (n > 1) v.y = t[1];
            if (n > 2) v.z = t[2];
          }
        }
        pre[rw][r] = v;
      }
    }
    const uint32_t* __restrict__ tc = tab + (uint64_t)c * nbin;
#pragma unroll
    for (int j = 0; j < FOLD_BPT; j++) tabn[j] = bin[j] < nbin ? tc[bin[j]] : 0u;
  };
  if (nchunk) fetch(0);
  for (uint32_t c = 0; c < nchunk; c++) {
    __syncthreads();                                    // previous chunk fully consumed
#pragma unroll
    for (int rw = 0; rw < NROW; rw++)
#pragma unroll
      for (uint32_t r = 0; r < MAXR; r++)
        if (tid + r * nt < NF4) ((float4*)(fold_lds + rw * RS))[tid + r * nt] = pre[rw][r];
    uint32_t tabc[FOLD_BPT];
#pragma unroll
    for (int j = 0; j < FOLD_BPT; j++) tabc[j] = tabn[j];
    if (c + 1 < nchunk) fetch(c + 1);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FOLD_BPT; j++) {
      const uint32_t n = tabc[j] >> 11, x0 = (tabc[j] & 2047u) * NDIM;
#pragma unroll 4
      for (uint32_t h = 0; h < n; h++)
#pragma unroll
        for (int r = 0; r < NROW; r++)
#pragma unroll
          for (int d = 0; d < NDIM; d++) acc[j][r][d] += fold_lds[r * RS + x0 + h * NDIM + d];
    }
  }
#pragma unroll
  for (int j = 0; j < FOLD_BPT; j++)
    if (bin[j] < nbin)
#pragma unroll
      for (int r = 0; r < NROW; r++)
#pragma unroll
        for (int d = 0; d < NDIM; d++) out[r * prof_span + bin[j] * NDIM + d] = acc[j][r][d];
}

// LONG: profile[row][bin][dim] += partial sums of the time segments, in time order
__global__ __launch_bounds__(256) void k_fold_combine(float* __restrict__ prof, const uint64_t prof_span, const float* __restrict__ part,
                                                      const uint32_t nrow, const uint32_t row_floats, const uint32_t nseg)
{
  const uint64_t n = (uint64_t)nrow * row_floats;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / row_floats, k = i - r * row_floats;
    float a = prof[r * prof_span + k];
    for (uint32_t sg = 0; sg < nseg; sg++) a += part[(uint64_t)sg * n + i];
    prof[r * prof_span + k] = a;
  }
}

// Zeroed samples (Fold.C:853-866, FoldCUDA.cu:415-470 fold1bin*hits): when the input carries zeroed (RFI-excised) samples
// the number of samples a bin really received differs from channel to channel, so hits[] is kept per channel and counts,
// for polarisation 0, the samples whose first float is not zero.  One thread per (channel, bin) walks the bin's intervals.
__global__ __launch_bounds__(256) void k_fold_count_hits(const float* __restrict__ in, const uint64_t chan_stride, const uint32_t ndim,
                                                         uint32_t* __restrict__ hits, const uint32_t nbin,
                                                         const uint32_t* __restrict__ bin_start, const Interval* __restrict__ iv)
{
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (b >= nbin) return;
  const float* __restrict__ row = in + c * chan_stride;
  uint32_t n = 0;
  for (uint32_t i = bin_start[b]; i < bin_start[b + 1]; i++) {
    const Interval v = iv[i];
    const float* __restrict__ x = row + v.offset * ndim;
    for (uint32_t h = 0; h < v.hits; h++) n += x[(uint64_t)h * ndim] != 0.0f;
  }
  hits[(uint64_t)c * nbin + b] += n;
}

// Four-pass fused fold, second half: thread (chan, bin) walks the bin's intervals in time order and adds the piece sums
// k_inv_b<., true> left for every Tt-sample segment the interval covers: piece A of a segment entered at its first kept
// sample, piece B of a segment entered behind its cut (fold_internal.h).  One dependent chain of float4 adds per
// (chan, bin), 1/Tt of the samples long.
__global__ __launch_bounds__(256) void k_fold_segsum(float* __restrict__ prof, const uint64_t prof_span, const uint32_t nbin,
                                                     const float4* __restrict__ msum, const uint32_t npart, const uint32_t nkeep,
                                                     const uint32_t nfilt_pos, const int logTt, const int logMa, const int logMb,
                                                     const uint32_t* __restrict__ bin_start, const Interval* __restrict__ iv)
{
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (b >= nbin) return;
  const uint32_t i0 = bin_start[b], i1 = bin_start[b + 1];
  if (i0 == i1) return;
  const int logNt = logMa - logTt;                        // tiles (blocks of Tt values of t1) per channel
  const uint32_t hi = nfilt_pos + nkeep;
  float4* __restrict__ pp = (float4*)(prof + (uint64_t)c * prof_span) + b;
  float4 acc = *pp;
  const float4* __restrict__ mc = msum + (((uint64_t)c * npart) << (logNt + logMb + 1));
  // The pieces are fetched eight at a time and added in time order: their addresses depend on the plan alone (one load per
  // iteration, with a 64-bit division in front of it, left this kernel at one memory round trip per segment: 180 us per block
  // at -F 64:D).  `next` steps through the segments of the bin's intervals; part and position advance without divisions.
  uint32_t i = i0;
  uint64_t idat = 0, end = 0;
  uint32_t part = 0, pos = 0;
  auto next = [&](uint64_t& a) -> bool {
    while (idat >= end) {
      if (i >= i1) return false;
      const Interval v = iv[i++];
      idat = v.offset;
      end = v.offset + v.hits;
      part = (uint32_t)(idat / nkeep);
      pos = (uint32_t)(idat - (uint64_t)part * nkeep) + nfilt_pos;
    }
    const uint32_t seg = pos >> logTt;                                          // position = seg*Tt + j = t1 + (t2 << logMa)
    const uint32_t first = (seg << logTt) > nfilt_pos ? (seg << logTt) : nfilt_pos;         // the segment's first kept position
    const uint32_t last = ((seg + 1) << logTt) < hi ? ((seg + 1) << logTt) : hi;           // one past its last kept position
    const uint32_t t1blk = seg & ((1u << logNt) - 1), t2 = seg >> logNt;
    a = (((((uint64_t)part << logNt) + t1blk) << logMb) + t2) * 2 + (pos == first ? 0 : 1);
    idat += last - pos;                                // to the segment's end (an interval that ends inside took piece A, which ends there)
    pos = last;
    if (pos >= hi) { part++; pos = nfilt_pos; }        // the interval goes on in the next part
    return true;
  };
  for (;;) {
    float4 q[8];
    int n = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      uint64_t a;
      if (n == u && next(a)) { q[u] = mc[a]; n = u + 1; }
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (u < n) { acc.x += q[u].x; acc.y += q[u].y; acc.z += q[u].z; acc.w += q[u].w; }
    if (n < 8) break;
  }
  *pp = acc;
}

}  // namespace dspsr_amd

using namespace dspsr_amd;

static void slot_free(PlanSlot& s)
{
  if (s.h_bin_start) (void)hipHostFree(s.h_bin_start);
  if (s.h_iv) (void)hipHostFree(s.h_iv);
  if (s.d_bin_start) (void)hipFree(s.d_bin_start);
  if (s.d_iv) (void)hipFree(s.d_iv);
  if (s.h_aux) (void)hipHostFree(s.h_aux);
  if (s.d_aux) (void)hipFree(s.d_aux);
  if (s.done) (void)hipEventDestroy(s.done);
  if (s.ready) (void)hipEventDestroy(s.ready);
  s = PlanSlot();
}

static bool slot_reserve(PlanSlot& s, size_t nbin1, size_t niv)
{
  if (!s.done && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return false;
  if (!s.ready && hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess) return false;
  if (nbin1 > s.bin_cap) {
    if (s.h_bin_start) (void)hipHostFree(s.h_bin_start);
    if (s.d_bin_start) (void)hipFree(s.d_bin_start);
    s.h_bin_start = nullptr; s.d_bin_start = nullptr; s.bin_cap = 0;
    if (hipHostMalloc((void**)&s.h_bin_start, nbin1 * sizeof(uint32_t)) != hipSuccess) return false;
    if (hipMalloc((void**)&s.d_bin_start, nbin1 * sizeof(uint32_t)) != hipSuccess) return false;
    s.bin_cap = nbin1;
  }
  if (niv > s.iv_cap) {
    const size_t n = niv + niv / 2 + 16;
    if (s.h_iv) (void)hipHostFree(s.h_iv);
    if (s.d_iv) (void)hipFree(s.d_iv);
    s.h_iv = nullptr; s.d_iv = nullptr; s.iv_cap = 0;
    if (hipHostMalloc((void**)&s.h_iv, n * sizeof(Interval)) != hipSuccess) return false;
    if (hipMalloc((void**)&s.d_iv, n * sizeof(Interval)) != hipSuccess) return false;
    s.iv_cap = n;
  }
  return true;
}

// Plans go to the device on the fold's own stream, not in the compute stream: there a pair of small host-to-device copies
// stood between the last kernel of one block and the first of the next (about 45 us of idle device per block in the kernel
// trace, 3.5 % of a cfg4 block), although the kernel that reads the plan is launched several passes later.  The pinned
// source and the device copy of a slot are only rewritten after its `done` event (recorded on the compute stream behind the
// consumer) has been waited for on the host.
struct PlanCopy { void* dst; const void* src; size_t bytes; };
static hipError_t plan_upload(dspsr_amd_fold* f, PlanSlot& sl, const PlanCopy* c, const int n)
{
  hipError_t e = hipSuccess;
  if (!f->upload) e = hipStreamCreateWithFlags(&f->upload, hipStreamNonBlocking);
  for (int i = 0; i < n && e == hipSuccess; i++)
    if (c[i].bytes) e = hipMemcpyAsync(c[i].dst, c[i].src, c[i].bytes, hipMemcpyHostToDevice, f->upload);
  if (e == hipSuccess) e = hipEventRecord(sl.ready, f->upload);
  return e;
}
int fold_plan_wait(dspsr_amd_fold* f, PlanSlot* slot)
{
  const hipError_t e = hipStreamWaitEvent(f->ctx->stream, slot->ready, 0);
  if (e != hipSuccess) return ctx_fail(f->ctx, DSPSR_AMD_EHIP, "fold: plan wait: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_create(dspsr_amd_ctx* ctx, dspsr_amd_fold** out)
{
  if (!ctx || !out) return DSPSR_AMD_EINVAL;
  dspsr_amd_fold* f = new dspsr_amd_fold;
  f->ctx = ctx;
  *out = f;
  return DSPSR_AMD_OK;
}

extern "C" void dspsr_amd_fold_destroy(dspsr_amd_fold* f)
{
  if (!f) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  if (f->profile && !f->bound) (void)hipFree(f->profile);
  if (f->part) (void)hipFree(f->part);
  if (f->upload) { (void)hipStreamSynchronize(f->upload); (void)hipStreamDestroy(f->upload); }
  slot_free(f->slot[0]);
  slot_free(f->slot[1]);
  delete f;
}

static int fold_check_shape(dspsr_amd_fold* f, const char* who, uint32_t nchan, uint32_t npol, uint32_t ndim, uint32_t nbin)
{
  if (ndim != 1 && ndim != 2 && ndim != 4)
    return ctx_fail(f->ctx, DSPSR_AMD_EINVAL, "%s: ndim=%u not in {1,2,4}", who, ndim);
  if (!nchan || !npol || !nbin) return ctx_fail(f->ctx, DSPSR_AMD_EINVAL, "%s: zero dimension", who);
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_set_shape(dspsr_amd_fold* f, uint32_t nchan, uint32_t npol, uint32_t ndim, uint32_t nbin)
{
  if (!f) return DSPSR_AMD_EINVAL;
  const int rc = fold_check_shape(f, "dspsr_amd_fold_set_shape", nchan, npol, ndim, nbin);
  if (rc != DSPSR_AMD_OK) return rc;
  if (f->bound) {
    if (nchan == f->nchan && npol == f->npol && ndim == f->ndim && nbin == f->nbin) return DSPSR_AMD_OK;
    return ctx_fail(f->ctx, DSPSR_AMD_ESTATE, "dspsr_amd_fold_set_shape: the profile is bound to a caller's buffer of another "
                                              "shape; bind again (dspsr_amd_fold_bind_profile)");
  }
  const size_t need = (size_t)nchan * npol * nbin * ndim;
  if (need != f->profile_floats) {
    // PhaseSeries::mixable/resize on a changed shape starts a new, zeroed profile (Fold.C:495-508)
    (void)hipStreamSynchronize(f->ctx->stream);
    if (f->profile) (void)hipFree(f->profile);
    f->profile = nullptr;
    f->profile_floats = 0;
    if (hipMalloc((void**)&f->profile, need * sizeof(float)) != hipSuccess)
      return ctx_fail(f->ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_fold_set_shape: hipMalloc(%zu floats) failed", need);
    f->profile_floats = need;
    (void)hipMemsetAsync(f->profile, 0, need * sizeof(float), f->ctx->stream);
  }
  f->nchan = nchan; f->npol = npol; f->ndim = ndim; f->nbin = nbin;
  f->span = (uint64_t)nbin * ndim;
  return DSPSR_AMD_OK;
}

// Fold::Engine::setup (Fold.C:968-1011) caches output = get_profiles()->get_datptr(0,0) and output_span =
// get_nfloat_span(): the engine folds INTO the device PhaseSeries that Fold::get_output() hands to prepare_output /
// zero / mixable.  The buffer stays the caller's (never freed or zeroed here except through dspsr_amd_fold_zero).
extern "C" int dspsr_amd_fold_bind_profile(dspsr_amd_fold* f, float* profile_dev, uint64_t span_floats, uint32_t nchan,
                                           uint32_t npol, uint32_t ndim, uint32_t nbin)
{
  if (!f) return DSPSR_AMD_EINVAL;
  if (!profile_dev) {                                      // unbind: back to a library-owned profile of the same shape
    if (f->bound) { f->profile = nullptr; f->bound = false; f->profile_floats = 0; }
    return nchan ? dspsr_amd_fold_set_shape(f, nchan, npol, ndim, nbin) : DSPSR_AMD_OK;
  }
  const int rc = fold_check_shape(f, "dspsr_amd_fold_bind_profile", nchan, npol, ndim, nbin);
  if (rc != DSPSR_AMD_OK) return rc;
  if (span_floats < (uint64_t)nbin * ndim)
    return ctx_fail(f->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_fold_bind_profile: span=%llu floats < nbin*ndim=%llu",
                    (unsigned long long)span_floats, (unsigned long long)nbin * ndim);
  if (f->profile && !f->bound) {
    (void)hipStreamSynchronize(f->ctx->stream);
    (void)hipFree(f->profile);
  }
  f->profile = profile_dev;
  f->bound = true;
  f->profile_floats = 0;
  f->span = span_floats;
  f->nchan = nchan; f->npol = npol; f->ndim = ndim; f->nbin = nbin;
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_set_nbin(dspsr_amd_fold* f, uint32_t nbin)   // FoldCUDA.cu:64-70
{
  if (!f) return DSPSR_AMD_EINVAL;
  f->current_bin = f->folding_nbin = nbin;
  f->current_hits = 0;
  f->ndat_fold = 0;
  f->binplan.clear();
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_set_ndat(dspsr_amd_fold* f, uint64_t ndat, uint64_t /*idat_start*/)  // :72-82
{
  if (!f) return DSPSR_AMD_EINVAL;
  if (f->binplan.capacity() < ndat) f->binplan.reserve(ndat < (1u << 20) ? ndat : (1u << 20));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_set_bin(dspsr_amd_fold* f, uint64_t idat, double d_ibin, double /*bins_per_sample*/)
{
  if (!f) return DSPSR_AMD_EINVAL;
  const uint32_t ibin = (uint32_t)d_ibin;                 // FoldCUDA.cu:87
  if (ibin != f->current_bin) {
    if (!f->binplan.empty()) f->binplan.back().hits = f->current_hits;
    RunBin start; start.offset = idat; start.ibin = ibin; start.hits = 0;
    f->binplan.push_back(start);
    f->current_bin = ibin;
    f->current_hits = 0;
  }
  f->ndat_fold++;
  f->current_hits++;
  return DSPSR_AMD_OK;
}

// The plan loop of Fold.C:744-787, the same double recurrence sample by sample (the sums are not associative, so nothing is
// skipped), written so that the loop-carried chain is the one addition: phi -= floor(phi) changes phi only when phi is
// outside [0, 1) (phi - 0.0 == phi), the run-length bookkeeping of set_bin (FoldCUDA.cu:84-113) is inlined and hits[] is
// updated once per run.  At 6 MHz output rates the plan loop is what a block waits for.
// Weights (Fold.C:686-716,746-763): sample idat belongs to weight (idat + weight_idat) / ndatperweight; samples of a zero
// weight are dropped from the plan (binplan = folding_nbin in the reference: not folded, not counted in hits or
// ndat_folded) -- the hook a weighted input (dropped packets, RFI flagging upstream) needs; weights == NULL: none.
static int fold_set_bins_impl(dspsr_amd_fold* f, double phi, double phase_per_sample, uint64_t ndat, uint64_t idat_start,
                              const uint32_t* weights, uint64_t nweights, uint64_t ndatperweight, uint64_t weight_idat,
                              uint32_t* hits_host, uint64_t* ndat_folded)
{
  if (!f) return DSPSR_AMD_EINVAL;
  if (!f->folding_nbin) return ctx_fail(f->ctx, DSPSR_AMD_ESTATE, "dsp::Fold::fold nbin not set");
  const double double_nbin = (double)f->folding_nbin;     // Fold.C:719
  const uint32_t nbin = f->folding_nbin;
  uint32_t cur_bin = f->current_bin, cur_hits = f->current_hits;
  uint32_t counted = cur_hits;          // hits of the open run already added to hits_host by an earlier call
  const uint64_t end = idat_start + ndat;
  uint64_t folded = 0, iweight = 0, idat_nextweight = ~0ull;
  bool bad = false;
  if (weights && ndatperweight) {                         // Fold.C:686-716
    iweight = (idat_start + weight_idat) / ndatperweight;
    idat_nextweight = (iweight + 1) * ndatperweight - weight_idat;
    if (iweight >= nweights)
      return ctx_fail(f->ctx, DSPSR_AMD_ESTATE, "dsp::Fold::fold iweight=%llu >= nweight=%llu", (unsigned long long)iweight,
                      (unsigned long long)nweights);
    bad = weights[iweight] == 0;
  }
  // cur_bin == nbin (the value set_nbin leaves, FoldCUDA.cu:64-70) means "no run open": the last entry of the plan, if
  // any, already has its final hits
  auto close_run = [&]() {              // a dropped sample ends the open run: the next kept sample starts a new one
    if (cur_bin < nbin) {
      if (!f->binplan.empty()) f->binplan.back().hits = cur_hits;
      if (hits_host) hits_host[cur_bin] += cur_hits - counted;
    }
    cur_bin = nbin; cur_hits = 0; counted = 0;
  };
  // (a run costs the short cut about 70 ns, a sample of the loop 1 ns: it pays from runs of a hundred samples on -- the headline's
  //  34-sample runs keep the loop)
  if ((!weights || !ndatperweight) && phase_per_sample >= 0.0 && phase_per_sample * double_nbin * 128.0 <= 1.0) {
    // No weights, long runs: the plan run by run (host_prep.cpp fold_plan_run: the values of the recurrence below without walking the samples of
    // a run -- a 50 MHz channel's block is 10 M samples and 1100 runs).  Same bookkeeping as the sample loop: a new run where the bin
    // changes, hits[] once per run.
    uint64_t idat = idat_start;
    while (idat < end) {
      uint32_t ibin;
      uint64_t n = fold_plan_run(&phi, phase_per_sample, double_nbin, end - idat, &ibin);
      if (ibin >= nbin) {
        f->current_bin = cur_bin; f->current_hits = cur_hits;
        return ctx_fail(f->ctx, DSPSR_AMD_EINVAL, "dsp::Fold::fold ibin=%u >= nbin=%u", ibin, nbin);
      }
      if (ibin != cur_bin) {
        if (cur_bin < nbin) {
          if (!f->binplan.empty()) f->binplan.back().hits = cur_hits;
          if (hits_host) hits_host[cur_bin] += cur_hits - counted;
        }
        RunBin start; start.offset = idat; start.ibin = ibin; start.hits = 0;
        f->binplan.push_back(start);
        cur_bin = ibin;
        cur_hits = 0;
        counted = 0;
      }
      cur_hits += (uint32_t)n;
      folded += n;
      idat += n;
    }
  } else
  for (uint64_t idat = idat_start; idat < end; idat++) {
    if (idat >= idat_nextweight) {                        // Fold.C:746-763
      iweight++;
      if (iweight >= nweights) {
        f->current_bin = cur_bin; f->current_hits = cur_hits;
        return ctx_fail(f->ctx, DSPSR_AMD_ESTATE, "dsp::Fold::fold iweight=%llu >= nweight=%llu", (unsigned long long)iweight,
                        (unsigned long long)nweights);
      }
      bad = weights[iweight] == 0;
      idat_nextweight += ndatperweight;
    }
    if (!(phi >= 0.0 && phi < 1.0)) phi -= floor(phi);
    const uint32_t ibin = (uint32_t)(phi * double_nbin);
    phi += phase_per_sample;
    if (ibin >= nbin) {
      f->current_bin = cur_bin; f->current_hits = cur_hits;
      return ctx_fail(f->ctx, DSPSR_AMD_EINVAL, "dsp::Fold::fold ibin=%u >= nbin=%u", ibin, nbin);
    }
    if (bad) {
      close_run();
      continue;
    }
    if (ibin != cur_bin) {                // set_bin: a new run starts
      if (cur_bin < nbin) {               // (an open run ends here)
        if (!f->binplan.empty()) f->binplan.back().hits = cur_hits;
        if (hits_host) hits_host[cur_bin] += cur_hits - counted;
      }
      RunBin start; start.offset = idat; start.ibin = ibin; start.hits = 0;
      f->binplan.push_back(start);
      cur_bin = ibin;
      cur_hits = 0;
      counted = 0;
    }
    cur_hits++;
    folded++;
  }
  if (hits_host && cur_bin < nbin) hits_host[cur_bin] += cur_hits - counted;
  f->ndat_fold += folded;
  f->current_bin = cur_bin;
  f->current_hits = cur_hits;
  if (ndat_folded) *ndat_folded = folded;
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_set_bins(dspsr_amd_fold* f, double phi, double phase_per_sample, uint64_t ndat,
                                       uint64_t idat_start, uint32_t* hits_host, uint64_t* ndat_folded)
{
  return fold_set_bins_impl(f, phi, phase_per_sample, ndat, idat_start, nullptr, 0, 0, 0, hits_host, ndat_folded);
}

extern "C" int dspsr_amd_fold_set_bins_weighted(dspsr_amd_fold* f, double phi, double phase_per_sample, uint64_t ndat,
                                                uint64_t idat_start, const uint32_t* weights_host, uint64_t nweights,
                                                uint64_t ndatperweight, uint64_t weight_idat, uint32_t* hits_host,
                                                uint64_t* ndat_folded)
{
  if (weights_host && !ndatperweight) return DSPSR_AMD_EINVAL;
  return fold_set_bins_impl(f, phi, phase_per_sample, ndat, idat_start, weights_host, nweights, ndatperweight, weight_idat,
                            hits_host, ndat_folded);
}

extern "C" uint64_t dspsr_amd_fold_get_ndat_folded(const dspsr_amd_fold* f) { return f ? f->ndat_fold : 0; }
extern "C" float* dspsr_amd_fold_profiles_dev(dspsr_amd_fold* f) { return f ? f->profile : nullptr; }

extern "C" int dspsr_amd_fold_zero(dspsr_amd_fold* f)
{
  if (!f) return DSPSR_AMD_EINVAL;
  if (!f->profile) return DSPSR_AMD_OK;
  const size_t row = (size_t)f->nbin * f->ndim * sizeof(float);
  hipError_t e = f->span == (uint64_t)f->nbin * f->ndim
                     ? hipMemsetAsync(f->profile, 0, row * f->nchan * f->npol, f->ctx->stream)
                     : hipMemset2DAsync(f->profile, f->span * sizeof(float), 0, row, (size_t)f->nchan * f->npol, f->ctx->stream);
  if (e != hipSuccess) return ctx_fail(f->ctx, DSPSR_AMD_EHIP, "dspsr_amd_fold_zero: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

static int fold_fold_impl(dspsr_amd_fold* f, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                          uint32_t* hits_dev);

extern "C" int dspsr_amd_fold_fold(dspsr_amd_fold* f, const float* in_dev, uint64_t in_chan_stride,
                                   uint64_t in_pol_stride)
{
  return fold_fold_impl(f, in_dev, in_chan_stride, in_pol_stride, nullptr);
}

extern "C" int dspsr_amd_fold_fold_zeroed(dspsr_amd_fold* f, const float* in_dev, uint64_t in_chan_stride,
                                          uint64_t in_pol_stride, uint32_t* hits_dev)
{
  if (!hits_dev) return DSPSR_AMD_EINVAL;
  return fold_fold_impl(f, in_dev, in_chan_stride, in_pol_stride, hits_dev);
}

static int fold_fold_impl(dspsr_amd_fold* f, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                          uint32_t* hits_dev)
{
  if (!f || !in_dev) return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = f->ctx;
  if (!f->profile) return ctx_fail(ctx, DSPSR_AMD_ESTATE, "dspsr_amd_fold_fold: set_shape not called");
  if (f->folding_nbin != f->nbin)    // Fold.C:806-809
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dsp::Fold::fold folding_nbin != output->nbin (%u != %u)",
                    f->folding_nbin, f->nbin);
  if (f->binplan.empty()) return DSPSR_AMD_OK;             // send_binplan :160-161
  if (f->current_hits) f->binplan.back().hits = f->current_hits;   // :163-164
  f->current_hits = 0;

  // bucket the time-ordered intervals by phase bin (stable => time order kept inside a bin)
  const uint32_t nbin = f->nbin;
  const size_t niv = f->binplan.size();
  PlanSlot& sl = f->slot[f->next_slot];
  f->next_slot ^= 1;
  hipError_t e = hipSuccess;
  if (sl.pending) {            // the fold that last used this slot (two calls ago) must have consumed it
    e = hipEventSynchronize(sl.done);
    if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_fold_fold: %s", hipGetErrorString(e));
    sl.pending = false;
  }
  if (!slot_reserve(sl, nbin + 1, niv))
    return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_fold_fold: plan allocation failed");
  // sample span covered by the plan (intervals are time ordered)
  uint64_t first = f->binplan.front().offset, last = f->binplan.back().offset + f->binplan.back().hits;
  first -= first % 4;                                  // keeps 16-byte alignment of the chunk loads for any ndim
  const bool aligned = ((uintptr_t)in_dev % 16 == 0) && (in_chan_stride % 4 == 0) && (in_pol_stride % 4 == 0);
  // One walk over the runs decides which kernel folds them: the longest run (re-associated sums, see FOLD_LONG_RUN) and whether
  // the plan fits the dense per-chunk table of k_fold_dense -- at most one run per (chunk, phase bin), runs cut at the chunk ends;
  // a plan with two runs of a bin inside a chunk (a folding period shorter than the chunk) does not.  The table is also refused
  // when it would be more than a quarter of the bytes it helps to fold (few channels, many bins).  (This host code runs once per
  // block next to a kernel of a few hundred microseconds: at Benchmark/fold.csh's shape the three separate walks and the bucket
  // sort below, which the dense kernel does not read, made the host as slow as the device -- profiles/r05_experiments.txt item 6.)
  uint32_t max_run = 0;
  bool one_per_chunk = aligned && nbin <= (uint32_t)FOLD_BPT * 1024;
  size_t ntab = 0;
  if (one_per_chunk) {
    const uint64_t nchunk = (last - first + FOLD_CHUNK - 1) / FOLD_CHUNK;
    ntab = (size_t)nchunk * nbin;
    const uint64_t data_words = (last - first) * (uint64_t)f->nchan * f->npol * f->ndim;
    one_per_chunk = ntab <= ((size_t)1 << 24) && 4 * (uint64_t)ntab <= data_words;
  }
  if (one_per_chunk) {
    f->cursor.assign(nbin, ~0u);                            // (scratch: chunk of the bin's previous piece)
    uint32_t* const lastc = f->cursor.data();
    for (const RunBin& r : f->binplan) {
      if (r.hits > max_run) max_run = r.hits;
      if (r.hits == 0 || !one_per_chunk) continue;
      const uint64_t c0 = (r.offset - first) / FOLD_CHUNK, c1 = (r.offset - first + r.hits - 1) / FOLD_CHUNK;
      if (lastc[r.ibin] == (uint32_t)c0) one_per_chunk = false;             // a second run of this bin in the chunk
      lastc[r.ibin] = (uint32_t)c1;
    }
  } else {
    for (const RunBin& r : f->binplan) if (r.hits > max_run) max_run = r.hits;
  }
  const bool lng = aligned && nbin <= (uint32_t)FOLD_BPT * 1024 && max_run >= FOLD_LONG_RUN;   // re-associated sums (see FOLD_LONG_RUN)
  const bool will_dense = one_per_chunk && !lng;
  // the intervals bucketed by phase bin (stable => time order kept inside a bin): what the walk kernels and the per-channel hit
  // count of a zeroed input read -- not the dense kernel
  const bool need_iv = !will_dense || hits_dev;
  if (need_iv) {
    for (uint32_t b = 0; b <= nbin; b++) sl.h_bin_start[b] = 0;
    for (const RunBin& r : f->binplan) sl.h_bin_start[r.ibin + 1]++;
    for (uint32_t b = 0; b < nbin; b++) sl.h_bin_start[b + 1] += sl.h_bin_start[b];
    f->cursor.assign(sl.h_bin_start, sl.h_bin_start + nbin);
    for (const RunBin& r : f->binplan) {
      Interval v; v.offset = r.offset; v.hits = r.hits; v.pad = 0;
      sl.h_iv[f->cursor[r.ibin]++] = v;
    }
  }
  bool dense = will_dense;
  {
    if (dense) {
      if (ntab > sl.aux_cap) {
        if (sl.h_aux) (void)hipHostFree(sl.h_aux);
        if (sl.d_aux) (void)hipFree(sl.d_aux);
        sl.h_aux = nullptr; sl.d_aux = nullptr; sl.aux_cap = 0;
        const size_t n = ntab + ntab / 4 + 1024;
        if (hipHostMalloc((void**)&sl.h_aux, n * sizeof(uint32_t)) != hipSuccess || hipMalloc((void**)&sl.d_aux, n * sizeof(uint32_t)) != hipSuccess)
          return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_fold_fold: plan allocation failed");
        sl.aux_cap = n;
      }
      ::memset((void*)sl.h_aux, 0, ntab * sizeof(uint32_t));
      for (const RunBin& r : f->binplan) {
        uint64_t off = r.offset - first;
        uint32_t left = r.hits;
        while (left) {
          const uint64_t c = off / FOLD_CHUNK;
          const uint32_t s0 = (uint32_t)(off % FOLD_CHUNK), n = left < FOLD_CHUNK - s0 ? left : FOLD_CHUNK - s0;
          sl.h_aux[c * nbin + r.ibin] = s0 | (n << 11);
          off += n;
          left -= n;
        }
      }
    }
  }
  {
    const PlanCopy pc[3] = {{sl.d_bin_start, sl.h_bin_start, need_iv ? (nbin + 1) * sizeof(uint32_t) : 0},
                            {sl.d_iv, sl.h_iv, need_iv ? niv * sizeof(Interval) : 0},
                            {sl.d_aux, sl.h_aux, dense ? ntab * sizeof(uint32_t) : 0}};
    e = plan_upload(f, sl, pc, 3);
  }
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_fold_fold: plan copy: %s", hipGetErrorString(e));
  if (fold_plan_wait(f, &sl) != DSPSR_AMD_OK) return DSPSR_AMD_EHIP;
  const uint32_t nrow = f->npol * f->nchan;
  // exact mode: rows x bin groups, at least two workgroups per CU when the band has few channels
  uint32_t nsplit = 1;
  if (!lng)
    while (nsplit < 8 && (uint64_t)nrow * nsplit < 512 && nbin / (2 * nsplit) >= 64) nsplit *= 2;
  uint32_t threads = nbin < 1024 ? ((nbin + 63) / 64) * 64 : 1024;
  if (aligned && nbin <= (uint32_t)FOLD_BPT * 1024) {
    // FOLD_BPT bins per thread: 256-thread workgroups for nbin <= 1024, so that four of them share a CU and
    // keep 4 x 32 KiB of chunk loads in flight (the kernel is HBM-latency bound per workgroup)
    const uint32_t bins_wg = (nbin + nsplit - 1) / nsplit;
    threads = ((bins_wg + FOLD_BPT - 1) / FOLD_BPT + 63) / 64 * 64;
    if (threads < 256) threads = 256;
    if (threads > 1024) threads = 1024;
    // LONG: rows x time segments (about four workgroups per CU), every sample read once
    const uint32_t nchunk = (uint32_t)((last - first + FOLD_CHUNK - 1) / FOLD_CHUNK);
    uint32_t nseg = 1, cps = nchunk;
    if (lng) {
      nseg = (4 * ctx->ncu + nrow - 1) / nrow;
      if (nseg > nchunk) nseg = nchunk;
      if (nseg > 65535) nseg = 65535;
      if (nseg < 1) nseg = 1;
      cps = (nchunk + nseg - 1) / nseg;
      nseg = (nchunk + cps - 1) / cps;
      const size_t need = (size_t)nseg * nrow * nbin * f->ndim;
      if (need > f->part_floats) {
        (void)hipStreamSynchronize(ctx->stream);
        if (f->part) (void)hipFree(f->part);
        f->part = nullptr; f->part_floats = 0;
        if (hipMalloc((void**)&f->part, need * sizeof(float)) != hipSuccess)
          return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_fold_fold: hipMalloc of %zu partial-sum floats failed", need);
        f->part_floats = need;
      }
    }
    // planes of one channel folded together (ndim < 4): 4 or 2 rows per workgroup when the channels alone fill the chip
    // (one row per workgroup at Benchmark/fold.csh's shape, four times the workgroups: 311-313 against 321-331 Msamples/s, same box)
    const uint32_t nrw = (f->ndim * f->npol == 4 && f->ndim < 4 && (uint64_t)f->nchan * (lng ? nseg : nsplit) >= 2 * ctx->ncu)
                             ? f->npol : 1u;
    dim3 grid(f->npol / nrw, f->nchan, lng ? nseg : nsplit);
    const size_t lds = ((size_t)FOLD_CHUNK + (lng ? FOLD_CHUNK / FOLD_MB : 0)) * f->ndim * nrw * sizeof(float);
#define FOLD_DENSE(ND, NR) hipLaunchKernelGGL((k_fold_dense<ND, NR>), grid, dim3(threads), lds, ctx->stream, in_dev, in_chan_stride, \
                                             in_pol_stride, f->profile, f->span, nbin, sl.d_aux, first, last)
#define FOLD_LAUNCH(ND, LG, NR) hipLaunchKernelGGL((k_fold_chunked<ND, LG, NR>), grid, dim3(threads), lds, ctx->stream, in_dev, \
                                               in_chan_stride, in_pol_stride, f->profile, f->span, nbin, sl.d_bin_start, sl.d_iv, first, last, \
                                               f->part, cps)
    if (dense) {
      if (f->ndim == 4) FOLD_DENSE(4, 1);
      else if (f->ndim == 2 && nrw == 2) FOLD_DENSE(2, 2);
      else if (f->ndim == 2) FOLD_DENSE(2, 1);
      else if (nrw == 4) FOLD_DENSE(1, 4);
      else FOLD_DENSE(1, 1);
    } else
    if (f->ndim == 4) { if (lng) FOLD_LAUNCH(4, true, 1); else FOLD_LAUNCH(4, false, 1); }
    else if (f->ndim == 2 && nrw == 2) { if (lng) FOLD_LAUNCH(2, true, 2); else FOLD_LAUNCH(2, false, 2); }
    else if (f->ndim == 2) { if (lng) FOLD_LAUNCH(2, true, 1); else FOLD_LAUNCH(2, false, 1); }
    else if (nrw == 4) { if (lng) FOLD_LAUNCH(1, true, 4); else FOLD_LAUNCH(1, false, 4); }
    else { if (lng) FOLD_LAUNCH(1, true, 1); else FOLD_LAUNCH(1, false, 1); }
#undef FOLD_LAUNCH
#undef FOLD_DENSE
    if (lng) {
      const uint64_t n = (uint64_t)nrow * nbin * f->ndim;
      uint32_t gx = (uint32_t)((n + 255) / 256);
      if (gx > 4 * ctx->ncu) gx = 4 * ctx->ncu;
      hipLaunchKernelGGL(k_fold_combine, dim3(gx), dim3(256), 0, ctx->stream, f->profile, f->span, f->part, nrow, nbin * f->ndim, nseg);
    }
  } else {
    dim3 grid(f->npol, f->nchan, nsplit);
    if (f->ndim == 4)
      hipLaunchKernelGGL(k_fold_direct<4>, grid, dim3(threads), 0, ctx->stream, in_dev, in_chan_stride,
                         in_pol_stride, f->profile, f->span, nbin, sl.d_bin_start, sl.d_iv);
    else if (f->ndim == 2)
      hipLaunchKernelGGL(k_fold_direct<2>, grid, dim3(threads), 0, ctx->stream, in_dev, in_chan_stride,
                         in_pol_stride, f->profile, f->span, nbin, sl.d_bin_start, sl.d_iv);
    else
      hipLaunchKernelGGL(k_fold_direct<1>, grid, dim3(threads), 0, ctx->stream, in_dev, in_chan_stride,
                         in_pol_stride, f->profile, f->span, nbin, sl.d_bin_start, sl.d_iv);
  }
  if (hits_dev)      // per-channel hits of a zeroed input: polarisation 0, first float of every planned sample
    hipLaunchKernelGGL(k_fold_count_hits, dim3((nbin + 255) / 256, f->nchan), dim3(256), 0, ctx->stream, in_dev, in_chan_stride, f->ndim,
                       hits_dev, nbin, sl.d_bin_start, sl.d_iv);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(sl.done, ctx->stream);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_fold_fold: %s", hipGetErrorString(e));
  sl.pending = true;
  f->binplan.clear();
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_synch(dspsr_amd_fold* f, float* profile_host)   // FoldCUDA.cu:127-152
{
  if (!f || !profile_host) return DSPSR_AMD_EINVAL;
  if (!f->profile) return ctx_fail(f->ctx, DSPSR_AMD_ESTATE, "dspsr_amd_fold_synch: no profile");
  const size_t row = (size_t)f->nbin * f->ndim * sizeof(float);      // host copy is packed [chan][pol][nbin][ndim]
  hipError_t e = hipMemcpy2DAsync(profile_host, row, f->profile, f->span * sizeof(float), row, (size_t)f->nchan * f->npol,
                                  hipMemcpyDeviceToHost, f->ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(f->ctx->stream);
  if (e != hipSuccess) return ctx_fail(f->ctx, DSPSR_AMD_EHIP, "dspsr_amd_fold_synch: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

int fold_build_part_plan(dspsr_amd_fold* f, uint32_t nkeep, uint32_t npart, const uint32_t** d_start,
                         const Interval** d_iv, PlanSlot** slot)
{
  // Layout on the device (one uint32 array + the interval array):
  //   start[0 .. npart]                  : first active-bin entry of every part (start[npart] = total)
  //   start[align4(npart+1) + 4*e + 0..3] : entry e = { bin, first interval, count << 16 | hits0, offset0 }
  // Only the phase bins that receive samples in a part are listed, so a workgroup finds its work with two
  // dependent loads (entry, then interval + accumulator) instead of walking all nbin bins.
  dspsr_amd_ctx* ctx = f->ctx;
  if (f->current_hits && !f->binplan.empty()) f->binplan.back().hits = f->current_hits;   // FoldCUDA.cu:163-164
  f->current_hits = 0;
  const uint32_t nbin = f->nbin;
  // count the pieces: a run is cut at every multiple of nkeep
  size_t npiece = 0;
  for (const RunBin& r : f->binplan) {
    if (!r.hits) continue;
    const uint64_t p0 = r.offset / nkeep, p1 = (r.offset + r.hits - 1) / nkeep;
    if (p1 >= npart)
      return ctx_fail(ctx, DSPSR_AMD_EINVAL, "fused fold: plan sample %llu lies beyond the %u parts of this call",
                      (unsigned long long)(r.offset + r.hits - 1), npart);
    npiece += (size_t)(p1 - p0 + 1);
  }
  auto for_each_piece = [&](auto&& fn) {
    for (const RunBin& r : f->binplan) {
      uint64_t off = r.offset, left = r.hits;
      while (left) {
        const uint64_t part = off / nkeep, within = off % nkeep;
        const uint64_t n = left < nkeep - within ? left : nkeep - within;
        fn((uint32_t)part, r.ibin, within, (uint32_t)n);
        off += n; left -= n;
      }
    }
  };
  // bucket the pieces by (part, bin), time order kept inside a bucket
  const size_t nb1 = (size_t)npart * nbin;
  std::vector<uint32_t>& cnt = f->cursor;
  cnt.assign(nb1 + 1, 0u);
  for_each_piece([&](uint32_t part, uint32_t ibin, uint64_t, uint32_t) { cnt[(size_t)part * nbin + ibin + 1]++; });
  size_t nentry = 0;
  for (size_t i = 0; i < nb1; i++) { if (cnt[i + 1]) nentry++; cnt[i + 1] += cnt[i]; }     // cnt[i] = first interval of bucket i
  PlanSlot& sl = f->slot[f->next_slot];
  f->next_slot ^= 1;
  if (sl.pending) {
    hipError_t e = hipEventSynchronize(sl.done);
    if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "fused fold: %s", hipGetErrorString(e));
    sl.pending = false;
  }
  const size_t ent_off = ((size_t)npart + 1 + 3) & ~(size_t)3;         // entries are 16-byte aligned uint4
  const size_t nwords = ent_off + 4 * nentry;
  if (!slot_reserve(sl, nwords, npiece ? npiece : 1))
    return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "fused fold: plan allocation failed");
  std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
  for_each_piece([&](uint32_t part, uint32_t ibin, uint64_t within, uint32_t n) {
    Interval v; v.offset = within; v.hits = n; v.pad = 0;
    sl.h_iv[fill[(size_t)part * nbin + ibin]++] = v;
  });
  uint32_t* st = sl.h_bin_start;
  uint32_t* ent = st + ent_off;
  for (size_t i = npart + 1; i < ent_off; i++) st[i] = 0;
  size_t e = 0;
  for (uint32_t part = 0; part < npart; part++) {
    st[part] = (uint32_t)e;
    for (uint32_t b = 0; b < nbin; b++) {
      const size_t i = (size_t)part * nbin + b;
      const uint32_t n = cnt[i + 1] - cnt[i];
      if (!n) continue;
      const Interval& first = sl.h_iv[cnt[i]];
      // {bin, first interval index, count << 16 | hits of the first interval, offset of the first interval}:
      // count, hits and offsets are < nkeep <= 8192 on the three-pass path
      ent[4 * e] = b; ent[4 * e + 1] = cnt[i]; ent[4 * e + 2] = (n << 16) | first.hits; ent[4 * e + 3] = (uint32_t)first.offset;
      e++;
    }
  }
  st[npart] = (uint32_t)e;
  const PlanCopy pc[2] = {{sl.d_bin_start, sl.h_bin_start, nwords * sizeof(uint32_t)}, {sl.d_iv, sl.h_iv, npiece * sizeof(Interval)}};
  const hipError_t er = plan_upload(f, sl, pc, 2);      // the caller waits (fold_plan_wait) in front of the kernel that reads it
  if (er != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "fused fold: plan copy: %s", hipGetErrorString(er));
  f->binplan.clear();
  *d_start = sl.d_bin_start;
  *d_iv = sl.d_iv;
  *slot = &sl;
  return DSPSR_AMD_OK;
}

int fold_combine_partials(dspsr_amd_fold* f, const float* part, uint32_t nseg, uint32_t chan0, uint32_t nchan)
{
  // profile rows [chan0, chan0 + nchan) += partial profiles of the runs 1 .. nseg of a segmented fused launch (packed
  // [seg][chan - chan0][nbin][ndim], npol 1), in run order
  const uint32_t nrow = nchan * f->npol;
  const uint64_t n = (uint64_t)nrow * f->nbin * f->ndim;
  uint32_t gx = (uint32_t)((n + 255) / 256);
  if (gx > 4 * f->ctx->ncu) gx = 4 * f->ctx->ncu;
  hipLaunchKernelGGL(k_fold_combine, dim3(gx), dim3(256), 0, f->ctx->stream, f->profile + (uint64_t)chan0 * f->npol * f->span, f->span,
                     part, nrow, f->nbin * f->ndim, nseg);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(f->ctx, DSPSR_AMD_EHIP, "fused fold: combine: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

int fold_part_plan_submitted(dspsr_amd_fold* f, PlanSlot* slot)
{
  hipError_t e = hipEventRecord(slot->done, f->ctx->stream);
  if (e != hipSuccess) return ctx_fail(f->ctx, DSPSR_AMD_EHIP, "fused fold: %s", hipGetErrorString(e));
  slot->pending = true;
  return DSPSR_AMD_OK;
}

int fold_build_segment_plan(dspsr_amd_fold* f, uint64_t ndat, uint32_t seg, bool* ok, const uint32_t** d_run_off,
                            const uint32_t** d_blk_first, const uint32_t** d_bin_start, const Interval** d_iv, PlanSlot** slot)
{
  dspsr_amd_ctx* ctx = f->ctx;
  *ok = false;
  if (f->binplan.empty() || ndat == 0 || ndat >= (1ull << 32)) return DSPSR_AMD_OK;
  // (the open run's hits are final only once the plan is consumed: look at them without closing it)
  const size_t nrun = f->binplan.size();
  auto hits_of = [&](size_t i) { return i + 1 == nrun && f->current_hits ? f->current_hits : f->binplan[i].hits; };
  uint64_t expect = 0;
  for (size_t i = 0; i < nrun; i++) {
    const RunBin& r = f->binplan[i];
    const uint32_t h = hits_of(i);
    if (r.offset != expect || h == 0) return DSPSR_AMD_OK;                 // a gap (dropped samples) or an offset start
    if (i > 0 && i + 1 < nrun && h < seg) return DSPSR_AMD_OK;             // an inner interval shorter than a segment
    expect += h;
  }
  if (expect != ndat) return DSPSR_AMD_OK;
  if (f->current_hits) f->binplan.back().hits = f->current_hits;           // FoldCUDA.cu:163-164
  f->current_hits = 0;
  const uint32_t nbin = f->nbin;
  PlanSlot& sl = f->slot[f->next_slot];
  f->next_slot ^= 1;
  if (sl.pending) {
    const hipError_t e = hipEventSynchronize(sl.done);
    if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "fused fold: %s", hipGetErrorString(e));
    sl.pending = false;
  }
  const size_t nblk = (size_t)(ndat >> 10) + 1, naux = nrun + 1 + nblk;
  if (!slot_reserve(sl, nbin + 1, nrun)) return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "fused fold: plan allocation failed");
  if (naux > sl.aux_cap) {
    if (sl.h_aux) (void)hipHostFree(sl.h_aux);
    if (sl.d_aux) (void)hipFree(sl.d_aux);
    sl.h_aux = nullptr; sl.d_aux = nullptr; sl.aux_cap = 0;
    const size_t n = naux + naux / 2 + 16;
    if (hipHostMalloc((void**)&sl.h_aux, n * sizeof(uint32_t)) != hipSuccess || hipMalloc((void**)&sl.d_aux, n * sizeof(uint32_t)) != hipSuccess)
      return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "fused fold: plan allocation failed");
    sl.aux_cap = n;
  }
  uint32_t* off = sl.h_aux;
  uint32_t* blk = sl.h_aux + nrun + 1;
  for (size_t i = 0; i < nrun; i++) off[i] = (uint32_t)f->binplan[i].offset;
  off[nrun] = (uint32_t)ndat;
  size_t q = 0;
  for (size_t i = 0; i < nblk; i++) {
    const uint64_t s0 = (uint64_t)i << 10;
    while (q + 1 < nrun && off[q + 1] <= s0) q++;
    blk[i] = (uint32_t)q;
  }
  // the same intervals bucketed by phase bin, time order kept inside a bin (as dspsr_amd_fold_fold)
  for (uint32_t b = 0; b <= nbin; b++) sl.h_bin_start[b] = 0;
  for (const RunBin& r : f->binplan) sl.h_bin_start[r.ibin + 1]++;
  for (uint32_t b = 0; b < nbin; b++) sl.h_bin_start[b + 1] += sl.h_bin_start[b];
  f->cursor.assign(sl.h_bin_start, sl.h_bin_start + nbin);
  for (const RunBin& r : f->binplan) {
    Interval v; v.offset = r.offset; v.hits = r.hits; v.pad = 0;
    sl.h_iv[f->cursor[r.ibin]++] = v;
  }
  const PlanCopy pc[3] = {{sl.d_aux, sl.h_aux, naux * sizeof(uint32_t)}, {sl.d_bin_start, sl.h_bin_start, (nbin + 1) * sizeof(uint32_t)},
                          {sl.d_iv, sl.h_iv, nrun * sizeof(Interval)}};
  hipError_t e = plan_upload(f, sl, pc, 3);             // the caller waits (fold_plan_wait) in front of the kernel that reads it
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "fused fold: plan copy: %s", hipGetErrorString(e));
  f->binplan.clear();
  *d_run_off = sl.d_aux;
  *d_blk_first = sl.d_aux + nrun + 1;
  *d_bin_start = sl.d_bin_start;
  *d_iv = sl.d_iv;
  *slot = &sl;
  *ok = true;
  return DSPSR_AMD_OK;
}

int fold_segment_combine(dspsr_amd_fold* f, const float* msum, uint32_t chan0, uint32_t nchan, uint32_t npart, uint32_t nkeep,
                         uint32_t nfilt_pos, int logTt, int logMa, int logMb, const uint32_t* d_bin_start, const Interval* d_iv)
{
  // (profile of npol 1 x ndim 4, rows 16-byte aligned: checked by the caller)
  hipLaunchKernelGGL(k_fold_segsum, dim3((f->nbin + 255) / 256, nchan), dim3(256), 0, f->ctx->stream,
                     f->profile + (uint64_t)chan0 * f->span, f->span, f->nbin, (const float4*)msum, npart, nkeep, nfilt_pos, logTt, logMa,
                     logMb, d_bin_start, d_iv);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(f->ctx, DSPSR_AMD_EHIP, "fused fold: segment combine: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}
