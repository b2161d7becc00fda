// Convolving filterbank, forward pass 2 (rows of the spectrum); see fb_common.h
#include "fb_common.h"

namespace dspsr_amd {

// ------------------------------------------------------------------------------------ P2
// Rr-point forward FFTs along T2 adjacent rows ka of A (one contiguous block) -> spectrum rows
// s' = kb, bin m = ka, stored as X[s'/T3][m][s'%T3].
template <int LOGF, int LOGT>
__global__ __launch_bounds__(512) void k_fwd_rows(const FbGeom g, const cf* __restrict__ A, cf* __restrict__ X,
                                                  const cf* __restrict__ tw, const uint32_t nparts,
                                                  const uint32_t nseq, const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logT = LOGT >= 0 ? LOGT : g.logT2, logT3 = g.logX3;     // X layout block factor
  const uint32_t T2 = 1u << logT, T3 = 1u << logT3;
  const uint64_t L = 1ull << (g.logM + LOGF);
  const uint32_t ntile = 1u << (g.logM - logT);
  const uint32_t total = ntile * nseq * nparts;
  const int logNt = g.logM - logT;          // ntile = 2^logNt
  // The sequences of the launch are walked backwards: the parts written last -- the ones still in the Infinity Cache
  // when the launch ends -- are then the ones the inverse pass, which walks the parts forwards, meets first
  // (+0.8 % Msamples/s in three alternating runs)
  auto seq_of = [&](const uint32_t item) -> uint64_t { return nseq * nparts - 1 - (item >> logNt); };

  // the prefetch keeps the loaded 16-byte pairs untouched (any use would wait for the loads at once);
  // they are rearranged into split form when the tile is started
  auto fetch = [&](const uint32_t item, float4 (&y)[NPAIR]) {
    const uint32_t tile = item & (ntile - 1);
    const cf* __restrict__ Ablk = A + seq_of(item) * L + (((uint64_t)tile << LOGF) << logT);     // g.logR == LOGF
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++)
        y[(g2 / 2) * P::R1 + i] = ld_stream((const float4*)&Ablk[first_stage_elem<LOGF>(tid, logT, g2, i)]);
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;      // behind the exchange buffer (16-byte aligned)
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  // copy-out of the staged tile (end of the tile loop): thread part of the addresses, once per kernel
  const uint32_t co_swz = (PTS * blockDim.x) >= 256 ? 3u : 0u;
  const uint32_t co_l0 = 2 * threadIdx.x, co_n2 = LOGT >= 0 ? (2u << (LOGF + LOGT - LOG_PTS)) : 2 * blockDim.x;   // full tiles: a constant
  const int co_sh = logT + logT3;
  const bool co_fast = (co_n2 & 63) == 0 && (g.xblocked || ((co_n2 >> co_sh) != 0 && (co_n2 & ((1u << co_sh) - 1)) == 0));   // uniform
  const uint32_t co_lds = lds_pad(co_l0 ^ (((co_l0 >> 4) & co_swz) << 1)), co_lstep = co_n2 + ((co_n2 >> 6) << 2);
  const uint32_t co_goff = (uint32_t)((g.xblocked ? (uint64_t)co_l0
                                                   : ((((uint64_t)(co_l0 >> co_sh) << g.logM) << logT3) + (co_l0 & ((1u << co_sh) - 1)))) * sizeof(cf));
  const uint64_t co_gstep = g.xblocked ? (uint64_t)co_n2 : (((uint64_t)(co_n2 >> co_sh) << g.logM) << logT3);   // elements of X per pair step
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  float4 y[NPAIR];
  fetch(item, y);
  FB_ST_BEGIN(2);
  for (;;) {
    asm volatile("" : "+v"(tid));   // per-tile index math stays inside the loop (see wgfft)
    cx2 x[NPAIR];
    FB_ST(2, 0);                     // (waits for the prefetched tile first)
#pragma unroll
    for (int i = 0; i < NPAIR; i++) x[i] = make_cx2(make_float2(y[i].x, y[i].y), make_float2(y[i].z, y[i].w));
    // The twiddle W_L^{nb*ka} between the two forward passes is applied HERE, to the elements pass 2 has just loaded, not
    // to pass 1's outputs: pass 1 is bound by the vector instructions it issues (31 packed complex products and 10 sin/cos
    // per thread and tile for this twiddle alone), pass 2 by the fabric with its vector unit two thirds idle.  The product
    // nb*ka is symmetric: the column pair is (ka, ka + 1), the position nb = pos0 + i*S.
    {
      const uint32_t tile_t = item & (ntile - 1);
      constexpr uint32_t S = 1u << (LOGF - P::LOGR1);
#pragma unroll
      for (int g2 = 0; g2 < P::G1; g2 += 2) {
        const uint32_t eb = P::G1 * tid + g2;
        cx2 (&xg)[P::R1] = *reinterpret_cast<cx2 (*)[P::R1]>(&x[(g2 / 2) * P::R1]);
        apply_pass_twiddle<P::R1>(xg, tile_t * T2 + (eb & (T2 - 1)), eb >> logT, S, g.logM + LOGF, tw, g.tw_lo);
      }
    }
    FB_ST(2, 1);
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    if (more) fetch(next, y);
    FB_ST(2, 2);

    const uint32_t tile = item & (ntile - 1);
    cf* __restrict__ Xseq = X + seq_of(item) * g.xstride;
    // last-stage outputs go to LDS in X-layout order [s'/T3][klo][s'%T3]; after a barrier the tile is
    // written out as whole runs of T2*T3 elements with 16-byte-per-lane stores.  XOR swizzle of the image
    // (bits 1,2 ^= bits 4,5) against bank conflicts of the 8-byte scatter (42 % of this pass's LDS cycles)
    const uint32_t swz = (PTS * blockDim.x) >= 256 ? 3u : 0u;
    auto store = [&](const uint32_t klo, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      auto img = [&](const uint32_t l) { return lds_pad(l ^ (((l >> 4) & swz) << 1)); };
      const uint32_t l0 = ((((p >> logT3) << logT) + klo) << logT3) | (p & (T3 - 1));
      const uint32_t step = pstride << logT;            // image index step per k (pstride is a multiple of T3)
      const bool aff = (step & 63) == 0 && (pstride & (T3 - 1)) == 0;
      const uint32_t b0 = img(l0), b1 = img(l0 + T3), sp = step + (step >> 4);
      if (aff) {                                       // uniform
#pragma unroll
        for (int k = 0; k < R; k++) {
          float* __restrict__ d0 = (float*)&lds[b0 + k * sp];
          float* __restrict__ d1 = (float*)&lds[b1 + k * sp];
          d0[0] = v[k].x[0]; d0[1] = v[k].y[0];
          d1[0] = v[k].x[1]; d1[1] = v[k].y[1];
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t srow = k * pstride + p;
          const uint32_t l = ((((srow >> logT3) << logT) + klo) << logT3) | (srow & (T3 - 1));
          lds[img(l)] = cx2_lo(v[k]);
          lds[img(l + T3)] = cx2_hi(v[k]);
        }
      }
    };
    wgfft<LOGF, -1, true>(lds, ltw_off, tid, logT, x, store);
    __syncthreads();
    FB_ST(2, 3);
    {
      const uint32_t nthr = blockDim.x;
      if (co_fast) {
        // thread part of the addresses computed once per kernel, per-pair part an immediate / a uniform step (see pass 1)
        const char* __restrict__ gb = (const char*)(Xseq + (g.xblocked ? (uint64_t)tile * g.xblock : ((uint64_t)(tile * T2) << logT3)));
#pragma unroll
        for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
          float4 pr[4];
#pragma unroll
          for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[co_lds + (j4 + q) * co_lstep];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; q++) st_stream((float4*)(gb + (uint64_t)(j4 + q) * co_gstep * sizeof(cf) + co_goff), pr[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 4
        for (int jj = 0; jj < PTS / 2; jj++) {
          const uint32_t l = 2 * (tid + jj * nthr);
          const uint32_t blkX = l >> (logT + logT3), within = l & ((1u << (logT + logT3)) - 1);
          const float4 pr = *(const float4*)&lds[lds_pad(l ^ (((l >> 4) & swz) << 1))];
          // four-pass mode (xblocked): the tile's image [kb][ka % T2] IS its block of X -- one contiguous 2^14-element
          // store instead of runs of T2 elements scattered over the natural order
          const uint64_t xo = g.xblocked ? (uint64_t)tile * g.xblock + l
                                         : ((((uint64_t)blkX << g.logM) + tile * T2) << logT3) + within;
          st_stream((float4*)&Xseq[xo], pr);
        }
      }
    }
    FB_ST(2, 4);
    FB_ST_TILE(2, 5);
    if (!more) break;
    item = next;
  }
  FB_ST_END(2);
}


template <int... I> static k2_t pick2(int logf, bool full, iseq<I...>)
{
  static const k2_t t[] = {k_fwd_rows<I, -1>...};
  static const k2_t f[] = {k_fwd_rows<I, full_logt(I)>...};
  return full ? f[logf] : t[logf];
}
k2_t fb_pick2(int logf, bool full) { return pick2(logf, full, seq_t()); }

}  // namespace dspsr_amd

FB_ST_READER(fwd_rows)
