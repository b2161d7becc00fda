// Convolving filterbank, two-pass inverse of long responses (k_inv_a, k_inv_b); see fb_common.h
#include "fb_common.h"

namespace dspsr_amd {

// ------------------------------------------------------------------------------------ P3a / P3b
// Two-pass inverse transform for freq_res = Ma*Mb beyond one workgroup tile (and for nchan_subband = 1,
// i.e. dsp::Convolution): bin m = m1*Mb + m2, output sample t = t1 + Ma*t2.
//   P3a k_inv_a : spectrum X (bin k = c*freq_res + m, blocked by pass-2 tile: FbGeom::xblocked) -> Hermitian split / pol
//                 select -> x chirp
//                 -> inverse Ma-point FFTs over m1 for Tm adjacent m2 -> x conj(W_M^{m2*t1})
//                 -> U[c][t1/Tt][m2][t1%Tt][pol]
//   P3b k_inv_b : inverse Mb-point FFTs over m2 for Tt adjacent t1 (one contiguous block of U)
//                 -> keep window on t = t1 + Ma*t2 -> complex output or fused detection
// Columns of both tiles are (column, pol) pairs, so the thread's two butterflies are the two polarisations.
// REAL: real input (one packed sequence per part, the polarisations separated by the Hermitian split) or complex input
// (npol sequences), fixed at compile time: the split loop and the mirror addresses are then free of per-element branches
// (the run-time form cost one uniform branch per element and load, 5577 ISA lines per tile at -x 262144).
// FULL: the tile is the whole workgroup tile (Ma * 2*Tm = 2^14 elements, 512 threads): the column count is then a
// compile-time constant and the exchange addresses of the transform fold (wgfft_stage's uniform selects otherwise cost
// two branches per element: 107 per tile in the -x 262144 listing).
template <int LOGF, bool BLOCKED, bool REAL, bool FULL>
__global__ __launch_bounds__(512) void k_inv_a(const FbGeom g, const cf* __restrict__ X, const cf* __restrict__ kernel,
                                               cf* __restrict__ U, const cf* __restrict__ tw, const uint32_t nparts,
                                               const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logTm = FULL ? 13 - LOGF : g.logTm, logT = logTm + 1, logTt = g.logTt;
  const uint32_t Tm = 1u << logTm, Tt = 1u << logTt;
  const uint64_t L = 1ull << (g.logM + g.logR);
  const uint32_t nseq = REAL ? 1 : g.npol;
  const bool npol2 = g.npol == 2;
  const uint32_t ntile = 1u << (g.logMb - logTm);          // m2 tiles per channel
  const uint32_t per_part = ntile * g.C;
  const uint32_t total = per_part * nparts;
  const int logNt = g.logMb - logTm;        // ntile = 2^logNt
  struct Abk { cf a, b; };

  // Blocked spectrum (FbGeom::xblocked, written by k_fwd_rows): bin k = ka + Fa*kb lies at
  // (ka >> logT2)*xblock + (kb << logT2 | ka % T2).  The tile's Ma x Tm bins are enumerated in MEMORY order -- the low
  // bits of m2 (inside a run of T2), then the bits of m1 that fall into kb (consecutive in memory), then the rest -- so a
  // wave's load covers whole runs (1 KB at -F 64:D -x 262144) although a first-stage butterfly needs bins Ma/16 rows apart;
  // the split and chirp-multiplied elements change to butterfly order through the exchange buffer (one extra LDS round trip).
  // Loading in butterfly order instead touches 32-byte pieces 128 KB apart: +54 % on this pass (r02 experiments, item 18).
  const int nlow = g.logT2 < logTm ? g.logT2 : logTm;
  const int hs0 = g.logM - g.logMb, hs = hs0 < 0 ? 0 : (hs0 > LOGF ? LOGF : hs0);    // m1 bits below `hs` stay in ka
  const int nhh = LOGF - hs, nlh = logTm - nlow;
  const uint32_t maskA = (1u << g.logM) - 1, maskT = (1u << g.logT2) - 1;
  auto tile_elem = [&](const uint32_t e, uint32_t& m1, uint32_t& j) {
    const uint32_t w = e & ((1u << nlow) - 1), e1 = e >> nlow;
    const uint32_t mhh = e1 & ((1u << nhh) - 1), e2 = e1 >> nhh;
    j = ((e2 & ((1u << nlh) - 1)) << nlow) | w;
    m1 = (mhh << hs) | (e2 >> nlh);
  };
  auto xa = [&](const uint32_t k) -> uint32_t {            // k < L
    const uint32_t ka = k & maskA, kb = k >> g.logM;
    return (ka >> g.logT2) * g.xblock + ((kb << g.logT2) | (ka & maskT));
  };
  auto xk = [&](const uint32_t k) -> uint32_t {            // the chirp, permuted likewise by set_kernel (no padding)
    const uint32_t ka = k & maskA, kb = k >> g.logM;
    return (ka >> g.logT2) * g.kblock + ((kb << g.logT2) | (ka & maskT));
  };
  // The enumeration is a permutation of index BITS, and the element index of a thread's i-th element is tid + i*nthr
  // (nthr a power of two), so bin index, spectrum address, chirp address and staging address of that element all split
  // into a part that depends on the thread, a part that depends on the item (uniform) and one uniform increment per
  // bit of i: disjoint bit fields add.  16 elements then cost one vector add each instead of a full decode (the decode
  // per element made this pass issue 2650 vector instructions per thread and tile, 42 % of them integer).
  auto stg = [&](const uint32_t m1, const uint32_t j) { return lds_pad(((m1 << logTm) + j) << 1); };
  uint32_t thr_k, thr_st, Dk[4], Dxa[4], Dxk[4], Dst[4];
  {
    uint32_t m1, j;
    tile_elem(tid, m1, j);
    thr_k = (m1 << g.logMb) + j;
    thr_st = stg(m1, j);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      tile_elem((uint32_t)blockDim.x << b, m1, j);
      Dk[b] = (m1 << g.logMb) + j;
      Dxa[b] = xa(Dk[b]);
      Dxk[b] = xk(Dk[b]);
      Dst[b] = stg(m1, j);
    }
  }
  const uint32_t thr_xa = xa(thr_k), thr_xk = xk(thr_k);
  // Mirror bin L - k = ~k + 1 (real input).  ~k complements every bit field, so its address is xa(L-1) - xa(k); the + 1
  // adds 1 when the low T2 bits of k are not all zero, else carries into the row-group field (+ xblock - (T2-1)), else
  // into kb.  Which case applies is decided by the thread/item part of k unless the increment of element i reaches into
  // ka (uniform test): one select and one subtraction per element instead of a second full address computation.
  const uint32_t XAM = xa((uint32_t)L - 1), dCarryA = g.xblock - maskT,
                 dCarryB = (1u << g.logT2) - maskT - (maskA >> g.logT2) * g.xblock;
  // (the element increments never reach the low T2 bits of k: dspsr_amd_filterbank_create uses the blocked layout only
  //  when Mb >= T2 and the workgroup has at least T2 threads)
  auto inc = [&](const uint32_t (&D)[4], const int i) {
    return ((i & 1) ? D[0] : 0u) + ((i & 2) ? D[1] : 0u) + ((i & 4) ? D[2] : 0u) + ((i & 8) ? D[3] : 0u);
  };

  // The prefetch of the next tile is issued in NCH groups spread over the tile -- behind the split, inside the order
  // exchange, behind the butterflies of the first stages (wgfft's `mid` hook) -- instead of one burst of 32 loads per thread:
  // the burst blocked every wave in its load instructions for a quarter of the tile while the memory pipeline, still
  // draining the previous tile's stores, accepted them (stamps: 11.6k of 45.7k cycles, and as many again at the next barrier).
  // chunk < 0: all elements; otherwise the elements i with i % NCH == chunk.
  // (measured at cfg1opt / cfg1, same box: one burst 63.7k / 10.0k Msamples/s, two groups 67.2k / 9.9k, four 77.3k / 10.3-10.6k,
  //  five 75.1k, six 77.1k: four it is -- the natural order has no exchange to hide a group in and takes one in front of the
  //  copy-out stores instead)
  constexpr int CO_CHUNK = BLOCKED ? 0 : 1;
  constexpr int NMID = P::NS < 2 ? P::NS : 2;
  constexpr int NCH = 1 + (BLOCKED ? 1 : 0) + NMID + CO_CHUNK;
  auto fetch = [&](const uint32_t item, Abk (&raw)[PTS / 2], const int chunk) {
    const uint32_t r = item / nparts, part = item - r * nparts;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    const cf* __restrict__ X0s = X + (uint64_t)part * nseq * g.xstride;
    if constexpr (BLOCKED) {
      const uint32_t kt = (c << g.logMf) + tile * Tm;
      const uint32_t k0 = kt + thr_k, a0 = xa(kt) + thr_xa;
      if constexpr (REAL) {
        const uint32_t lowT = k0 & maskT, lowA = k0 & maskA;
        const uint32_t E2 = XAM + (lowT ? 1u : (lowA ? dCarryA : dCarryB));            // increment in kb only
        const uint32_t dE = XAM + (lowT ? 1u : dCarryA) - E2;                          // increment reaches into ka: E2 + dE
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) {
          if (chunk >= 0 && i % NCH != chunk) continue;
          const uint32_t ia = a0 + inc(Dxa, i);
          const uint32_t into_ka = (inc(Dk, i) & maskA) ? 1u : 0u;                     // uniform: a scalar, no branch
          uint32_t ib = E2 + into_ka * dE - ia;
          if (i == 0) ib = k0 == 0 ? 0u : ib;                                          // bin 0 is its own mirror
          Abk q;
          q.a = ld_stream(X0s + ia);
          q.b = ld_stream(X0s + ib);
          raw[i] = q;
        }
      } else {
        const cf* __restrict__ X1s = npol2 ? X0s + g.xstride : X0s;                    // (one polarisation: loaded twice, zeroed below)
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) {
          if (chunk >= 0 && i % NCH != chunk) continue;
          const uint32_t ia = a0 + inc(Dxa, i);
          Abk q;
          q.a = ld_stream(X0s + ia);
          q.b = ld_stream(X1s + ia);
          raw[i] = q;
        }
      }
      return;
    }
    // element i of the first-stage butterfly is bin k0 + i*step (m1 advances by MS): base plus a multiple of a
    // wave-uniform step; the mirror bin L - k runs down with the same step (k = 0, its own mirror, can only be i = 0)
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const int64_t step = (int64_t)MS << g.logMb;
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;
      const uint32_t j = (eb & ((1u << logT) - 1)) >> 1, m1b = eb >> logT;
      const uint64_t k0 = ((uint64_t)c << g.logMf) + ((uint64_t)m1b << g.logMb) + tile * Tm + j;
      const cf* __restrict__ pa = X0s + k0;
      const cf* __restrict__ pb = REAL ? X0s + (L - k0) : pa + (npol2 ? L : 0);
      const int64_t stepb = REAL ? -step : step;
      const cf* __restrict__ pb0 = (REAL && k0 == 0) ? X0s : pb;
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        if (chunk >= 0 && ((g2 / 2) * P::R1 + i) % NCH != chunk) continue;
        Abk q;
        q.a = ld_stream(pa + i * step);
        q.b = ld_stream(i == 0 ? pb0 : pb + i * stepb);
        raw[(g2 / 2) * P::R1 + i] = q;
      }
    }
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  // Items are (tile, part) pairs with the part running fastest, and every workgroup takes one contiguous range of them: it
  // walks the parts of a tile one after the other, so the tile's chirp is loaded once and stays in registers
  // (one chirp read per launch instead of one per part: -1/7 of this pass's traffic at 8 parts per launch).
  uint32_t item = (uint32_t)(((uint64_t)total * blockIdx.x) / gridDim.x);
  const uint32_t item_end = (uint32_t)(((uint64_t)total * (blockIdx.x + 1)) / gridDim.x);
  if (item >= item_end) return;
  uint32_t next;
  Abk raw[PTS / 2];
  fetch(item, raw, -1);
  // The chirp of a tile stays in registers while the workgroup walks the tile's parts (loaded when the tile changes: 16
  // loads per thread less on 7 of 8 items; cfg1opt +2.5 %).  32 registers: the full-tile kernels fit them with 0-7 spilled
  // registers, except the blocked ones whose stages are all radix 16 (13-18 spills: those re-read the chirp per part from the
  // L2, as the generic kernels do).
  constexpr bool KEEPK = FULL && !(BLOCKED && LOGF % 4 == 0 && LOGF > 0);
  cf kk[PTS / 2];
  uint32_t kk_r = ~0u;
  FB_ST_BEGIN(4);
  for (;;) {
    asm volatile("" : "+v"(tid));
    const uint32_t r = item / nparts, part = item - r * nparts;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    cx2 x[NPAIR];
    FB_ST(4, 0);                     // (waits for the prefetched tile first)
    {
      if (!KEEPK || r != kk_r) {
      kk_r = r;
      if (BLOCKED && kernel) {
        const uint32_t c0 = xk((c << g.logMf) + tile * Tm) + thr_xk;
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) kk[i] = kernel[c0 + inc(Dxk, i)];
      } else if (kernel) {
        constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
#pragma unroll
        for (int g2 = 0; g2 < P::G1; g2 += 2) {
          const uint32_t eb = P::G1 * tid + g2;
          const cf* __restrict__ pk = kernel + ((uint64_t)c << g.logMf) + ((uint64_t)(eb >> logT) << g.logMb) + tile * Tm +
                                      ((eb & ((1u << logT) - 1)) >> 1);
#pragma unroll
          for (int i = 0; i < P::R1; i++) kk[(g2 / 2) * P::R1 + i] = pk[((uint64_t)i * MS) << g.logMb];
        }
      } else {
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) kk[i] = make_float2(1.f, 0.f);
      }
      }
#pragma unroll
      for (int i = 0; i < PTS / 2; i++) {
        const Abk q = raw[i];
        cf x0, x1;
        if constexpr (REAL) {
          x0 = make_float2(0.5f * (q.a.x + q.b.x), 0.5f * (q.a.y - q.b.y));
          x1 = make_float2(0.5f * (q.a.y + q.b.y), 0.5f * (q.b.x - q.a.x));
        } else {
          x0 = q.a;
          x1 = npol2 ? q.b : make_float2(0.f, 0.f);
        }
        x[i] = cmuls(make_cx2(x0, x1), kk[i]);
      }
    }
    next = item + 1;
    const bool more = next < item_end;
    FB_ST(4, 1);
    // unconditional (the last item of the range is fetched again and dropped: 1/64 of the reads at 8 parts per launch).  Under
    // `if (more)` the loads went to fresh registers and the copies into `raw` at the end of the conditional block waited for
    // them (`s_waitcnt vmcnt(0)` straight behind the 32 loads in the ISA): the prefetch overlapped nothing.
    const uint32_t nitem = more ? next : item;
    fetch(nitem, raw, 0);
    FB_ST(4, 2);
    if constexpr (BLOCKED) {
      // memory order -> butterfly order: element (m1, j) of the tile (both polarisations, 16 bytes) at word pair m1*Tm + j
      __syncthreads();                         // the previous tile's copy-out has finished with the buffer
#pragma unroll
      for (int i = 0; i < PTS / 2; i++)
        *(float4*)&lds[thr_st + inc(Dst, i)] = make_float4(x[i].x[0], x[i].x[1], x[i].y[0], x[i].y[1]);
      if (NCH > 1) fetch(nitem, raw, 1);
      __syncthreads();
#pragma unroll
      for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
        for (int i = 0; i < P::R1; i++) {
          const float4 pr = *(const float4*)&lds[lds_pad(first_stage_elem<LOGF>(tid, logT, g2, i))];
          x[(g2 / 2) * P::R1 + i].x = (v2f){pr.x, pr.y};
          x[(g2 / 2) * P::R1 + i].y = (v2f){pr.z, pr.w};
        }
      __syncthreads();                         // before the first stage's exchange overwrites the buffer
    }

    FB_ST(4, 3);
    cf* __restrict__ Uc = U + ((uint64_t)part * g.C + c) * (2ull << g.logMf);
    // staged image order [t1/Tt][j][t1%Tt][pol]: whole runs of Tm*Tt*2 elements go out with 16-byte stores
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      const uint32_t j = col >> 1;
      apply_pass_twiddle_inv<R>(v, tile * Tm + j, p, pstride, g.logMf, tw, g.tw_lo_m);
#pragma unroll
      for (int k = 0; k < R; k++) {
        const uint32_t t1 = k * pstride + p;
        const uint32_t l = (((((t1 >> logTt) << logTm) + j) << logTt) | (t1 & (Tt - 1))) << 1;
        *(float4*)&lds[lds_pad(l)] = make_float4(v[k].x[0], v[k].y[0], v[k].x[1], v[k].y[1]);
      }
    };
    auto mid = [&](const int phase) {
      if (NCH > 1 && phase >= 1 && phase <= NMID) fetch(nitem, raw, (BLOCKED ? 1 : 0) + phase);
    };
    wgfft<LOGF, +1, true>(lds, ltw_off, tid, logT, x, store, mid);
    __syncthreads();
    if (CO_CHUNK && NCH > 1) fetch(nitem, raw, NCH - 1);
    FB_ST(4, 4);
    {
      const uint32_t nthr = blockDim.x;
      const int logRun = logTm + logTt + 1;
      const uint32_t n2 = 2 * nthr;
      if ((n2 & 63) == 0 && (n2 >> logRun) != 0 && (n2 & ((1u << logRun) - 1)) == 0) {      // uniform
        // pair jj = pair 0 + jj*2*nthr elements: constant step in the padded image, uniform step in U (see pass 1)
        const uint32_t l0 = 2 * tid, lstep = n2 + ((n2 >> 6) << 2), lb = lds_pad(l0);
        const uint32_t goff = (uint32_t)(((((uint64_t)(l0 >> logRun) << g.logMb) << (logTt + 1)) + (l0 & ((1u << logRun) - 1))) * sizeof(cf));
        const uint64_t gstep = (((uint64_t)(n2 >> logRun) << g.logMb) << (logTt + 1)) * sizeof(cf);
        const char* __restrict__ gb = (const char*)(Uc + ((uint64_t)(tile * Tm) << (logTt + 1)));
#pragma unroll
        for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
          float4 pr[4];
#pragma unroll
          for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[lb + (j4 + q) * lstep];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; q++)
            st_stream((float4*)(gb + (uint64_t)(j4 + q) * gstep + goff), pr[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 4
        for (int jj = 0; jj < PTS / 2; jj++) {
          const uint32_t l = 2 * (tid + jj * nthr);
          const uint32_t tb = l >> logRun, within = l & ((1u << logRun) - 1);
          const float4 pr = *(const float4*)&lds[lds_pad(l)];
          st_stream((float4*)&Uc[((((uint64_t)tb << g.logMb) + tile * Tm) << (logTt + 1)) + within], pr);
        }
      }
    }
    FB_ST(4, 5);
    FB_ST_TILE(4, 6);
    if (!more) break;
    item = next;
  }
  FB_ST_END(4);
}

// FOLDB (FbOut kind 4): the tile holds, for one channel, Mb runs of Tt consecutive output samples (run t2 = samples
// t1 + Ma*t2, t1 in the tile's block): exactly the micro-blocks the long-run fold (fold.hip, FOLD_LONG_RUN) sums first.
// The detected samples are staged in the exchange buffer ([t2][t1], XOR-swizzled so that both the stage's writes and the
// per-run reads are conflict free); thread t2 adds its run in time order, cut at the one phase-bin boundary it may hold
// (the host admits this path only for plans whose inner intervals are >= Tt samples), and writes the two piece sums --
// 1/16 of the detected bytes instead of all of them.  fold_segment_combine (fold.hip) then adds, per (channel, bin), the
// pieces of the bin's intervals in time order.  Deterministic; equal to the time-order sum to float rounding like the
// long-run fold itself (other micro-block boundaries, so not bit-equal to it).
// FULL: whole workgroup tile (Mb * 2*Tt = 2^14 elements, 512 threads), column count fixed at compile time (see k_inv_a).
template <int LOGF, bool FOLDB, bool FULL>
__global__ __launch_bounds__(512) void k_inv_b(const FbGeom g, const cf* __restrict__ U, const FbOut out,
                                               const cf* __restrict__ tw, const uint64_t part0, const uint32_t nparts,
                                               const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logTt = FULL ? 13 - LOGF : g.logTt, logT = logTt + 1;
  const uint32_t ntile = 1u << (g.logMa - logTt);          // t1 blocks per channel
  const uint32_t per_part = ntile * g.C;
  const uint32_t total = per_part * nparts;
  const int logNt = g.logMa - logTt;        // ntile = 2^logNt

  auto fetch = [&](const uint32_t item, float4 (&y)[NPAIR]) {
    const uint32_t part = item / per_part, r = item - part * per_part;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    const cf* __restrict__ blk = U + ((uint64_t)part * g.C + c) * (2ull << g.logMf) + (((uint64_t)tile << g.logMb) << logT);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++)
        y[(g2 / 2) * P::R1 + i] = ld_stream((const float4*)&blk[first_stage_elem<LOGF>(tid, logT, g2, i)]);
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  uint32_t item, next;
  uint32_t jn = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, jn, run, total, item)) return;
  float4 y[NPAIR];
  fetch(item, y);
  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; i++) x[i] = make_cx2(make_float2(y[i].x, y[i].y), make_float2(y[i].z, y[i].w));
    const uint32_t lpart = item / per_part, r = item - lpart * per_part;
    const uint64_t part = part0 + lpart;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++jn, run, total, next);
    if (more) fetch(next, y);

    // (uniform output kind / ndim decided once per butterfly, not per element: the compiler does not unswitch them out of
    //  the unrolled loop; the keep window is the only per-element test)
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if constexpr (FOLDB) {
        // detected sample (run t2 = k*pstride + p, position j = col/2 in the run) -> float4 slot t2*Tt + (j ^ t2 % Tt)
        const uint32_t j = col >> 1, Ttm = (1u << logTt) - 1;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t t2 = k * pstride + p;
          float q[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, q);
          *(float4*)&lds[2 * ((t2 << logTt) + (j ^ (t2 & Ttm)))] = make_float4(q[0], q[1], q[2], q[3]);
        }
        return;
      }
      if (out.kind == 0) return;
      const uint32_t chan = out.chan0 + c;
      const uint32_t t1 = (tile << logTt) + (col >> 1);
      float* __restrict__ row = out.base + chan * out.chan_stride;
      auto each = [&](auto&& emit) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t pos = ((k * pstride + p) << g.logMa) + t1;
          if (pos < g.nfilt_pos || pos >= g.nfilt_pos + g.nkeep) continue;
          emit(pos - g.nfilt_pos, cx2_lo(v[k]), cx2_hi(v[k]));
        }
      };
      if (out.kind == 1) {
        float2* __restrict__ o0 = (float2*)(row + part * out.part_step);
        if (g.npol == 2)
          each([&](const uint32_t t, const cf va, const cf vb) {
            st_stream(o0 + t, va);
            st_stream((float2*)((float*)(o0 + t) + out.pol_stride), vb);
          });
        else
          each([&](const uint32_t t, const cf va, const cf) { st_stream(o0 + t, va); });
        return;
      }
      const uint64_t idat0 = part * g.nkeep;
      if (out.ndim == 4) {
        float4* __restrict__ o = (float4*)row + idat0;
        each([&](const uint32_t t, const cf va, const cf vb) {
          float q[4];
          detect4(va, vb, out.state, q);
          st_stream(o + t, make_float4(q[0], q[1], q[2], q[3]));
        });
      } else if (out.ndim == 2) {
        float2* __restrict__ o = (float2*)row + idat0;
        float2* __restrict__ o1 = (float2*)(row + out.pol_stride) + idat0;
        each([&](const uint32_t t, const cf va, const cf vb) {
          float q[4];
          detect4(va, vb, out.state, q);
          st_stream(o + t, make_float2(q[0], q[1]));
          st_stream(o1 + t, make_float2(q[2], q[3]));
        });
      } else {
        float* __restrict__ o = row + idat0;
        each([&](const uint32_t t, const cf va, const cf vb) {
          float q[4];
          detect4(va, vb, out.state, q);
          o[t] = q[0];
          o[out.pol_stride + t] = q[1];
          o[2 * out.pol_stride + t] = q[2];
          o[3 * out.pol_stride + t] = q[3];
        });
      }
    };
    wgfft<LOGF, +1, FOLDB>(lds, ltw_off, tid, logT, x, store);
    if constexpr (FOLDB) {
      __syncthreads();                                       // the tile's detected samples are staged
      const uint32_t Tt = 1u << logTt, Ttm = Tt - 1;
      for (uint32_t t2 = tid; t2 < (1u << LOGF); t2 += blockDim.x) {
        // run t2: output positions pos0 .. pos0 + Tt - 1 of the backward transform; kept: [nfilt_pos, nfilt_pos + nkeep)
        const uint32_t pos0 = (t2 << g.logMa) + (tile << logTt);
        const uint32_t lo = g.nfilt_pos, hi = g.nfilt_pos + g.nkeep;
        const uint32_t jlo = pos0 >= lo ? 0u : (lo - pos0 < Tt ? lo - pos0 : Tt), jhi = pos0 + Tt <= hi ? Tt : (hi > pos0 ? hi - pos0 : 0u);
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
        if (jlo < jhi) {
          const uint32_t i0 = (uint32_t)part * g.nkeep + (pos0 + jlo - g.nfilt_pos), i1 = i0 + (jhi - jlo);   // sample span in the block
          uint32_t qi = out.blk_first[i0 >> 10];
          while (out.pstart[qi + 1] <= i0) qi++;             // interval that holds sample i0 (inner intervals are >= Tt samples)
          const uint32_t cut = out.pstart[qi + 1] < i1 ? out.pstart[qi + 1] : i1;
          const uint32_t jc = jlo + (cut - i0);
          const float4* __restrict__ src = (const float4*)&lds[2 * (t2 << logTt)];
          for (uint32_t j = jlo; j < jc; j++) { const float4 q = src[j ^ (t2 & Ttm)]; sa.x += q.x; sa.y += q.y; sa.z += q.z; sa.w += q.w; }
          for (uint32_t j = jc; j < jhi; j++) { const float4 q = src[j ^ (t2 & Ttm)]; sb.x += q.x; sb.y += q.y; sb.z += q.z; sb.w += q.w; }
        }
        float4* __restrict__ o = (float4*)out.base + ((((uint64_t)c * out.nparts_plan + part) * ntile + tile) << (LOGF + 1)) + 2 * t2;
        o[0] = sa;
        o[1] = sb;
      }
      // (the next tile's first exchange write sits behind a barrier: wgfft)
    }
    if (!more) break;
    item = next;
  }
}


template <int... I> static k3a_t pick3a(int logf, bool blocked, bool real, bool full, iseq<I...>)
{
  static const k3a_t tn[] = {k_inv_a<I, false, false, false>...};
  static const k3a_t tb[] = {k_inv_a<I, true, false, false>...};
  static const k3a_t rn[] = {k_inv_a<I, false, true, false>...};
  static const k3a_t rb[] = {k_inv_a<I, true, true, false>...};
  static const k3a_t tnf[] = {k_inv_a<I, false, false, true>...};
  static const k3a_t tbf[] = {k_inv_a<I, true, false, true>...};
  static const k3a_t rnf[] = {k_inv_a<I, false, true, true>...};
  static const k3a_t rbf[] = {k_inv_a<I, true, true, true>...};
  if (full) return real ? (blocked ? rbf[logf] : rnf[logf]) : (blocked ? tbf[logf] : tnf[logf]);
  return real ? (blocked ? rb[logf] : rn[logf]) : (blocked ? tb[logf] : tn[logf]);
}
template <int... I> static k3b_t pick3b(int logf, bool foldb, bool full, iseq<I...>)
{
  static const k3b_t t[] = {k_inv_b<I, false, false>...};
  static const k3b_t f[] = {k_inv_b<I, true, false>...};
  static const k3b_t tf[] = {k_inv_b<I, false, true>...};
  static const k3b_t ff[] = {k_inv_b<I, true, true>...};
  // (FOLDB with a radix-2 / radix-4 remainder stage -- LOGF % 4 == 1, 2 -- spills 12-20 registers in the full-tile form and
  //  none in the generic one: those lengths keep the generic kernel)
  if (full && !(foldb && (logf % 4 == 1 || logf % 4 == 2))) return foldb ? ff[logf] : tf[logf];
  return foldb ? f[logf] : t[logf];
}
k3a_t fb_pick3a(int logf, bool blocked, bool real, bool full) { return pick3a(logf, blocked, real, full, seq_t()); }
k3b_t fb_pick3b(int logf, bool foldb, bool full) { return pick3b(logf, foldb, full, seq_t()); }

}  // namespace dspsr_amd

FB_ST_READER(four_pass)
