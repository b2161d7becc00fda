// Convolving filterbank, two-pass path of short responses (k_raw_cols, k_fwd_col1q, k_rows_inv); see fb_common.h
#include "fb_common.h"

namespace dspsr_amd {

// ------------------------------------------------------------------------------------ two-pass path (short responses)
// A part needs log2 L forward and log2 M inverse radix-2 levels; a workgroup tile holds 14.  When log2 L + log2 M <= 27
// (complex dual-pol input; the 50 MHz sub-band geometry -F 512:D -x 512 is 18 + 9) TWO tiles cover them, and the spectrum
// never makes its round trip through HBM:  L = Fa * Fb with Fb = 2^13 / M and Fa = L / Fb <= 2^14 (at 2^14 one whole column per
// tile), sample n = nb + Fb*na, bin k = ka + Fa*kb -- and with ka = a*M + j that bin is bin j of channel c = a + (Fa/M)*kb.
// Fa < 2^14 (fewer channels): pass 1 is the ordinary k_raw_transpose + k_fwd_cols on a geometry of its own (M = Fa, Rr = Fb,
// T2 = freq_res: the A layout below is exactly theirs); only Fa = 2^14 needs P0' / P1'.
//   P0' k_raw_cols   the 8-bit block regrouped per column and polarisation: Rt[part][pol][nb][na]
//   P1' k_fwd_col1   ONE Fa = 2^14-point FFT per tile (column nb of one polarisation): the even and the odd samples are the
//                    two interleaved columns of a 2^13-point wgfft, combined by one radix-2 step in registers;
//                    out A[pol][a][nb][j] (each (a, nb) run M contiguous elements)
//   P2' k_rows_inv   tile = (a, part), both polarisations: x W_L^{nb*ka}, Fb-point FFTs over nb in registers (-> kb, i.e. Fb
//                    whole channels), x chirp, through the exchange buffer, inverse M-point FFTs over j, keep window,
//                    detection, fold -- k_fwd_rows and k_inv_chan in one tile.
// Traffic per part at the sub-band geometry: 1 + 1 (regroup) + 1 + 4 (pass 1) + 4 + chirp (pass 2) MB instead of
// 1 + 1 + 1 + 4 + 4 + 4 + 4 + chirp, and one kernel's load / store phases less.

// P0': sample t = nb + Fb*na of a part, 4 bytes (p0 re, p0 im, p1 re, p1 im) -> Rt[part][pol][nb][na] byte pairs (re, im).
// A block regroups NS consecutive samples (NS/Fb rows na of all Fb columns) through LDS: 16-byte loads, 16-byte stores in
// runs of 2*NS/Fb bytes.
__global__ __launch_bounds__(256) void k_raw_cols(const FbGeom g, const FbIn in, uint16_t* __restrict__ Rt, const uint64_t part0)
{
  constexpr uint32_t NS = 8192, MAXFB = 64;
  __shared__ __attribute__((aligned(16))) uint16_t sm[2 * (NS + 8 * MAXFB)];
  const uint32_t tid = threadIdx.x;
  const int logFb = g.logFb2;
  const uint32_t Fb = 1u << logFb, R = NS >> logFb, RP = R + 8;      // rows of the block, row pitch (16-byte aligned, bank skew)
  const uint64_t part = blockIdx.y;
  const uint32_t na0 = blockIdx.x * R;
  const uint64_t t0 = (part0 + part) * in.part_step + ((uint64_t)na0 << logFb);
  if (in.nchan == 1) {
    const uint4* __restrict__ src = (const uint4*)((const uint8_t*)in.base + 4 * t0);
#pragma unroll 4
    for (uint32_t q = 0; q < NS / 4 / 256; q++) {
      const uint32_t v = tid + 256 * q;
      const uint4 w = src[v];
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t sidx = 4 * v + k, nb = sidx & (Fb - 1), r = sidx >> logFb;
        sm[nb * RP + r] = (uint16_t)(ww[k] & 0xffffu);
        sm[(Fb + nb) * RP + r] = (uint16_t)(ww[k] >> 16);
      }
    }
  } else {
    // several input channels in the block (byte ((t*nchan + c)*npol + p)*2 + d): this channel's word of every sample
    const uint32_t* __restrict__ src = (const uint32_t*)in.base;
    for (uint32_t sidx = tid; sidx < NS; sidx += 256) {
      const uint32_t w = src[(t0 + sidx) * in.nchan + in.ichan], nb = sidx & (Fb - 1), r = sidx >> logFb;
      sm[nb * RP + r] = (uint16_t)(w & 0xffffu);
      sm[(Fb + nb) * RP + r] = (uint16_t)(w >> 16);
    }
  }
  __syncthreads();
  const uint32_t r8n = R >> 3;                                         // 16-byte units per (pol, nb) row of the block
  for (uint32_t u = tid; u < 2 * Fb * r8n; u += 256) {
    const uint32_t r8 = u % r8n, row = u / r8n, seq = row >> logFb, nb = row & (Fb - 1);
    const uint4 val = *(const uint4*)&sm[row * RP + 8 * r8];
    uint16_t* __restrict__ dst = Rt + (((part * 2 + seq) << logFb) + nb) * 16384ull + na0 + 8 * r8;
    *(uint4*)dst = val;
  }
}

// P1'.  2^14 = 16^3 * 4: the column as FOUR interleaved sub-sequences
// y_c[m] = y[4m + c] -- the four columns of a 2^12-point wgfft (three radix-16 stages, none of them a remainder stage) -- and
// the last radix-4 level in registers:  Y[P + 2^12 q] = sum_c (-i)^(c q) W^(c P) F_c[P],  W = exp(-2 pi i / 2^14).
// The last stage leaves a thread the pair (F_c, F_c+1)[P_k], P_k = p + 256 k, with c = 0 in even lanes and c = 2 in odd lanes of
// the same p: each lane twiddles its own pair (apply_pass_twiddle: W^(c P) for the columns c, c + 1), the two lanes swap pairs
// (DPP) and each computes two of the four outputs -- even lanes rows P and P + 2^13, odd lanes P + 2^12 and P + 3*2^12.
// Against the first form (round 4: even / odd halves as a 2^13-point wgfft with an exchanged radix-2 stage + one radix-2 step in
// registers) one whole exchanged stage -- 16 b128 reads and 16 b128 writes per thread, two barriers -- is replaced by about 350
// vector instructions: -12 % (profiles/r04_experiments.txt item 9).
struct Col1qOut {
  cf* img;
  const cf* tw;
  const cf* tw_lo;
  int h;
  template <int R> DEV void operator()(const uint32_t col, const uint32_t p, const uint32_t pstride, cx2 (&v)[R])
  {
    static_assert(R == 16, "k_fwd_col1q: radix-16 last stage");
    apply_pass_twiddle<R>(v, col, p, pstride, 14, tw, tw_lo);            // v[k] = (G_c, G_c+1)[p + k*pstride], G_c = W^(c P) F_c
    const bool odd = col != 0;                                            // col = 0 (c = 0, 1) or 2 (c = 2, 3)
    const float sg = odd ? -1.0f : 1.0f;
    auto swp = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, false)); };
    // staged image: natural order of ka, 8-byte elements, rows of odd q moved by 8 elements (XOR of bit 3) so that the even
    // and the odd lanes of a store fall on different banks (their rows differ by a multiple of 2^12 elements)
    const uint32_t q1 = odd ? 1u : 0u;
    const uint32_t b1 = lds_pad((p ^ (q1 << 3)) + (q1 << 12)), b2 = lds_pad((p ^ (q1 << 3)) + ((q1 + 2) << 12));
#pragma unroll
    for (int k = 0; k < R; k++) {
      const cx2 own = v[k];
      cx2 rc;
      rc.x = (v2f){swp(own.x[0]), swp(own.x[1])};
      rc.y = (v2f){swp(own.y[0]), swp(own.y[1])};
      // even: (G0 + G2, G1 + G3) = (A, C) ; odd: (G0 - G2, G1 - G3) = (B, D)
      cx2 sm;
      sm.x = own.x * sg + rc.x;
      sm.y = own.y * sg + rc.y;
      const cf sa = cx2_lo(sm), sb = cx2_hi(sm);
      const cf r = odd ? make_float2(sb.y, -sb.x) : sb;                   // odd: -i D
      const uint32_t c = k * pstride;                                     // multiple of 64: the padding carries over
      const uint32_t o = c + ((c >> 6) << 2);
      img[b1 + o] = make_float2(sa.x + r.x, sa.y + r.y);                  // even: Y[P] = A + C          odd: Y[P + 2^12] = B - i D
      img[b2 + o] = make_float2(sa.x - r.x, sa.y - r.y);                  // even: Y[P + 2^13] = A - C   odd: Y[P + 3*2^12] = B + i D
    }
  }
};

template <int RAWW>
__global__ __launch_bounds__(512) void k_fwd_col1q(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                   const cf* __restrict__ tw, const uint32_t nparts,
                                                   const uint32_t nseq, const uint32_t run)
{
  constexpr int LOGF = 12, LOGT = 2;
  typedef FftPlan<LOGF> P;
  static_assert(P::REM == 0 && P::R1 == 16 && P::G1 == 2, "k_fwd_col1q: 16 x 16 x 16");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logFb = g.logFb2, logMi = g.logMf;
  const uint32_t Mi = 1u << logMi;
  const uint32_t total = (nseq * nparts) << logFb;
  auto fetch = [&](const uint32_t item, RawW<RAWW> (&raw)[PTS / 2]) {
    const uint64_t t0 = (uint64_t)item << 14;
#pragma unroll
    for (int i = 0; i < P::R1; i++) raw[i] = fetch_pair<RAWW>(g, in, 0, t0 + 2 * tid + 1024u * i);   // samples 4*pos + c, c = 2*(tid & 1) + {0, 1}
  };
  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  const uint32_t co_lds0 = lds_pad(2 * threadIdx.x), co_lds1 = lds_pad((2 * threadIdx.x) ^ 8u);
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  RawW<RAWW> raw[PTS / 2];
  fetch(item, raw);
  FB_ST_BEGIN(6);
  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
    FB_ST(6, 0);
#pragma unroll
    for (int h = 0; h < NPAIR; h++) {
      cf a, b;
      decode_pair<RAWW>(g, in, raw[h], a, b, 0);
      x[h] = make_cx2(a, b);
    }
    FB_ST(6, 1);
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    if (more) fetch(next, raw);
    FB_ST(6, 2);
    Col1qOut out;
    out.img = lds;
    out.tw = tw;
    out.tw_lo = g.tw_lo;
    out.h = 0;
    wgfft<LOGF, -1, true>(lds, ltw_off, tid, LOGT, x, out);
    __syncthreads();
    FB_ST(6, 3);
    const uint32_t nb = item & ((1u << logFb) - 1);
    cf* __restrict__ Aseq = A + ((uint64_t)(item >> logFb) << (14 + logFb));
#pragma unroll
    for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
      float4 pr[4];
#pragma unroll
      for (int q = 0; q < 4; q++)                                        // row block jj holds q = jj / 4: rows of odd q are XOR-8 swizzled
        pr[q] = *(const float4*)&lds[((((j4 + q) >> 2) & 1) ? co_lds1 : co_lds0) + (j4 + q) * (1024u + 64u)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t ka = 2 * tid + 1024u * (j4 + q);
        st_stream((float4*)&Aseq[((((ka >> logMi) << logFb) + nb) << logMi) + (ka & (Mi - 1))], pr[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    FB_ST(6, 4);
    FB_ST_TILE(6, 5);
    if (!more) break;
    item = next;
  }
  FB_ST_END(6);
}

// P2': see the head of this section.  LOGM + LOGFB == 13: a tile is Fb channels x 2 polarisations x M bins = 2^14 points;
// a thread holds, for NJ = 16 / Fb bins j = tid + 512*jq, the Fb rows nb of both polarisations (pair = (pol 0, pol 1)).
// Items, the fused fold (exact time order per tile, or segmented over part runs) and the output forms are k_inv_chan's.
template <int LOGM, int LOGFB, int EPI>
__global__ __launch_bounds__(512) void k_rows_inv(const FbGeom g, const cf* __restrict__ A,
                                                  const cf* __restrict__ kernel, const FbOut out,
                                                  const cf* __restrict__ tw, const uint64_t part0,
                                                  const uint32_t nparts, const uint32_t run)
{
  static_assert(LOGM + LOGFB == 13 && LOGFB >= 1 && LOGFB <= 4, "k_rows_inv: Fb channels x 2 pols x M bins = 2^14 points");
  constexpr bool FOLD = EPI == 1, SEARCH = EPI == 2;         // (the epilogues of k_inv_chan)
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  // (the inverse 512-point transforms as even / odd 256-point halves + a radix-2 step in registers measured -3 % fused but +28 %
  //  on the form that writes its output -- 32-byte store runs --: profiles/r04_experiments.txt item 10; not kept)
  constexpr int logT3 = LOGFB, logT = LOGFB + 1;                     // columns: (kb, pol)
  constexpr uint32_t Fb = 1u << LOGFB, T3 = Fb, NJ = 16 / Fb;
  const int logCa = g.logFa2 - LOGM;                                // Fa / M: channel stride between the rows kb of a tile
  const int logL = g.logFa2 + LOGFB;
  const uint64_t L = 1ull << logL;
  const uint32_t ntile = 1u << logCa;
  struct Abk { cf a, b; };
  struct Item { uint32_t tile, lp; };
  auto chan_of = [&](const uint32_t tile, const uint32_t kb) { return tile + (kb << logCa); };
  // Loads: the lane pair (2q, 2q + 1) needs bins j = 2q, 2q + 1 of both polarisations.  The even lane loads the two bins of
  // polarisation 0, the odd lane those of polarisation 1 -- one aligned 16-byte load each instead of two 8-byte ones (half the
  // load instructions: 8-byte-per-lane streams run at 5.6 TB/s, 16-byte ones at 7.1, tools/load_width_probe.hip; the wave
  // time of the prefetch is the ISSUE of its loads) -- and the halves are swapped between the two lanes when the tile is consumed.
  auto fetch = [&](const Item item, Abk (&raw)[PTS / 2]) {
    const cf* __restrict__ A0 = A + (uint64_t)item.lp * 2 * L + ((tid & 1u) ? L : 0) + (((uint64_t)item.tile << LOGFB) << LOGM) + (tid & ~1u);
#pragma unroll
    for (uint32_t jq = 0; jq < NJ; jq++)
#pragma unroll
      for (uint32_t nb = 0; nb < Fb; nb++) {
        const float4 v = ld_stream((const float4*)(A0 + (nb << LOGM) + 512u * jq));
        Abk r;
        r.a = make_float2(v.x, v.y);             // even lane: pol 0 of bin j     | odd lane: pol 1 of bin j - 1
        r.b = make_float2(v.z, v.w);             //            pol 0 of bin j + 1 |           pol 1 of bin j
        raw[jq * Fb + nb] = r;
      }
  };
  auto swap1 = [](const cf v) {                  // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
    return make_float2(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0xB1, 0xf, 0xf, false)),
                       __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0xB1, 0xf, 0xf, false)));
  };
  auto load_chirp = [&](const Item item, cf (&kk)[PTS / 2]) {
    if (kernel) {
#pragma unroll
      for (uint32_t jq = 0; jq < NJ; jq++)
#pragma unroll
        for (uint32_t kb = 0; kb < Fb; kb++) kk[jq * Fb + kb] = kernel[((uint64_t)chan_of(item.tile, kb) << LOGM) + tid + 512u * jq];
    } else {
#pragma unroll
      for (int i = 0; i < PTS / 2; i++) kk[i] = make_float2(1.f, 0.f);
    }
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGM>(lds, ltw_off, tw, tid, blockDim.x);
  const uint32_t plan_off = (ltw_off + ltw_entries_dev<LOGM>() + 1) & ~1u;     // (the host sizes the tables for LOGM)
  uint32_t* psl = nullptr;
  const uint4* __restrict__ fent_all = nullptr;
  bool use_psl = false;
  if constexpr (FOLD) {
    psl = (uint32_t*)&lds[plan_off + 4 * out.plan_cap];
    fent_all = (const uint4*)(out.pstart + ((out.nparts_plan + 1 + 3) & ~3u));
  }
  uint32_t jt = 0;
  Item item, next;
  uint32_t j = 0;
  uint32_t fold_b = blockIdx.x;
  const uint32_t fnseg = FOLD && out.nseg > 1 ? out.nseg : 1u;
  const uint32_t fntg = gridDim.x / fnseg;
  const uint32_t fseg = fnseg > 1 ? fold_b / fntg : 0u;
  const uint32_t fpps = (nparts + fnseg - 1) / fnseg;
  const uint32_t fp0 = fseg * fpps;
  const uint32_t fnp = fp0 >= nparts ? 0u : (nparts - fp0 < fpps ? nparts - fp0 : fpps);
  if (fnseg > 1) fold_b -= fseg * fntg;
  if constexpr (FOLD) {
    if (fnp == 0) return;
    use_psl = out.plan_cap > 0 && fnp + 1 <= FB_PSL_MAX;          // offsets of this workgroup's run of parts: psl[lp - fp0]
    if (use_psl) {
      for (uint32_t q = tid; q <= fnp; q += blockDim.x) psl[q] = out.pstart[part0 + fp0 + q];
      __syncthreads();
    }
  }
  const bool tile_major = EPI != 0 || (ntile >= gridDim.x && ntile % gridDim.x == 0);
  auto next_item = [&](const uint32_t jj, Item& it) -> bool {
    if (EPI != 0 || tile_major) {
      const uint32_t q = jj / fnp;
      it.tile = fold_b + q * fntg;
      it.lp = fp0 + (jj - q * fnp);
      return it.tile < ntile;
    } else {
      const uint32_t grid = gridDim.x, b = blockIdx.x;
      uint32_t hi, lo;
      if (grid & 7) {
        const uint32_t lin = b + jj * grid;
        hi = lin / run; lo = lin - hi * run;
      } else {
        const uint32_t q = jj * (grid >> 3) + (b >> 3);
        const uint32_t qr = q / run;
        hi = qr * 8 + (b & 7); lo = q - qr * run;
      }
      if (run == nparts) { it.tile = hi; it.lp = lo; }
      else { const uint32_t lin = hi * run + lo; it.tile = lin / nparts; it.lp = lin - it.tile * nparts; }
      return it.tile < ntile;
    }
  };
  if (!next_item(j, item)) return;
  Abk raw[PTS / 2];
  fetch(item, raw);
  cf kk[PTS / 2];
  uint32_t kk_tile = ~0u;
  uint32_t fe0_cur = 0, fn_cur = 0;
  auto plan_fetch = [&](const Item it) {
    if constexpr (FOLD) {
      const uint32_t lp = it.lp;
      if (use_psl) { fe0_cur = psl[lp - fp0]; fn_cur = psl[lp - fp0 + 1] - fe0_cur; }
      else { fe0_cur = out.pstart[part0 + lp]; fn_cur = out.pstart[part0 + lp + 1] - fe0_cur; }
    }
  };
  const bool plan_dma_ok = FOLD && FftPlan<LOGM>::NS >= 2 && use_psl;
  auto plan_dma = [&](const Item it, const uint32_t buf) {
    if constexpr (FOLD) {
      const uint32_t fe0 = psl[it.lp - fp0], fn = psl[it.lp - fp0 + 1] - fe0;
      if (fn <= out.plan_cap && tid < fn)
        lds_dma_b128((const void*)(fent_all + fe0 + tid), lds_byte_addr((const uint4*)&lds[plan_off] + buf * out.plan_cap + (tid & ~63u)));
    }
  };
  if (plan_dma_ok) plan_dma(item, 0);
  FB_ST_BEGIN(7);

  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
    FB_ST(7, 0);                     // (waits for the prefetched tile first)
    {
      if (item.tile != kk_tile) {
        load_chirp(item, kk);
        kk_tile = item.tile;
      }
      if constexpr (FOLD) {
        plan_fetch(item);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // forward transform over the rows nb of this tile, per bin j: x W_L^{nb*ka} (the twiddle between the two forward passes,
      // ka = a*M + j), radix-Fb butterfly -> kb in natural order = the tile's Fb channels; x chirp (Response::operate)
#pragma unroll
      for (uint32_t jq = 0; jq < NJ; jq++) {
        cx2 v[Fb];
#pragma unroll
        for (uint32_t nb = 0; nb < Fb; nb++) {
          const Abk r = raw[jq * Fb + nb];
          const bool odd = tid & 1u;
          const cf recv = swap1(odd ? r.a : r.b);       // the even lane hands over bin j + 1 of pol 0, the odd one bin j - 1 of pol 1
          v[nb] = make_cx2(odd ? recv : r.a, odd ? r.b : recv);
        }
        const uint32_t ka = (item.tile << LOGM) + tid + 512u * jq;
        uint32_t jw[4];
        cf t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) jw[q] = (ka << q) & (uint32_t)(L - 1);
        twiddles_big(t, jw, logL, tw, g.tw_lo);
        apply_powers<Fb>(v, t[0], t[1], t[2], t[3]);
        fftR<Fb, -1>(v);
#pragma unroll
        for (uint32_t kb = 0; kb < Fb; kb++) x[jq * Fb + kb] = cmuls(v[kb], kk[jq * Fb + kb]);
      }
    }
    FB_ST(7, 1);
    const bool more = next_item(++j, next);
    if (more) fetch(next, raw);
    FB_ST(7, 2);
    // rows -> bins: element (bin j, column 2*kb + pol) of the inverse transform's tile, as the stages exchange them
    __syncthreads();                     // every wave has finished with the previous tile's image (last stage / fold phase)
#pragma unroll
    for (uint32_t jq = 0; jq < NJ; jq++) {
      const uint32_t jb = tid + 512u * jq;
      const uint32_t e0 = jb << logT;
#pragma unroll
      for (uint32_t kb = 0; kb < Fb; kb++) {
        const cx2 q = x[jq * Fb + kb];
        *(float4*)&lds[lds_pad(e0 + 2 * kb)] = make_float4(q.x[0], q.x[1], q.y[0], q.y[1]);
      }
    }
    __syncthreads();
    FB_ST(7, 3);

    const uint32_t tile = item.tile;
    const uint64_t part = part0 + item.lp;
    [[maybe_unused]] TsPart tsp = {0, 0, 0, 0};
    [[maybe_unused]] float ts_carry_pre = 0.f;
    if constexpr (SEARCH) tsp = ts_part(out, (uint32_t)part, g.nkeep);
    const uint32_t fcr = (16u >> logT3) & 15u, fcs_r = ((g.nkeep + 15u - fcr) & ~15u) + fcr;
    const uint32_t fcs = ((2u * fcs_r) << logT3) <= PTS * blockDim.x ? fcs_r : g.nkeep;
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if constexpr (FOLD) {
        const uint32_t slo = col >> 1;
        const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t t = t0 + (int32_t)(k * pstride);
          if ((uint32_t)t >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          *(float4*)&lds[2 * (slo * fcs + (uint32_t)t)] = make_float4(r[0], r[1], r[2], r[3]);
        }
        return;
      }
      if constexpr (SEARCH) {
        const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t t = t0 + (int32_t)(k * pstride);
          if ((uint32_t)t >= g.nkeep) continue;
          ts_stage((float*)lds, out, tsp, col >> 1, (uint32_t)t, cx2_lo(v[k]), cx2_hi(v[k]));
        }
        return;
      }
      if (out.kind == 0) return;
      const uint32_t chan = out.chan0 + chan_of(tile, col >> 1);
      float* __restrict__ row = out.base + chan * out.chan_stride;
      const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
      float2* __restrict__ o2 = (float2*)(row + part * out.part_step) + t0;
      float4* __restrict__ o4 = (float4*)row + ((int64_t)(part * g.nkeep) + t0);
      if (out.kind == 1) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float2* o = o2 + k * pstride;
          st_stream(o, cx2_lo(v[k]));
          st_stream((float2*)((float*)o + out.pol_stride), cx2_hi(v[k]));
        }
      } else if (out.ndim == 4) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          st_stream(o4 + k * pstride, make_float4(r[0], r[1], r[2], r[3]));
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t ts = t0 + (int32_t)(k * pstride);
          if ((uint32_t)ts >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          const uint64_t idat = part * g.nkeep + (uint32_t)ts;
          if (out.ndim == 2) {
            st_stream(&((float2*)row)[idat], make_float2(r[0], r[1]));
            st_stream(&((float2*)(row + out.pol_stride))[idat], make_float2(r[2], r[3]));
          } else {
            row[idat] = r[0];
            row[out.pol_stride + idat] = r[1];
            row[2 * out.pol_stride + idat] = r[2];
            row[3 * out.pol_stride + idat] = r[3];
          }
        }
      }
    };
    uint32_t f_e0 = 0, f_nact = 0;
    const uint4* __restrict__ ent = nullptr;
    const uint4* planl = nullptr;
    bool in_lds = false;
    uint4 en_pre = make_uint4(0, 0, 0, 0), en_pre2 = make_uint4(0, 0, 0, 0);
    float4 acc_pre = make_float4(0.f, 0.f, 0.f, 0.f), acc_pre2 = acc_pre;
    constexpr bool PRE = FOLD && FftPlan<LOGM>::NS >= 2;
    constexpr bool PRE2 = PRE && LOGFB >= 3;              // >= 8 channels per tile: a thread may fold a second item
    if constexpr (FOLD) {
      f_e0 = fe0_cur;
      f_nact = fn_cur;
      ent = fent_all + f_e0;
      planl = (const uint4*)&lds[plan_off] + (jt & 1) * out.plan_cap;
      in_lds = plan_dma_ok && f_nact <= out.plan_cap;
    }
    const bool planes2 = FOLD && out.prof_planes == 2;
    const uint64_t plane = fseg == 0 ? out.plane_stride : 2ull * out.nbin;
    auto acc_row = [&](const uint32_t w) -> float* {
      const uint32_t cl = chan_of(tile, w & (T3 - 1));
      return (float*)(fseg == 0 ? (float4*)out.base + (uint64_t)(out.chan0 + cl) * out.prof_span4
                                : (float4*)out.part + ((uint64_t)(fseg - 1) * g.C + cl) * out.nbin);
    };
    auto acc_load = [&](float* row, const uint32_t b) -> float4 {
      if (planes2) {
        const float2 u = *(const float2*)(row + 2 * b), v = *(const float2*)(row + plane + 2 * b);
        return make_float4(u.x, u.y, v.x, v.y);
      }
      return *(const float4*)(row + 4 * b);
    };
    auto acc_store = [&](float* row, const uint32_t b, const float4 a) {
      if (planes2) {
        *(float2*)(row + 2 * b) = make_float2(a.x, a.y);
        *(float2*)(row + plane + 2 * b) = make_float2(a.z, a.w);
      } else {
        *(float4*)(row + 4 * b) = a;
      }
    };
    auto mid = [&](const int phase) {
      if constexpr (SEARCH) {
        // the open output sample's partial sum (this workgroup's store at the end of the previous part; FROM_LDS: the rows -> bins
        // exchange in front of the transform has passed two barriers since)
        const uint32_t npo = out.state == DSPSR_AMD_PPQQ ? 2u : 1u;
        if (phase == 1 && tsp.phi && tid < (npo << logT3))
          ts_carry_pre = ts_carry_load(out, (out.chan0 + chan_of(tile, tid / npo)) * npo + tid % npo);
      }
      if constexpr (PRE) {
        if (phase == 2 && in_lds && tid < (f_nact << logT3)) {
          en_pre = planl[tid >> logT3];
          acc_pre = acc_load(acc_row(tid), en_pre.x);
        }
        // (many channels per tile: a part's active bins x Fb channels exceed the workgroup, so a thread folds a second
        //  item -- its accumulator is requested here as well instead of costing a memory round trip in the fold phase)
        if constexpr (PRE2) {
          if (phase == 2 && in_lds && tid + 512u < (f_nact << logT3)) {
            en_pre2 = planl[(tid + 512u) >> logT3];
            acc_pre2 = acc_load(acc_row(tid + 512u), en_pre2.x);
          }
        }
        if (phase == 2 && plan_dma_ok && more) plan_dma(next, (jt + 1) & 1);
      }
    };
    wgfft<LOGM, +1, EPI != 0, true>(lds, ltw_off, tid, logT, x, store, mid);
    FB_ST(7, 4);
    if constexpr (SEARCH) {
      __syncthreads();                       // the tile's detected samples are staged
      const uint32_t npo = out.state == DSPSR_AMD_PPQQ ? 2u : 1u;
      ts_reduce((const float*)lds, out, tsp, npo << logT3, ts_carry_pre, true, tid, blockDim.x, [&](const uint32_t slo) { return chan_of(tile, slo); });
    }
    if constexpr (FOLD) {
      __syncthreads();
      const bool pre = PRE && in_lds;
      for (uint32_t w = tid; w < (f_nact << logT3); w += blockDim.x) {
        const uint32_t slo = w & (T3 - 1);
        uint4 en;
        float4 acc;
        if (pre && w == tid) {
          en = en_pre;
          acc = acc_pre;
        } else if (PRE2 && pre && w == tid + 512u) {
          en = en_pre2;
          acc = acc_pre2;
        } else {
          en = in_lds ? planl[w >> logT3] : ent[w >> logT3];
          acc = acc_load(acc_row(w), en.x);
        }
        const uint32_t nint = en.z >> 16;
        float* __restrict__ pp = acc_row(w);
        uint32_t off = en.w, hits = en.z & 0xffffu;
        for (uint32_t i = 0;;) {
          const float4* __restrict__ src = (const float4*)&lds[2 * (slo * fcs + off)];
          uint32_t h = 0;
          for (; h + 8 <= hits; h += 8) {
            float4 sm[8];
#pragma unroll
            for (int q = 0; q < 8; q++) sm[q] = src[h + q];
#pragma unroll
            for (int q = 0; q < 8; q++) { acc.x += sm[q].x; acc.y += sm[q].y; acc.z += sm[q].z; acc.w += sm[q].w; }
          }
          for (; h < hits; h++) {
            const float4 sm = src[h];
            acc.x += sm.x; acc.y += sm.y; acc.z += sm.z; acc.w += sm.w;
          }
          if (++i >= nint) break;
          const Interval iv = out.piv[en.y + i];
          off = (uint32_t)iv.offset; hits = iv.hits;
        }
        acc_store(pp, en.x, acc);
      }
    }
    FB_ST(7, 5);
    FB_ST_TILE(7, 6);
    if (!more) break;
    item = next;
    jt++;
  }
  FB_ST_END(7);
}

k1c_t fb_pick_col1() { return k_fwd_col1q<1>; }
k3_t fb_pick_rinv(int logm, int epi)
{
#define FB_RINV(M, B) (epi == 1 ? k_rows_inv<M, B, 1> : epi == 2 ? k_rows_inv<M, B, 2> : k_rows_inv<M, B, 0>)
  switch (logm) {
    case 9: return FB_RINV(9, 4);
    case 10: return FB_RINV(10, 3);
    case 11: return FB_RINV(11, 2);
    case 12: return FB_RINV(12, 1);
    default: return nullptr;
  }
#undef FB_RINV
}
void fb_launch_raw_cols(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0)
{
  hipLaunchKernelGGL(k_raw_cols, grid, dim3(256), 0, stream, g, in, Rt, part0);
}

}  // namespace dspsr_amd

FB_ST_READER(two_pass)
