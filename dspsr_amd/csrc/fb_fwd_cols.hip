// Convolving filterbank, forward pass 1 (and the regroup / sub-sequence helpers in front of it); see fb_common.h
#include "fb_common.h"

namespace dspsr_amd {

// ------------------------------------------------------------------------------------ P0
// 8-bit pre-transposition: P1 needs, for every na (stride Rr samples apart), the T1 adjacent samples of
// its tile -- 2*T1 bytes per 2*Rr-byte row.  Reading those straight from the block costs one 128-byte line
// per 8 useful bytes and per lane, so for 8-bit real dual-pol input the window of each part is first
// regrouped (2 bytes per sample pair, coalesced both ways through LDS) into
//   Rt[part][tile][na][T1]  (pol0,pol1) byte pairs
// Both the generic order and the CASPSR 4-sample interleave are accepted.
__global__ __launch_bounds__(256) void k_raw_transpose(const FbGeom g, const FbIn in, uint16_t* __restrict__ Rt,
                                                       const uint64_t part0)
{
  // block: 64 rows (na) x 256 columns (nb) of byte pairs; rows are read as 16-byte pieces (8 samples),
  // written as T-sample (2T-byte) pieces of 64 consecutive rows = 128*T contiguous bytes per tile
  constexpr uint32_t ROWS = 64, COLS = 256, PITCH = COLS / 2 + 1;       // 32-bit words per LDS row (+1: bank skew)
  __shared__ uint32_t sm[ROWS * PITCH];
  const uint32_t tid = threadIdx.x;
  const uint32_t M = 1u << g.logM, Rr = 1u << g.logR;
  const int logT = g.logT1;
  const uint32_t nb0 = blockIdx.x * COLS, na0 = blockIdx.y * ROWS;
  // complex dual-pol input (generic order, 4 bytes per sample: p0 re, p0 im, p1 re, p1 im): one polarisation = one
  // sequence per blockIdx.z, its (re, im) byte pairs take the place of the (pol0, pol1) pairs of real input
  const uint32_t nsq = g.real_input ? 1u : g.npol;
  const uint64_t part = blockIdx.z / nsq;
  const uint32_t seq = blockIdx.z % nsq;
  const uint64_t t0 = (part0 + part) * in.part_step;
  const uint32_t ncol = Rr - nb0 < COLS ? Rr - nb0 : COLS, nrow = M - na0 < ROWS ? M - na0 : ROWS;
  if (ncol % 8 == 0) {
    for (uint32_t q = tid; q < nrow * (ncol / 8); q += 256) {       // 8 samples (16 bytes) per thread and step
      const uint32_t r = q / (ncol / 8), c8 = (q % (ncol / 8)) * 8;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c8;   // multiple of 4 (8 unless t0 is odd*4)
      uint32_t w[4];
      if (!g.real_input) {                                             // 8 samples x 4 bytes, keep this polarisation
        const uint4* p = (const uint4*)((const uint8_t*)in.base + 4 * t);
        const uint4 s0 = p[0], s1 = p[1];
        const int sh = 16 * seq;
        w[0] = ((s0.x >> sh) & 0xffffu) | (((s0.y >> sh) & 0xffffu) << 16);
        w[1] = ((s0.z >> sh) & 0xffffu) | (((s0.w >> sh) & 0xffffu) << 16);
        w[2] = ((s1.x >> sh) & 0xffffu) | (((s1.y >> sh) & 0xffffu) << 16);
        w[3] = ((s1.z >> sh) & 0xffffu) | (((s1.w >> sh) & 0xffffu) << 16);
      } else if (in.kind == 2) {                                       // CASPSR: 4 B pol0 | 4 B pol1
        const uint32_t* p = (const uint32_t*)((const uint8_t*)in.base + (t >> 2) * 8);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const uint32_t p0 = p[2 * h], p1 = p[2 * h + 1];
          w[2 * h] = (p0 & 0xff) | ((p1 & 0xff) << 8) | ((p0 & 0xff00) << 8) | ((p1 & 0xff00) << 16);
          w[2 * h + 1] = ((p0 >> 16) & 0xff) | (((p1 >> 16) & 0xff) << 8) | ((p0 >> 24) << 16) | ((p1 >> 24) << 24);
        }
      } else {
        const uint32_t* p = (const uint32_t*)((const uint8_t*)in.base + 2 * t);
        w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; w[3] = p[3];
      }
#pragma unroll
      for (int h = 0; h < 4; h++) sm[r * PITCH + c8 / 2 + h] = w[h];
    }
  } else {                                                              // narrow problems: 2 samples per step
    for (uint32_t q = tid; q < nrow * (ncol / 2); q += 256) {
      const uint32_t r = q / (ncol / 2), c2 = (q % (ncol / 2)) * 2;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c2;
      uint32_t w;
      if (!g.real_input) {
        const uint32_t* p = (const uint32_t*)((const uint8_t*)in.base + 4 * t);
        w = ((p[0] >> (16 * seq)) & 0xffffu) | (((p[1] >> (16 * seq)) & 0xffffu) << 16);
      } else if (in.kind == 2) {
        const uint8_t* b = (const uint8_t*)in.base + (t >> 2) * 8 + (t & 3);
        w = (uint32_t)b[0] | ((uint32_t)b[4] << 8) | ((uint32_t)b[1] << 16) | ((uint32_t)b[5] << 24);
      } else {
        w = *(const uint32_t*)((const uint8_t*)in.base + 2 * t);
      }
      sm[r * PITCH + c2 / 2] = w;
    }
  }
  __syncthreads();
  uint32_t* __restrict__ dst = (uint32_t*)(Rt + (part * nsq + seq) * ((uint64_t)M << g.logR));
  const int logW = logT - 1;                        // 32-bit words per (row, tile) piece
  const uint32_t ntl = ncol >> logT, W = 1u << logW;
  if (logW == 1 && (nrow & 1) == 0) {
    // 4-column tiles (the headline geometry): two rows of a tile are 16 contiguous bytes of the output -- one
    // 16-byte store per lane instead of four 4-byte ones (narrow per-lane accesses stream slower on this chip,
    // tools/load_width_probe.hip)
    const uint32_t nr2 = nrow >> 1;
    for (uint32_t q = tid; q < ntl * nr2; q += 256) {
      const uint32_t r = (q % nr2) * 2, tl = q / nr2;
      const uint32_t* s0 = &sm[r * PITCH + 2 * tl];
      const uint32_t* s1 = s0 + PITCH;
      *(uint4*)&dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << 1)] = make_uint4(s0[0], s0[1], s1[0], s1[1]);
    }
    return;
  }
  if (logW >= 2) {                                                   // tiles of >= 8 columns: 16 bytes of a row piece per lane
    const int logV = logW - 2;
    for (uint32_t q = tid; q < (ntl * nrow) << logV; q += 256) {
      const uint32_t v4 = q & ((1u << logV) - 1), r = (q >> logV) % nrow, tl = (q >> logV) / nrow;
      const uint32_t* s0 = &sm[r * PITCH + (tl << logW) + 4 * v4];
      *(uint4*)&dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << logW) + 4 * v4] = make_uint4(s0[0], s0[1], s0[2], s0[3]);
    }
    return;
  }
  for (uint32_t q = tid; q < ntl * nrow * W; q += 256) {           // [tile][row][word]: runs of nrow*T pairs
    const uint32_t wd = q & (W - 1), r = (q >> logW) % nrow, tl = (q >> logW) / nrow;
    dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << logW) + wd] = sm[r * PITCH + (tl << logW) + wd];
  }
}

// The same regrouping for float32 input -- what dsp::Filterbank::Engine::perform is handed by DSPSR (the input is unpacked
// before the boundary): T1 adjacent samples of a row are 4*T1 bytes per polarisation row, 16-byte pieces 8 KB apart at the
// headline geometry, and pass 1 reading them in place ran four times slower than from 8-bit data (2223 against 562 us per
// 32 parts).  Elements are 8 bytes: (pol0, pol1) of a real sample pair, or (re, im) of one polarisation of complex input
//   Rt[part][seq][tile][na][T1]   (lives in the X scratch, which is idle until pass 2 writes it)
__global__ __launch_bounds__(256) void k_float_transpose(const FbGeom g, const FbIn in, cf* __restrict__ Rt, const uint64_t part0)
{
  constexpr uint32_t ROWS = FB_FT_ROWS, COLS = FB_FT_COLS, PITCH = COLS + 1;
  __shared__ cf sm[ROWS * PITCH];
  const uint32_t tid = threadIdx.x;
  const uint32_t M = 1u << g.logM, Rr = 1u << g.logR;
  const int logT = g.logT1;
  const uint32_t nb0 = blockIdx.x * COLS, na0 = blockIdx.y * ROWS;
  const uint32_t nsq = g.real_input ? 1u : g.npol;
  const uint64_t part = blockIdx.z / nsq;
  const uint32_t seq = blockIdx.z % nsq;
  const uint64_t t0 = (part0 + part) * in.part_step;
  const uint32_t ncol = Rr - nb0 < COLS ? Rr - nb0 : COLS, nrow = M - na0 < ROWS ? M - na0 : ROWS;   // ncol % 4 == 0 (host)
  const float* __restrict__ x = (const float*)in.base;
  if (g.real_input) {
    for (uint32_t q = tid; q < nrow * (ncol / 4); q += 256) {            // 4 samples of both polarisations per step
      const uint32_t r = q / (ncol / 4), c4 = (q % (ncol / 4)) * 4;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c4;
      const float4 p0 = ld_stream((const float4*)(x + t)), p1 = ld_stream((const float4*)(x + in.pol_stride + t));
      cf* d = &sm[r * PITCH + c4];
      d[0] = make_float2(p0.x, p1.x); d[1] = make_float2(p0.y, p1.y); d[2] = make_float2(p0.z, p1.z); d[3] = make_float2(p0.w, p1.w);
    }
  } else {
    for (uint32_t q = tid; q < nrow * (ncol / 2); q += 256) {            // 2 complex samples per step
      const uint32_t r = q / (ncol / 2), c2 = (q % (ncol / 2)) * 2;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c2;
      const float4 v = ld_stream((const float4*)(x + seq * in.pol_stride + 2 * t));
      cf* d = &sm[r * PITCH + c2];
      d[0] = make_float2(v.x, v.y); d[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();
  cf* __restrict__ dst = Rt + (part * nsq + seq) * ((uint64_t)M << g.logR);
  const uint32_t ntl = ncol >> logT, T = 1u << logT;                     // T >= 2: two elements (16 bytes) per lane
  for (uint32_t q = tid; q < ntl * nrow * (T / 2); q += 256) {
    const uint32_t h = q % (T / 2), r = (q / (T / 2)) % nrow, tl = q / ((T / 2) * nrow);
    const cf* s0 = &sm[r * PITCH + (tl << logT) + 2 * h];
    st_stream((float4*)&dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << logT) + 2 * h],
              make_float4(s0[0].x, s0[0].y, s0[1].x, s0[1].y));
  }
}

// ------------------------------------------------------------------------------------ P1
// M-point forward FFTs down T1 adjacent stride-Rr columns of one sequence of one part.
//   in : sample n = na*Rr + nb (8-bit or float32, converted on load), nb = tile*T1 + col
//   out: A[ka/T2][nb][ka%T2] = W_L^{nb*ka} * sum_na w[na*Rr+nb] W_M^{na*ka}
// Persistent: each workgroup walks its items (tile fastest, then sequence, then part) and
// prefetches the raw samples of the next item while transforming the current one.
// LOGT >= 0: the number of columns per tile (2^LOGT) is a compile-time constant (the usual full-size tile,
// LOGT = 14 - LOGF), so every LDS address and stride folds into immediates; LOGT = -1: taken from the geometry.
template <int LOGF, int RAWW, int LOGT>
__global__ __launch_bounds__(512) void k_fwd_cols(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                  const cf* __restrict__ tw, const uint64_t part0,
                                                  const uint32_t nparts, const uint32_t nseq, const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logT = LOGT >= 0 ? LOGT : g.logT1, logT2 = g.logT2;
  const uint32_t T = 1u << logT, T2 = 1u << logT2;
  const int logL = LOGF + g.logR;          // g.logM == LOGF
  const uint64_t L = 1ull << logL;
  const uint32_t ntile = 1u << (g.logR - logT);
  const uint32_t total = ntile * nseq * nparts;
  const int logNt = g.logR - logT;          // ntile = 2^logNt ; nseq is 1 or 2
  auto seq_of = [&](const uint32_t rest) { return nseq == 2 ? (rest & 1u) : 0u; };
  auto part_of = [&](const uint32_t rest) { return nseq == 2 ? (rest >> 1) : rest; };

  auto fetch = [&](const uint32_t item, RawW<RAWW> (&raw)[PTS / 2]) {
    const uint32_t tile = item & (ntile - 1);
    const uint32_t rest = item >> logNt;
    uint32_t seq = seq_of(rest);
    const bool pret = in.kind == 3 || in.kind == 5;   // pre-transposed: [part][tile][na][T] pairs, contiguous per tile
    uint64_t t0 = pret ? ((uint64_t)rest * ntile + tile) * ((uint64_t)T << LOGF)       // rest = part*nseq + seq
                       : (part0 + part_of(rest)) * in.part_step + tile * T;
    if constexpr (RAWW == 4) {
      // channel-batched convolution (FbIn::batch, float32 complex rows): `rest` = (part * npol + pol) * batch + channel
      if (in.batch) {                                  // uniform
        const uint32_t c = rest % in.batch, ps = rest / in.batch;
        seq = ps % (uint32_t)g.npol;
        t0 = (part0 + ps / (uint32_t)g.npol) * in.part_step + c * in.chan_stride_c + tile * T;
      }
    }
    // element i of a thread's first-stage butterfly is row na = nab + i*MS of one column pair: sample index =
    // base + i*step with a wave-uniform step (no per-element index arithmetic or branches between the loads)
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const uint64_t step = pret ? ((uint64_t)MS << logT) : ((uint64_t)MS << g.logR);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;               // element of the tile: row eb >> logT, column eb % T
      const uint64_t tb = t0 + (eb & (T - 1)) + (pret ? (uint64_t)((eb >> logT) << logT) : (((uint64_t)(eb >> logT)) << g.logR));
#pragma unroll
      for (int i = 0; i < P::R1; i++) raw[(g2 / 2) * P::R1 + i] = fetch_pair<RAWW>(g, in, seq, tb + i * step);
    }
  };

  // exchange buffer, then the stage twiddle tables (16-byte aligned)
  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, threadIdx.x, blockDim.x);
  // copy-out of the staged tile (see the end of the tile loop): thread part of the addresses, once per kernel
  const uint32_t co_swz = (PTS * blockDim.x) >= 256 ? 1u : 0u;
  const uint32_t co_l0 = 2 * threadIdx.x;
  const uint32_t co_n2 = LOGT >= 0 ? (2u << (LOGF + LOGT - LOG_PTS)) : 2 * blockDim.x;   // full tiles: a constant
  const int co_sh = logT + logT2;
  const bool co_fast = (co_n2 & 63) == 0 && (co_n2 >> co_sh) != 0 && (co_n2 & ((1u << co_sh) - 1)) == 0;   // uniform
  const uint32_t co_lds = lds_pad(co_l0 ^ (((co_l0 >> 4) & co_swz) << 3)), co_lstep = co_n2 + ((co_n2 >> 6) << 2);
  const uint32_t co_goff = (uint32_t)(((((uint64_t)(co_l0 >> co_sh) << g.logR) << logT2) + (co_l0 & ((1u << co_sh) - 1))) * sizeof(cf));
  const uint64_t co_gstep = ((uint64_t)(co_n2 >> co_sh) << g.logR) << logT2;       // elements of A per pair step
  auto copy_out = [&](const uint32_t tile, cf* __restrict__ Aseq) {
    const uint32_t swz = co_swz;
    const uint32_t nthr = blockDim.x;
    if (co_fast) {
      // pair jj of a thread is pair 0 plus jj*2*nthr elements: a constant step in the padded image (co_lstep) and a
      // uniform step in A (co_gstep) -- one LDS address and one 32-bit global offset per THREAD, computed before the
      // tile loop; the per-pair part is an immediate / a scalar-register base (this loop issued 23 % of the pass's
      // vector instructions as per-pair address arithmetic, 64-bit shifts included)
      const char* __restrict__ gb = (const char*)(Aseq + ((uint64_t)(tile * T) << logT2));
#pragma unroll
      for (int j4 = 0; j4 < PTS / 2; j4 += 4) {                  // four LDS reads in flight, then their stores
        float4 pr[4];
#pragma unroll
        for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[co_lds + (j4 + q) * co_lstep];
        __builtin_amdgcn_sched_barrier(0);                         // (the min-register scheduler would pair every read with its store)
#pragma unroll
        for (int q = 0; q < 4; q++) st_stream((float4*)(gb + (uint64_t)(j4 + q) * co_gstep * sizeof(cf) + co_goff), pr[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 4
      for (int jj = 0; jj < PTS / 2; jj++) {
        const uint32_t l = 2 * (tid + jj * nthr);                  // element index inside the staged image
        const uint32_t blkA = l >> (logT + logT2), within = l & ((1u << (logT + logT2)) - 1);
        const float4 pr = *(const float4*)&lds[lds_pad(l ^ (((l >> 4) & swz) << 3))];
        st_stream((float4*)&Aseq[((((uint64_t)blkA << g.logR) + tile * T) << logT2) + within], pr);
      }
    }
  };
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  RawW<RAWW> raw[PTS / 2];
  fetch(item, raw);
  FB_ST_BEGIN(1);
  for (;;) {
    asm volatile("" : "+v"(tid));   // per-tile index math stays inside the loop (see wgfft)
    cx2 x[NPAIR];
    FB_ST(1, 0);                     // (waits for the prefetched tile first)
    const uint32_t seq_cur = seq_of(item >> logNt);
#pragma unroll
    for (int h = 0; h < NPAIR; h++) {
      cf a, b;
      decode_pair<RAWW>(g, in, raw[h], a, b, seq_cur);
      x[h] = make_cx2(a, b);
    }
    FB_ST(1, 1);
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    if (more) fetch(next, raw);
    FB_ST(1, 2);

    const uint32_t tile = item & (ntile - 1);
    cf* __restrict__ Aseq = A + (uint64_t)(item >> logNt) * L;                 // sequence part*nseq + seq
    // last-stage outputs go to LDS in A-layout order [ka/T2][col][ka%T2]; after a barrier the tile is
    // written out as whole runs of T*T2 elements with 16-byte-per-lane stores.  The image is XOR-swizzled
    // (bit 3 ^= bit 4; pairs of elements stay together) so that the 8-byte scatter of a wave spreads over all
    // banks (17 % of this pass's LDS cycles were bank conflicts, profiles/r01d_lds_conflicts.txt).
    // (the twiddle W_L^{nb*ka} between the two forward passes is applied by pass 2, on load: see k_fwd_rows)
    const uint32_t swz = (PTS * blockDim.x) >= 256 ? 1u : 0u;
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      // image index of element k: l0 + k*(pstride << logT) (pstride is a multiple of T2), so when that step is a
      // multiple of 64 the swizzle and the padding of l0 carry over: one address per column, constant offsets
      auto img = [&](const uint32_t l) { return lds_pad(l ^ (((l >> 4) & swz) << 3)); };
      const uint32_t l0 = ((((p >> logT2) << logT) + col) << logT2) | (p & (T2 - 1));
      const uint32_t step = pstride << logT;
      const bool aff = (step & 63) == 0 && (pstride & (T2 - 1)) == 0;
      const uint32_t b0 = img(l0), b1 = img(l0 + T2), sp = step + (step >> 4);
      if (aff) {                                       // uniform
#pragma unroll
        for (int k = 0; k < R; k++) {
          float* __restrict__ d0 = (float*)&lds[b0 + k * sp];
          float* __restrict__ d1 = (float*)&lds[b1 + k * sp];
          d0[0] = v[k].x[0]; d0[1] = v[k].y[0];       // (re, im) of column col   (two dwords: no register shuffling)
          d1[0] = v[k].x[1]; d1[1] = v[k].y[1];       // column col + 1
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t ka = k * pstride + p;
          const uint32_t l = ((((ka >> logT2) << logT) + col) << logT2) | (ka & (T2 - 1));
          lds[img(l)] = cx2_lo(v[k]);
          lds[img(l + T2)] = cx2_hi(v[k]);
        }
      }
    };
    wgfft<LOGF, -1, true>(lds, ltw_off, tid, logT, x, store);
    __syncthreads();
    FB_ST(1, 3);
    copy_out(tile, Aseq);
    FB_ST(1, 4);
    FB_ST_TILE(1, 5);
    if (!more) break;
    item = next;
  }
  FB_ST_END(1);
}

// ------------------------------------------------------------------------------------ P1, paired tiles (round 3)
// Long transforms (L = Fa*Fb >= 2^25: -F 64:D at the optimal response length, dsp::Convolution shapes) leave pass 1 tiles of
// T1 = 2 columns and pass 2 tiles of T2 = 4 rows, so a pass-1 tile meets a pass-2 tile in T1*T2 = 8 elements: 64-byte runs
// of A, half a cache line per store run -- pass 1 then moves its bytes at 2.7 TB/s where the same bytes in 256-byte runs
// (headline geometry) go at 5.0 (tools/run_length_probe.hip: stores in 64-byte runs 3.2-3.4 TB/s at any stride, 128-byte runs
// 4.6-4.8).  Here a work item is a PAIR of adjacent tiles (columns 4j .. 4j+3): the two are transformed one after the other,
// the outputs of the first wait in registers (64) while the second runs through the one exchange buffer, and the four
// columns are then staged and copied out together, half the rows at a time (the buffer holds 2^14 elements: 4 columns x
// Fa/2 rows) -- runs of 2*T1*T2 elements, whole 128-byte lines.  The layout of A and everything behind it are unchanged.
// Full-size tiles of two columns only (Fa = 2^13 at 2^14 points per workgroup): the last stage is the radix-2 one, so the two
// outputs of a butterfly are row ka (lower half) and ka + Fa/2 (upper half).
struct KeepOut {
  cx2* o;
  uint32_t p0;
  int h;
  template <int R> DEV void operator()(const uint32_t, const uint32_t p, const uint32_t, cx2 (&v)[R])
  {
    static_assert(R == 2, "k_fwd_cols_dual: radix-2 last stage");
    if (h == 0) p0 = p;
    o[2 * h] = v[0];
    o[2 * h + 1] = v[1];
  }
};
template <int RAWW>
__global__ __launch_bounds__(512) void k_fwd_cols_dual(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                       const cf* __restrict__ tw, const uint64_t part0,
                                                       const uint32_t nparts, const uint32_t nseq, const uint32_t run)
{
  constexpr int LOGF = 13, LOGT = 1;
  typedef FftPlan<LOGF> P;
  static_assert(P::REM == 1 && PTS / 2 / 2 == 8, "k_fwd_cols_dual: 2^13-point columns, radix-2 last stage");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logT2 = g.logT2;
  const uint32_t T = 2, T2 = 1u << logT2;
  const int logL = LOGF + g.logR;
  const uint64_t L = 1ull << logL;
  const int logNp = g.logR - LOGT - 1;                  // pairs of tiles per sequence
  const uint32_t npair = 1u << logNp, ntile = npair << 1;
  const uint32_t total = npair * nseq * nparts;
  auto seq_of = [&](const uint32_t rest) { return nseq == 2 ? (rest & 1u) : 0u; };
  auto part_of = [&](const uint32_t rest) { return nseq == 2 ? (rest >> 1) : rest; };
  auto fetch = [&](const uint32_t item, const uint32_t sub, RawW<RAWW> (&raw)[PTS / 2]) {
    const uint32_t tile = ((item & (npair - 1)) << 1) | sub;
    const uint32_t rest = item >> logNp;
    const uint32_t seq = seq_of(rest);
    const bool pret = in.kind == 3 || in.kind == 5;   // pre-transposed: [part][tile][na][T] pairs, contiguous per tile
    const uint64_t t0 = pret ? ((uint64_t)rest * ntile + tile) * ((uint64_t)T << LOGF) : (part0 + part_of(rest)) * in.part_step + tile * T;
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const uint64_t step = pret ? ((uint64_t)MS << LOGT) : ((uint64_t)MS << g.logR);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;               // row eb >> 1, column eb & 1 (= 0)
      const uint64_t tb = t0 + (eb & (T - 1)) + (pret ? (uint64_t)((eb >> LOGT) << LOGT) : (((uint64_t)(eb >> LOGT)) << g.logR));
#pragma unroll
      for (int i = 0; i < P::R1; i++) raw[(g2 / 2) * P::R1 + i] = fetch_pair<RAWW>(g, in, seq, tb + i * step);
    }
  };
  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, threadIdx.x, blockDim.x);
  // staged image of one half: A-layout order l = ((ka_local / T2) * 4 + column) * T2 + ka % T2, padded by two elements
  // per 32 (a thread stages 32 consecutive elements: with the exchange buffer's padding of 4 per 64 the lanes of a wave would
  // meet in 8 banks); same size as the exchange buffer
  auto img = [](const uint32_t l) { return l + ((l >> 5) << 1); };
  // copy-out: 16-byte unit u = tid + 512*jj -> image element 2u, A element ((l >> sh) << logR << logT2) + (l & mask)
  const int sh = 2 + logT2;
  const uint32_t co_l0 = 2 * threadIdx.x;
  const uint32_t co_lds = img(co_l0), co_lstep = img(1024);                                   // 1024 is a multiple of 32
  const uint64_t co_goff = (((uint64_t)(co_l0 >> sh) << g.logR) << logT2) + (co_l0 & ((1u << sh) - 1));
  const uint64_t co_gstep = ((uint64_t)(1024u >> sh) << g.logR) << logT2;
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  RawW<RAWW> raw0[PTS / 2], raw1[PTS / 2];
  fetch(item, 0, raw0);
  fetch(item, 1, raw1);
  for (;;) {
    asm volatile("" : "+v"(tid));
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    const uint32_t seq_cur = seq_of(item >> logNp);
    cx2 o0[PTS / 2], o1[PTS / 2];
    uint32_t p0;
    {
      cx2 x[NPAIR];
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        cf a, b;
        decode_pair<RAWW>(g, in, raw0[h], a, b, seq_cur);
        x[h] = make_cx2(a, b);
      }
      if (more) fetch(next, 0, raw0);
      KeepOut keep{o0, 0u, 0};
      wgfft<LOGF, -1, false>(lds, ltw_off, tid, LOGT, x, keep);
      p0 = keep.p0;
    }
    {
      cx2 x[NPAIR];
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        cf a, b;
        decode_pair<RAWW>(g, in, raw1[h], a, b, seq_cur);
        x[h] = make_cx2(a, b);
      }
      if (more) fetch(next, 1, raw1);
      KeepOut keep{o1, 0u, 0};
      wgfft<LOGF, -1, false>(lds, ltw_off, tid, LOGT, x, keep);
    }
    const uint32_t pair = item & (npair - 1);
    cf* __restrict__ Aseq = A + (uint64_t)(item >> logNp) * L + ((uint64_t)(pair * 4) << logT2);
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      __syncthreads();                 // the exchange buffer (second transform's last stage / the other half's copy-out) has been read
#pragma unroll
      for (int h = 0; h < PTS / 4; h++) {
        const uint32_t p = p0 + h;                                             // row of the half
        const uint32_t l = (((p >> logT2) << 2) << logT2) | (p & (T2 - 1));     // column 0 of the four
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          const cx2 v = s2 ? o1[2 * h + hh] : o0[2 * h + hh];
          float* __restrict__ d0 = (float*)&lds[img(l + ((2 * s2) << logT2))];
          float* __restrict__ d1 = (float*)&lds[img(l + ((2 * s2 + 1) << logT2))];
          d0[0] = v.x[0]; d0[1] = v.y[0];
          d1[0] = v.x[1]; d1[1] = v.y[1];
        }
      }
      __syncthreads();
      const char* __restrict__ gb = (const char*)(Aseq + (uint64_t)hh * (L >> 1) + co_goff);
#pragma unroll
      for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
        float4 pr[4];
#pragma unroll
        for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[co_lds + (j4 + q) * co_lstep];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; q++) st_stream((float4*)(gb + (uint64_t)(j4 + q) * co_gstep * sizeof(cf)), pr[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!more) break;
    item = next;
  }
}
// ------------------------------------------------------------------------------------ nchan_subband = 3 * 2^k, 5 * 2^k
// dsp::Filterbank takes whatever length FFTW / cuFFT plans (Filterbank.C:107-155, FilterbankCUDA.cu:92-116), e.g. -F 96:D.
// Here the transform tiles are powers of two; a forward transform of L = R * L' points (R = 3 or 5) is computed as R
// interleaved sub-sequences w_c[m] = w[R m + c] -- each an ordinary power-of-two forward transform F_c (passes 0-2 unchanged) --
// and one radix-R step:  X[k + q L'] = sum_c W_R^(c q) W_L^(c k) F_c[k],  k < L', q < R.  Bin k + q L' lies in spectrum row
// q * Rr' + k / M: the R combined bands, stored one after the other in the power-of-two X layout, ARE the R * Rr' rows the
// inverse pass walks (k_inv_chan: rows nsub << logR).
//   k_sub_split   : the launch group's samples de-interleaved into R contiguous single-channel blocks (generic byte order /
//                   float rows), so that passes 0-2 see ordinary inputs
//   k_sub_combine : the radix-R step in place on the R sub-spectra of every (part, sequence)
// Element j of sub-sequence c of window lp is input sample t_first + lp*win_step + R*j + c.
// KIND / EB: source form and bytes per element -- 1: generic 8-bit order, one element = the EB = npol*ndim bytes of a sample;
// 2: CASPSR (4 B pol0 | 4 B pol1), EB = 2; 0: float32 rows, one element = the ndim floats of one polarisation (EB = 4 ndim).
// Fast form (FAST): a thread takes n*R consecutive input samples (n = 16 / EB) and writes, for every c, the n elements they hold
// as ONE 16-byte store -- a ninth of the memory instructions of the element-wise form (one load and one store per BYTE), which
// moved the input at 2.3 TB/s (profiles/r05_experiments.txt item 3).  The element-wise form keeps the tails and odd shapes.
template <int KIND, int EB> struct SplitElem { typedef uint32_t type; };
template <> struct SplitElem<0, 8> { typedef uint2 type; };
template <int KIND, int EB>
DEV typename SplitElem<KIND, EB>::type sub_split_load(const SubSplit& p, const uint32_t q, const uint64_t t)
{
  if constexpr (KIND == 0) {
    const float* __restrict__ x = (const float*)p.base + p.chan_off + q * p.pol_stride;
    if constexpr (EB == 8) return *(const uint2*)(x + 2 * t);
    else return __float_as_uint(x[t]);
  } else if constexpr (KIND == 2) {
    const uint8_t* __restrict__ b = (const uint8_t*)p.base + (t >> 2) * 8 + (t & 3);
    return (uint32_t)b[0] | ((uint32_t)b[4] << 8);
  } else {
    const uint8_t* __restrict__ b = (const uint8_t*)p.base + (t * p.nchan + p.ichan) * EB;
    if constexpr (EB == 4) return ((uintptr_t)b & 3) ? ((uint32_t)*(const uint16_t*)b | ((uint32_t)*(const uint16_t*)(b + 2) << 16)) : *(const uint32_t*)b;
    else if constexpr (EB == 2) return *(const uint16_t*)b;
    else return *b;
  }
}
template <int KIND, int EB, int RT>
__global__ __launch_bounds__(256) void k_sub_split(const SubSplit p, uint8_t* __restrict__ out)
{
  constexpr uint32_t N = 16 / EB;                                   // elements per 16-byte store
  constexpr uint32_t R = RT;
  const uint32_t nrow = KIND == 0 ? p.npol : 1u;                    // float: one row of elements per polarisation
  const uint64_t groups = p.wlen / N;                               // whole 16-byte groups per window and sub-sequence
  const uint64_t nunit = (uint64_t)p.nwin * nrow * groups;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < nunit; u += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t gq = u % groups, rest = u / groups;
    const uint32_t q = (uint32_t)(rest % nrow), lp = (uint32_t)(rest / nrow);
    const uint64_t m0 = gq * N, t0 = p.t_first + lp * p.win_step + R * m0;
    const uint64_t oj = (uint64_t)lp * p.wlen + m0;                  // element index inside the sub-sequence
    {
      typename SplitElem<KIND, EB>::type e[N * RT];
      // 8-bit sources: the thread's 16 R input bytes as whole dwords where the alignment allows (a load per BYTE otherwise)
      bool have = false;
      if constexpr (KIND == 2) {
        if ((t0 & 3) == 0) {                                        // whole 4-sample groups: 4 B pol0 | 4 B pol1
          const uint2* __restrict__ gp = (const uint2*)((const uint8_t*)p.base + (t0 >> 2) * 8);
          uint2 grp[2 * RT];
#pragma unroll
          for (uint32_t i = 0; i < 2 * RT; i++) grp[i] = gp[i];
#pragma unroll
          for (uint32_t i = 0; i < N * RT; i++)
            e[i] = ((grp[i >> 2].x >> (8 * (i & 3))) & 0xffu) | (((grp[i >> 2].y >> (8 * (i & 3))) & 0xffu) << 8);
          have = true;
        }
      } else if constexpr (KIND == 1) {
        const uint8_t* __restrict__ b = (const uint8_t*)p.base + t0 * EB;
        if (p.nchan == 1 && ((uintptr_t)b & 3) == 0) {
          uint32_t w[4 * RT];
#pragma unroll
          for (uint32_t i = 0; i < 4 * RT; i++) w[i] = ((const uint32_t*)b)[i];
#pragma unroll
          for (uint32_t i = 0; i < N * RT; i++) {
            if constexpr (EB == 4) e[i] = w[i];
            else if constexpr (EB == 2) e[i] = (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            else e[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
          }
          have = true;
        }
      }
      if (!have) {
#pragma unroll
        for (uint32_t i = 0; i < N * RT; i++) e[i] = sub_split_load<KIND, EB>(p, q, t0 + i);
      }
#pragma unroll
      for (uint32_t c = 0; c < (uint32_t)RT; c++) {
        uint4 v;
        if constexpr (EB == 8) v = make_uint4(e[c].x, e[c].y, e[RT + c].x, e[RT + c].y);
        else if constexpr (EB == 4) v = make_uint4(e[c], e[RT + c], e[2 * RT + c], e[3 * RT + c]);
        else if constexpr (EB == 2) v = make_uint4(e[c] | (e[RT + c] << 16), e[2 * RT + c] | (e[3 * RT + c] << 16),
                                                   e[4 * RT + c] | (e[5 * RT + c] << 16), e[6 * RT + c] | (e[7 * RT + c] << 16));
        else {
          uint32_t w[4];
#pragma unroll
          for (int k = 0; k < 4; k++) w[k] = e[(4 * k) * RT + c] | (e[(4 * k + 1) * RT + c] << 8) | (e[(4 * k + 2) * RT + c] << 16) | (e[(4 * k + 3) * RT + c] << 24);
          v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *(uint4*)(out + (uint64_t)c * p.sub_stride + ((uint64_t)q * p.nper + oj) * EB) = v;
      }
    }
  }
}
// element-wise form: elements [m_lo, wlen) of every window and sub-sequence (everything when the fast form does not apply)
template <int KIND, int EB>
__global__ __launch_bounds__(256) void k_sub_split_tail(const SubSplit p, uint8_t* __restrict__ out, const uint64_t m_lo)
{
  const uint32_t nrow = KIND == 0 ? p.npol : 1u;
  const uint64_t per = (p.wlen - m_lo) * p.R, n = (uint64_t)p.nwin * nrow * per;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = i % per, rest = i / per;
    const uint32_t q = (uint32_t)(rest % nrow), lp = (uint32_t)(rest / nrow);
    const uint64_t m = m_lo + w / p.R;
    const uint32_t c = (uint32_t)(w % p.R);
    const auto e = sub_split_load<KIND, EB>(p, q, p.t_first + lp * p.win_step + p.R * m + c);
    uint8_t* __restrict__ o = out + (uint64_t)c * p.sub_stride + ((uint64_t)q * p.nper + (uint64_t)lp * p.wlen + m) * EB;
    if constexpr (EB == 8) *(uint2*)o = e;
    else if constexpr (EB == 4) *(uint32_t*)o = e;
    else if constexpr (EB == 2) *(uint16_t*)o = (uint16_t)e;
    else *o = (uint8_t)e;
  }
}

// MSUB (freq_res = R * 2^k): the combined spectrum goes to a second buffer in PSEUDO-CHANNEL order -- bin R m' + r of channel c is
// bin m' of row c*R + r -- and, for real input, the mirror bins L - k where the inverse pass looks for them: row Rr-1-s, bin
// M' - m' (m' >= 1), row Rr - s, bin 0 (m' = 0).  mo = the caller's freq_res (R * M').
// rm = the factor of freq_res (the radix R of this kernel is nsub = rm times the odd factor of nchan_subband).
template <int R, bool MSUB>
__global__ __launch_bounds__(256) void k_sub_combine(const FbGeom g, cf* __restrict__ X, const uint32_t nseqs /* parts x sequences */,
                                                     cf* __restrict__ Xout, const uint32_t mo, const uint32_t rm, const OddTw wr)
{
  const int logLs = g.logM + g.logR;                     // sub-sequence length L'
  const uint32_t Ls = 1u << logLs, L = Ls * R;
  const uint32_t X3m = (1u << g.logX3) - 1, Mm = (1u << g.logM) - 1;
  const uint64_t n = (uint64_t)nseqs << logLs;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t o = (uint32_t)(i & (Ls - 1));
    cf* __restrict__ base = X + (i >> logLs) * (uint64_t)L + o;
    // X layout: offset o = ((s' >> logX3) * M + m) << logX3 | s' % X3  ->  bin k = s' * M + m
    const uint32_t t = o >> g.logX3, m = t & Mm, sp = ((t >> g.logM) << g.logX3) | (o & X3m);
    const uint32_t k = (sp << g.logM) + m;
    cf gq[R];
    gq[0] = base[0];
#pragma unroll
    for (int c = 1; c < R; c++) {
      // W_L^(c k), L = R L' (c k < 15 * 2^27: 32 bits; the multiple of L in it drops out of twiddle_odd's a = ... mod R)
      gq[c] = cmul(base[(uint64_t)c << logLs], twiddle_odd<R>((uint32_t)c * k, logLs, wr));
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
      cf acc = gq[0];
#pragma unroll
      for (int c = 1; c < R; c++) {
        const cf v = cmul(gq[c], wr.w[(c * q) % R]);
        acc.x += v.x; acc.y += v.y;
      }
      if constexpr (!MSUB) {
        base[(uint64_t)q << logLs] = acc;
      } else {
        const uint32_t kk = k + ((uint32_t)q << logLs);                      // natural bin of the whole transform
        const uint32_t Rr = (uint32_t)R << g.logR, N = g.real_input ? L >> 1 : L;
        const bool up = kk > N;                                               // (real input) a mirror bin
        const uint32_t kq = up ? L - kk : kk;
        const uint32_t cc = kq / mo, mm = kq - cc * mo, mi = mm / rm, r = mm - mi * rm, s = cc * rm + r;
        uint32_t row, bin;
        if (kk == N && g.real_input) { row = Rr >> 1; bin = 0; }              // (never read: the slot nothing else uses)
        else if (!up) { row = s; bin = mi; }
        else if (mi) { row = Rr - 1 - s; bin = (1u << g.logM) - mi; }
        else { row = Rr - s; bin = 0; }
        Xout[(i >> logLs) * (uint64_t)L + (((((uint64_t)(row >> g.logX3) << g.logM) + bin) << g.logX3) | (row & X3m))] = acc;
      }
    }
  }
}

template <int... I> static k1_t pick1(int logf, int raww, bool full, iseq<I...>)
{
  static const k1_t t4[] = {k_fwd_cols<I, 4, -1>...};
  static const k1_t t1[] = {k_fwd_cols<I, 1, -1>...};
  static const k1_t f4[] = {k_fwd_cols<I, 4, full_logt(I)>...};
  static const k1_t f1[] = {k_fwd_cols<I, 1, full_logt(I)>...};
  return full ? (raww == 1 ? f1[logf] : f4[logf]) : (raww == 1 ? t1[logf] : t4[logf]);
}
k1_t fb_pick1(int logf, int raww, bool full) { return pick1(logf, raww, full, seq_t()); }
// (8-bit input only: with float32 input the two tiles' prefetch alone is 128 registers)
k1_t fb_pick1_dual(int raww) { return raww == 1 ? k_fwd_cols_dual<1> : nullptr; }

void fb_launch_raw_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0)
{
  hipLaunchKernelGGL(k_raw_transpose, grid, dim3(256), 0, stream, g, in, Rt, part0);
}
void fb_launch_float_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, cf* Rt, uint64_t part0)
{
  hipLaunchKernelGGL(k_float_transpose, grid, dim3(256), 0, stream, g, in, Rt, part0);
}
// Any odd factor (run-time R <= ODD_MAX: 11, 13, 21, 25 ... -- dspsr -F 400:D, -F 25:D): one thread per OUTPUT bin kk = k + q L',
// X[kk] = sum_c W_L^(c kk) F_c[k], out of place (the R outputs of a position need the R inputs other threads are still reading).  R
// loads (the threads of the R bands share them in L2) and R twiddles per output instead of R per R outputs: a few times the cost of
// the instantiated kernels, for lengths they do not cover.
template <bool MSUB>
__global__ __launch_bounds__(256) void k_sub_combine_any(const FbGeom g, const cf* __restrict__ X, const uint32_t nseqs,
                                                         cf* __restrict__ Xout, const uint32_t mo, const uint32_t rm, const OddTw wr)
{
  const int logLs = g.logM + g.logR;
  const uint32_t R = wr.R, Ls = 1u << logLs, L = Ls * R;
  const uint32_t X3m = (1u << g.logX3) - 1, Mm = (1u << g.logM) - 1;
  const uint64_t n = (uint64_t)nseqs * L;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t o = (uint32_t)(i & (Ls - 1));
    const uint64_t rest = i >> logLs;
    const uint32_t q = (uint32_t)(rest % R);
    const uint64_t seq = rest / R;
    const cf* __restrict__ base = X + seq * (uint64_t)L + o;
    const uint32_t t = o >> g.logX3, m = t & Mm, sp = ((t >> g.logM) << g.logX3) | (o & X3m);
    const uint32_t k = (sp << g.logM) + m, kk = k + (q << logLs);
    cf acc = base[0];
    for (uint32_t c = 1; c < R; c++) {
      const cf v = cmul(base[(uint64_t)c << logLs], twiddle_odd_rt((uint64_t)c * kk, logLs, wr));
      acc.x += v.x; acc.y += v.y;
    }
    if constexpr (!MSUB) {
      Xout[seq * (uint64_t)L + ((uint64_t)q << logLs) + o] = acc;
    } else {
      const uint32_t Rr = R << g.logR, N = g.real_input ? L >> 1 : L;
      const bool up = kk > N;
      const uint32_t kq = up ? L - kk : kk;
      const uint32_t cc = kq / mo, mm = kq - cc * mo, mi = mm / rm, r = mm - mi * rm, s = cc * rm + r;
      uint32_t row, bin;
      if (kk == N && g.real_input) { row = Rr >> 1; bin = 0; }
      else if (!up) { row = s; bin = mi; }
      else if (mi) { row = Rr - 1 - s; bin = (1u << g.logM) - mi; }
      else { row = Rr - s; bin = 0; }
      Xout[seq * (uint64_t)L + (((((uint64_t)(row >> g.logX3) << g.logM) + bin) << g.logX3) | (row & X3m))] = acc;
    }
  }
}

template <int KIND, int EB>
static void sub_split_launch(hipStream_t stream, const SubSplit& p, uint8_t* out, uint32_t ncu)
{
  constexpr uint32_t N = 16 / EB;
  // the fast form needs 16-byte aligned stores: sub-blocks, polarisation rows and windows a multiple of 16 bytes apart
  const bool fast = (p.sub_stride % 16) == 0 && ((p.nper * EB) % 16) == 0 && ((p.wlen * EB) % 16) == 0 && ((uintptr_t)out % 16) == 0;
  uint64_t m_lo = 0;
  if (fast && p.wlen >= N) {
    m_lo = (p.wlen / N) * N;
    switch (p.R) {
      case 3: hipLaunchKernelGGL((k_sub_split<KIND, EB, 3>), dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
      case 5: hipLaunchKernelGGL((k_sub_split<KIND, EB, 5>), dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
      case 7: hipLaunchKernelGGL((k_sub_split<KIND, EB, 7>), dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
      case 9: hipLaunchKernelGGL((k_sub_split<KIND, EB, 9>), dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
      case 15: hipLaunchKernelGGL((k_sub_split<KIND, EB, 15>), dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
      default: m_lo = 0; break;
    }
  }
  if (m_lo < p.wlen) hipLaunchKernelGGL((k_sub_split_tail<KIND, EB>), dim3(8 * ncu), dim3(256), 0, stream, p, out, m_lo);
}
void fb_launch_sub_split(hipStream_t stream, const SubSplit& p, uint8_t* out, uint32_t ncu)
{
  const uint32_t es = p.npol * p.ndim;
  if (p.kind == 0) { if (p.ndim == 2) sub_split_launch<0, 8>(stream, p, out, ncu); else sub_split_launch<0, 4>(stream, p, out, ncu); }
  else if (p.kind == 2) sub_split_launch<2, 2>(stream, p, out, ncu);
  else if (es == 4) sub_split_launch<1, 4>(stream, p, out, ncu);
  else if (es == 2) sub_split_launch<1, 2>(stream, p, out, ncu);
  else sub_split_launch<1, 1>(stream, p, out, ncu);
}
cf* fb_launch_sub_combine(hipStream_t stream, const FbGeom& g, cf* X, uint32_t nseqs, uint32_t ncu, cf* Xalt, cf* Xout, uint32_t mo, uint32_t rm)
{
#define FB_SUBC(R)                                                                                                                          \
  case R:                                                                                                                                   \
    if (!Xout) hipLaunchKernelGGL((k_sub_combine<R, false>), dim3(8 * ncu), dim3(256), 0, stream, g, X, nseqs, X, 0u, 1u, make_odd_tw(R));   \
    else hipLaunchKernelGGL((k_sub_combine<R, true>), dim3(8 * ncu), dim3(256), 0, stream, g, X, nseqs, Xout, mo, rm, make_odd_tw(R));     \
    return Xout ? Xout : X;
  switch (g.nsub) { FB_SUBC(3) FB_SUBC(5) FB_SUBC(7) FB_SUBC(9) FB_SUBC(15) default: break; }
#undef FB_SUBC
  if (!Xout) hipLaunchKernelGGL((k_sub_combine_any<false>), dim3(8 * ncu), dim3(256), 0, stream, g, X, nseqs, Xalt, 0u, 1u, make_odd_tw(g.nsub));
  else hipLaunchKernelGGL((k_sub_combine_any<true>), dim3(8 * ncu), dim3(256), 0, stream, g, X, nseqs, Xout, mo, rm, make_odd_tw(g.nsub));
  return Xout ? Xout : Xalt;
}

}  // namespace dspsr_amd

FB_ST_READER(fwd_cols)
