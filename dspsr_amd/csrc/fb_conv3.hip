// dsp::Convolution with a response of 2^14 ... 2^21 points in THREE tile passes (round 5).
//
// Reference: Signal/General/Convolution.C:338-461 -- per (channel, polarisation, part): forward transform of M = n_fft complex samples,
// x response of the channel (Response.C:385-444), backward transform, samples [nfilt_pos, nfilt_pos + nsamp_step) kept.
//
// The filterbank object runs this shape (nchan_subband = 1) as four tile passes: two for the forward transform, two for the inverse
// one (fb_four_pass.hip), because the inverse of a FILTERBANK works on other sub-sequences of the spectrum than the forward
// transform's second pass produces.  For a convolution they are the same ones: with M = Fa * Fb,
//     n = Fb a + b,   k = c + Fa d,   t = t2 + Fb t1          (a, c, t1 < Fa;  b, d, t2 < Fb)
//     X[c + Fa d]  = sum_b W_Fb^{b d}  { W_M^{b c}  sum_a x[Fb a + b] W_Fa^{a c} }
//     y[t2 + Fb t1] = sum_c W_Fa^{-c t1} { W_M^{-c t2} sum_d X[c + Fa d] H[c + Fa d] W_Fb^{-d t2} }
// the sums over b (forward) and over d (inverse) both run along a "row" of fixed c.  So:
//   pass A  k_conv3_a : Fa-point forward transforms over a for TA adjacent b, x W_M^{b c}                     rows    -> S1
//   pass B  k_conv3_b : per row c (TB adjacent rows a tile): forward over b, x response, backward over d,
//                       x W_M^{-c t2} -- two transforms chained in one tile as in k_conv1 (fb_conv1.hip)       S1     -> S2
//   pass C  k_conv3_c : Fa-point backward transforms over c for TC adjacent t2, keep window, Detection         S2     -> rows
// Three reads and three writes of the data instead of four and four, and no spectrum in memory at all.
//
// A tile is 2^14 points; a column PAIR is the two polarisations of one column (one response factor and one twiddle for both; the
// thread that stores a sample holds both polarisations: Detection).  Layouts, per sequence seq = (channel, part) and both
// polarisations, in blocks of 2^14 elements -- one block = one pass-B tile, its input in S1 and its output in S2:
//   S1[seq][c / TB][b ][c % TB][pol]      pass A stages its tile and writes TA * TB * 2 elements (16 KB at M = 65536) per block
//   S2[seq][c / TB][t2][c % TB][pol]      pass B's own order; pass C gathers TC * TB * 2 elements per block
// Complex float32 rows with two polarisations (what Convolution::Engine::perform is handed behind a filterbank).
#include "fb_common.h"

namespace dspsr_amd {

struct Conv3Params {
  const float* in;                 // rows of the first channel: + chan * chan_stride + pol * pol_stride + part * in_step, (re, im) pairs
  uint64_t chan_stride, pol_stride, in_step;     // floats
  const cf* kern;                  // [nchan][M] of the first channel, every channel in pass-B order (fb_conv3_response_order), or null
  FbOut out;                       // kind 0 (none), 1 (complex rows), 2 (detected); chan0 = output row of the first channel
  cf* S1;
  cf* S2;
  uint32_t nchan, nparts;          // channels and parts of this launch group
  uint64_t part0;                  // first part of the group within the call
  uint32_t nfilt_pos, nkeep;
};

// the inverse inter-pass factor W_M^{-c t2}: applied by pass C to the elements it loads (one v_cos / v_sin pair each; the pass waits
// for memory) -- on the last-stage registers of pass B, which runs two transforms per tile with the response in registers, it cost 84
// bytes of scratch per lane
constexpr bool TW_IN_C = true;
// M = Fa * Fb.  Measured per 2^28 samples per polarisation in 128 channels (tools/conv_probe.py), (log2 Fa, log2 Fb), detected output:
// 2^14 (7, 7) / (6, 8) equal; 2^15 (7, 8) 7.0 ms, (6, 9) 7.9; 2^16 (8, 8) 7.4, (7, 9) 7.9; 2^17 (7, 10) 6.4, (8, 9) 6.6; 2^18 (7, 11) 5.7,
// (8, 10) 6.2, (9, 9) 8.7 (pass A spills); 2^19 (7, 12) 5.0, (8, 11) 5.1; 2^20 (8, 12) 5.4, (7, 13) 6.2; 2^21 (8, 13) 6.6 -- passes A / C want
// at least 32 columns per polarisation (256-byte runs of the input rows), pass B at least two rows per tile
// (4096-point rows in pass B -- three full radix-16 stages per transform -- spill 28 bytes per lane: 2^19 takes (8, 11); 2^20 has no other choice)
constexpr int conv3_la(int lm) { return lm >= 19 ? 8 : lm >= 17 ? 7 : lm / 2; }
constexpr int conv3_lb(int lm) { return lm - conv3_la(lm); }

// v[k] *= W_L^{+-nb (k pstride + p)} for BOTH halves of the pair (the two polarisations of column nb)
template <int R, bool CONJ> DEV void conv3_twiddle(cx2 (&v)[R], const uint32_t nb, const uint32_t p, const uint32_t pstride, const int logL)
{
  const uint32_t Lm = (1u << logL) - 1;
  const uint32_t a0 = (nb * p) & Lm, d0 = (nb * pstride) & Lm;            // (factors <= 2^13 and 2^8: every product below 2^21)
  constexpr int NP = R >= 16 ? 4 : R >= 8 ? 3 : R >= 4 ? 2 : R >= 2 ? 1 : 0;
  uint32_t j[1 + (NP ? NP : 1)];
  cf t[1 + (NP ? NP : 1)];
  j[0] = a0;
#pragma unroll
  for (int q = 0; q < (NP ? NP : 1); q++) j[1 + q] = (d0 << q) & Lm;
  twiddles_big(t, j, logL, (const cf*)nullptr, (const cf*)nullptr);        // logL <= 24: v_cos / v_sin, no table
  if (CONJ) {
#pragma unroll
    for (int q = 0; q < 1 + (NP ? NP : 1); q++) t[q].y = -t[q].y;
  }
  if constexpr (R > 1) {
    const cf w1 = t[1];
    const cf w2 = NP >= 2 ? t[1 + (NP >= 2 ? 1 : 0)] : w1, w4 = NP >= 3 ? t[1 + (NP >= 3 ? 2 : 0)] : w1, w8 = NP >= 4 ? t[1 + (NP >= 4 ? 3 : 0)] : w1;
    apply_powers<R>(v, w1, w2, w4, w8);
  }
#pragma unroll
  for (int k = 0; k < R; k++) v[k] = cmuls(v[k], t[0]);
}

// The kept window of one (channel, part, plane) as a buffer resource of nkeep elements: a sample outside [0, nkeep) -- its byte offset
// is negative (wraps) or beyond the last record -- is dropped by the hardware's range check.  No branch around a store, so the number
// of stores behind the next tile's prefetch is a constant and the wait for the prefetch at the top of a tile is vmcnt(stores), not
// vmcnt(0).  (Measured against the `continue` form: +-2 %, 16384 / 32768 points a little faster -- the pass waits for memory either way.)
DEV __amdgpu_buffer_rsrc_t win_rsrc(const float* p, const uint32_t bytes)
{
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
DEV void win_store(__amdgpu_buffer_rsrc_t r, const uint32_t off, const float4 v)
{
  __builtin_amdgcn_raw_buffer_store_b128((u4v){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, off, 0, 0);
}
DEV void win_store(__amdgpu_buffer_rsrc_t r, const uint32_t off, const float2 v)
{
  __builtin_amdgcn_raw_buffer_store_b64((u2v){__float_as_uint(v.x), __float_as_uint(v.y)}, r, off, 0, 0);
}
DEV void win_store(__amdgpu_buffer_rsrc_t r, const uint32_t off, const float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0); }

// passes A and C: workgroup -> first tile.  Consecutive workgroup ids go to consecutive XCDs (8 of them); the workgroups of XCD x take
// the tiles x * grid / 8 ... of every round, so neighbouring tiles run on the same XCD at about the same time
DEV uint32_t conv3_first_item()
{
  const uint32_t g = gridDim.x, b = blockIdx.x;
  return (g & 7) ? b : (b & 7) * (g >> 3) + (b >> 3);
}
// pass B: every workgroup takes one contiguous range of tiles
DEV bool conv3_range(const uint32_t total, uint32_t& item, uint32_t& item_end)
{
  item = (uint32_t)(((uint64_t)total * blockIdx.x) / gridDim.x);
  item_end = (uint32_t)(((uint64_t)total * (blockIdx.x + 1)) / gridDim.x);
  return item < item_end;
}

// ------------------------------------------------------------------------------------------------------------------------ pass A
template <int LM> struct Conv3AOut {
  cf* lds;
  uint32_t b0;                     // first column b of the tile
  template <int R> DEV void operator()(const uint32_t col, const uint32_t p, const uint32_t pstride, cx2 (&v)[R])
  {
    constexpr int LA = conv3_la(LM), LB = conv3_lb(LM), logTA = 13 - LA, logTB = 13 - LB;
    const uint32_t bl = col >> 1;
    conv3_twiddle<R, false>(v, b0 + bl, p, pstride, LM);
#pragma unroll
    for (int k = 0; k < R; k++) {
      const uint32_t c = k * pstride + p;
      const uint32_t l = ((((c >> logTB) << logTA) + bl) << logTB) | (c & ((1u << logTB) - 1));      // pair index of the staged image
      *(float4*)&lds[lds_pad(2 * l)] = make_float4(v[k].x[0], v[k].y[0], v[k].x[1], v[k].y[1]);
    }
  }
};

template <int LM>
__global__ __launch_bounds__(512) void k_conv3_a(const Conv3Params p, const cf* __restrict__ tw)
{
  constexpr int LA = conv3_la(LM), LB = conv3_lb(LM), logT = 14 - LA, logTA = 13 - LA, logTB = 13 - LB;
  typedef FftPlan<LA> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  constexpr uint32_t nt = 512, T = 1u << logT, TA = 1u << logTA;
  constexpr int logNbt = LB - logTA;                                       // tiles per sequence (= blocks per sequence: 2^(LM - 13))
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LA>(lds, ltw_off, tw, tid, nt);
  // tiles dealt round robin: the workgroups running at any moment then work on neighbouring tiles -- the pieces of the same rows
  // (256 bytes per row and tile at M = 65536) are requested at about the same time, and neighbouring tiles by workgroups of ONE
  // XCD (conv3_first_item): a part starts anywhere in a row, so the pieces straddle cache lines, and the line two neighbours share
  // is then fetched once into that XCD's L2
  const uint32_t item_end = (p.nchan * p.nparts) << logNbt, istep = gridDim.x;
  uint32_t item = conv3_first_item();
  if (item >= item_end) return;
  struct Pol2 { cf a, b; };
  auto fetch = [&](const uint32_t it, Pol2 (&raw)[NPAIR]) {
    const uint32_t seq = it >> logNbt, btile = it & ((1u << logNbt) - 1);
    const uint32_t chan = seq / p.nparts, part = seq - chan * p.nparts;
    const float* __restrict__ row = p.in + (uint64_t)chan * p.chan_stride + (p.part0 + part) * p.in_step + 2 * (btile << logTA);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        const uint32_t e = first_stage_elem<LA>(tid, logT, g2, i);
        const uint32_t j = (e & (T - 1)) >> 1, a = e >> logT;
        const float* __restrict__ q = row + 2 * ((a << LB) + j);
        Pol2 r;
        r.a = *(const float2*)q;
        r.b = *(const float2*)(q + p.pol_stride);
        raw[(g2 / 2) * P::R1 + i] = r;
      }
  };
  Pol2 raw[NPAIR];
  fetch(item, raw);
  for (;;) {
    asm volatile("" : "+v"(tid));
    const uint32_t seq = item >> logNbt, btile = item & ((1u << logNbt) - 1);
    cx2 x[NPAIR];
#pragma unroll
    for (int h = 0; h < NPAIR; h++) x[h] = make_cx2(raw[h].a, raw[h].b);
    const uint32_t next = item + istep;
    const bool more = next < item_end;
    fetch(more ? next : item, raw);

    Conv3AOut<LM> outA = {lds, btile << logTA};
    wgfft<LA, -1, true>(lds, ltw_off, tid, logT, x, outA);
    __syncthreads();
    // copy-out: the image is [c / TB][b - b0][c % TB][pol]; its piece c / TB goes to block c / TB of the sequence, behind the
    // pieces of the tiles in front of this one
    cf* __restrict__ Sq = p.S1 + ((uint64_t)seq << LM) * 2 + ((uint64_t)btile << (logTA + logTB + 1));
#pragma unroll
    for (int jj = 0; jj < PTS / 2; jj++) {
      const uint32_t l2 = 2 * (tid + jj * nt);
      const uint32_t chunk = l2 >> (logTA + logTB + 1), within = l2 & ((1u << (logTA + logTB + 1)) - 1);
      const float4 pr = *(const float4*)&lds[lds_pad(l2)];
      st_stream((float4*)&Sq[((uint64_t)chunk << 14) + within], pr);
    }
    if (!more) break;
    item = next;
  }
  (void)TA;
}

// ------------------------------------------------------------------------------------------------------------------------ pass B
template <int N> struct Conv3BFwd {
  cf* lds;
  const cf* kk;
  int logT;
  int h;
  template <int R> DEV void operator()(const uint32_t col, const uint32_t p, const uint32_t pstride, cx2 (&v)[R])
  {
#pragma unroll
    for (int k = 0; k < R; k++) {
      const cx2 q = cmuls(v[k], kk[h * R + k]);                            // Response::operate: both polarisations share the factor
      const uint32_t e = ((k * pstride + p) << logT) + col;                 // element (bin d, column) of the backward transform's tile
      *(float4*)&lds[lds_pad(e)] = make_float4(q.x[0], q.x[1], q.y[0], q.y[1]);
    }
  }
};

template <int LM>
__global__ __launch_bounds__(512) void k_conv3_b(const Conv3Params p, const cf* __restrict__ tw)
{
  constexpr int LB = conv3_lb(LM), logT = 14 - LB, logTB = 13 - LB;
  typedef FftPlan<LB> P;
  static_assert(P::NS >= 2, "k_conv3_b: at least two stages");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  constexpr uint32_t nt = 512, T = 1u << logT;
  constexpr int logNct = LM - 13;                                           // blocks per sequence
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LB>(lds, ltw_off, tw, tid, nt);
  // item = (chan, block, part), the part fastest: a workgroup walks the parts of one (channel, block) with its response in registers
  uint32_t item, item_end;
  if (!conv3_range((p.nchan * p.nparts) << logNct, item, item_end)) return;
  auto decode = [&](const uint32_t it, uint32_t& chan, uint32_t& blk, uint32_t& part) {
    const uint32_t cb = it / p.nparts;
    part = it - cb * p.nparts;
    chan = cb >> logNct;
    blk = cb & ((1u << logNct) - 1);
  };
  auto fetch = [&](const uint32_t it, float4 (&y)[NPAIR]) {
    uint32_t chan, blk, part;
    decode(it, chan, blk, part);
    const cf* __restrict__ B = p.S1 + ((((uint64_t)chan * p.nparts + part) << logNct) + blk) * (1u << 14);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++)
        y[(g2 / 2) * P::R1 + i] = *(const float4*)&B[first_stage_elem<LB>(tid, logT, g2, i)];
  };
  // forward last stage: radix RL, pair h = butterflies 2h, 2h + 1 of the thread: column pair u % T, bins d = k * (Fb / RL) + pp
  constexpr int LOGRL = P::REM ? P::REM : 4, RL = 1 << LOGRL, GL = PTS / RL, HL = GL / 2, logPL = LB - LOGRL;
  cf kk[NPAIR];
  uint32_t kk_cb = ~0u;
  auto load_response = [&](const uint32_t chan, const uint32_t blk) {
    if (!p.kern) {
#pragma unroll
      for (int q = 0; q < NPAIR; q++) kk[q] = make_float2(1.f, 0.f);
      return;
    }
    // the response lies in the order of the tiles: bin c + Fa d of a channel at [c / TB][d][c % TB] (fb_conv3_response_order), so the
    // lanes of a load -- neighbouring c, then neighbouring d -- read consecutive elements
    const cf* __restrict__ kc = p.kern + ((uint64_t)chan << LM) + ((uint64_t)blk << (LB + logTB));
#pragma unroll
    for (int h = 0; h < HL; h++) {
      const uint32_t u = GL * tid + 2 * h, cl = (u & (T - 1)) >> 1, pp = (u >> logT) & ((1u << logPL) - 1);
#pragma unroll
      for (int k = 0; k < RL; k++) kk[h * RL + k] = kc[(((((uint32_t)k << logPL) + pp)) << logTB) + cl];
    }
  };
  float4 y[NPAIR];
  fetch(item, y);
  for (;;) {
    asm volatile("" : "+v"(tid));
    uint32_t chan, blk, part;
    decode(item, chan, blk, part);
    if ((item / p.nparts) != kk_cb) { load_response(chan, blk); kk_cb = item / p.nparts; }
    cx2 x[NPAIR];
#pragma unroll
    for (int h = 0; h < NPAIR; h++) x[h] = make_cx2(make_float2(y[h].x, y[h].y), make_float2(y[h].z, y[h].w));
    const uint32_t next = item + 1;
    const bool more = next < item_end;
    fetch(more ? next : item, y);

    Conv3BFwd<NPAIR> fwd = {lds, kk, logT, 0};
    wgfft<LB, -1, true>(lds, ltw_off, tid, logT, x, fwd);
    __syncthreads();                          // spectrum x response in the exchange buffer, element (bin d, column)

    cf* __restrict__ O = p.S2 + ((((uint64_t)chan * p.nparts + part) << logNct) + blk) * (1u << 14);
    const uint32_t c0 = blk << logTB;
    auto store = [&](const uint32_t col, const uint32_t pos, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if constexpr (!TW_IN_C) conv3_twiddle<R, true>(v, c0 + (col >> 1), pos, pstride, LM);
#pragma unroll
      for (int k = 0; k < R; k++)
        st_stream((float4*)&O[((k * pstride + pos) << logT) + col], make_float4(v[k].x[0], v[k].y[0], v[k].x[1], v[k].y[1]));
    };
    wgfft<LB, +1, false, true>(lds, ltw_off, tid, logT, x, store);
    if (!more) break;
    item = next;
  }
}

// ------------------------------------------------------------------------------------------------------------------------ pass C
// OF: output form -- 0 complex rows (two 8-byte stores per sample; also "no output": resources of zero records), 1 detected ndim 4 (one
// 16-byte store), 2 detected ndim 2 (two 8-byte stores), 3 detected ndim 1 (four 4-byte stores): the number of stores per tile is a
// constant of the instantiation
template <int LM, int OF>
__global__ __launch_bounds__(512) void k_conv3_c(const Conv3Params p, const cf* __restrict__ tw)
{
  constexpr int LA = conv3_la(LM), LB = conv3_lb(LM), logT = 14 - LA, logTC = 13 - LA, logTB = 13 - LB;
  typedef FftPlan<LA> P;
  static_assert(P::NS >= 2, "k_conv3_c: at least two stages");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  constexpr uint32_t nt = 512;
  constexpr int logNtt = LB - logTC;                                        // tiles per sequence (2^(LM - 13))
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LA>(lds, ltw_off, tw, tid, nt);
  // round robin with the neighbours on one XCD, as pass A: the runs of output samples two neighbouring tiles write (512 bytes each at
  // any 16-byte offset) meet in that XCD's L2 and leave it as whole lines
  const uint32_t item_end = (p.nchan * p.nparts) << logNtt, istep = gridDim.x;
  uint32_t item = conv3_first_item();
  if (item >= item_end) return;
  // the tile in memory order: piece `chunk` = [t2 - t2_0][c % TB][pol] of block `chunk` of the sequence
  auto fetch = [&](const uint32_t it, float4 (&y)[NPAIR]) {
    const uint32_t seq = it >> logNtt, ttile = it & ((1u << logNtt) - 1);
    const cf* __restrict__ Sq = p.S2 + ((uint64_t)seq << LM) * 2 + ((uint64_t)ttile << (logTC + logTB + 1));
#pragma unroll
    for (int jj = 0; jj < PTS / 2; jj++) {
      const uint32_t l2 = 2 * (tid + jj * nt);
      const uint32_t chunk = l2 >> (logTC + logTB + 1), within = l2 & ((1u << (logTC + logTB + 1)) - 1);
      y[jj] = *(const float4*)&Sq[((uint64_t)chunk << 14) + within];
    }
  };
  float4 y[NPAIR];
  fetch(item, y);
  for (;;) {
    asm volatile("" : "+v"(tid));
    const uint32_t seq = item >> logNtt, ttile = item & ((1u << logNtt) - 1);
    const uint32_t chan = seq / p.nparts, part_l = seq - chan * p.nparts;
    const uint32_t t20 = ttile << logTC;
    __syncthreads();                          // the previous tile's last exchange has been read by every wave
#pragma unroll
    for (int jj = 0; jj < PTS / 2; jj++) {
      const uint32_t l2 = 2 * (tid + jj * nt);
      const uint32_t chunk = l2 >> (logTC + logTB + 1), within = l2 & ((1u << (logTC + logTB + 1)) - 1);
      const uint32_t c = (chunk << logTB) + ((within >> 1) & ((1u << logTB) - 1)), tl = within >> (logTB + 1);
      float4 q = y[jj];
      if constexpr (TW_IN_C) {                 // x W_M^{-c t2}: one factor for both polarisations, v_cos / v_sin in revolutions
        const float ang = (float)((c * (t20 + tl)) & ((1u << LM) - 1)) * __uint_as_float((uint32_t)(127 - LM) << 23);
        const float wr = __builtin_amdgcn_cosf(ang), wi = __builtin_amdgcn_sinf(ang);
        q = make_float4(q.x * wr - q.y * wi, q.x * wi + q.y * wr, q.z * wr - q.w * wi, q.z * wi + q.w * wr);
      }
      *(float4*)&lds[lds_pad((c << logT) + 2 * tl)] = make_float4(q.x, q.z, q.y, q.w);     // split form
    }
    const uint32_t next = item + istep;
    const bool more = next < item_end;
    fetch(more ? next : item, y);
    __syncthreads();

    const FbOut& out = p.out;
    const uint64_t part = p.part0 + part_l;
    const float* row = out.base + (uint64_t)(out.chan0 + chan) * out.chan_stride;
    constexpr uint32_t ebytes = OF == 1 ? 16u : OF == 3 ? 4u : 8u;            // bytes per sample and plane
    const uint32_t wbytes = out.kind == 0 ? 0u : p.nkeep * ebytes;
    const float* w0 = OF == 0 ? row + part * out.part_step : row + part * p.nkeep * (ebytes / 4);
    const __amdgpu_buffer_rsrc_t r0 = win_rsrc(w0, wbytes), r1 = win_rsrc(w0 + out.pol_stride, OF == 1 ? 0u : wbytes),
                                 r2 = win_rsrc(w0 + 2 * out.pol_stride, OF == 3 ? wbytes : 0u),
                                 r3 = win_rsrc(w0 + 3 * out.pol_stride, OF == 3 ? wbytes : 0u);
    cx2 x[NPAIR];
    auto store = [&](const uint32_t col, const uint32_t pos, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      const uint32_t o0 = (t20 + (col >> 1) + (pos << LB) - p.nfilt_pos) * ebytes;       // (wraps below the window: out of range)
      const uint32_t ostep = (pstride << LB) * ebytes;
#pragma unroll
      for (int k = 0; k < R; k++) {
        const uint32_t o = o0 + k * ostep;
        if constexpr (OF == 0) {
          win_store(r0, o, cx2_lo(v[k]));
          win_store(r1, o, cx2_hi(v[k]));
        } else {
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          if constexpr (OF == 1) win_store(r0, o, make_float4(r[0], r[1], r[2], r[3]));
          else if constexpr (OF == 2) {
            win_store(r0, o, make_float2(r[0], r[1]));
            win_store(r1, o, make_float2(r[2], r[3]));
          } else {
            win_store(r0, o, r[0]);
            win_store(r1, o, r[1]);
            win_store(r2, o, r[2]);
            win_store(r3, o, r[3]);
          }
        }
      }
    };
    wgfft<LA, +1, false, true>(lds, ltw_off, tid, logT, x, store);
    if (!more) break;
    item = next;
  }
}

// ------------------------------------------------------------------------------------------------------------------------ host
typedef void (*kconv3_t)(Conv3Params, const cf*);
struct Conv3Kernels { kconv3_t a, b, c[4]; };
template <int... I> static Conv3Kernels pick_conv3(int logm, iseq<I...>)
{
  static const Conv3Kernels t[] = {{k_conv3_a<I + CONV3_MIN_LOGM>, k_conv3_b<I + CONV3_MIN_LOGM>,
                                    {k_conv3_c<I + CONV3_MIN_LOGM, 0>, k_conv3_c<I + CONV3_MIN_LOGM, 1>, k_conv3_c<I + CONV3_MIN_LOGM, 2>,
                                     k_conv3_c<I + CONV3_MIN_LOGM, 3>}}...};
  if (logm >= CONV3_MIN_LOGM && logm < CONV3_MIN_LOGM + (int)sizeof...(I)) return t[logm - CONV3_MIN_LOGM];
  return Conv3Kernels{nullptr, nullptr, {nullptr, nullptr, nullptr, nullptr}};
}
static Conv3Kernels conv3_kernels(int logm) { return pick_conv3(logm, mkseq<CONV3_MAX_LOGM - CONV3_MIN_LOGM + 1>::type()); }

void fb_conv3_response_order(int logM, const cf* natural, cf* ordered)
{
  const int la = conv3_la(logM), lb = conv3_lb(logM), ltb = 13 - lb;
  const uint32_t Fa = 1u << la, Fb = 1u << lb, TB = 1u << ltb;
  for (uint32_t blk = 0; blk < Fa / TB; blk++)
    for (uint32_t d = 0; d < Fb; d++)
      for (uint32_t cl = 0; cl < TB; cl++)
        ordered[(((uint64_t)blk << lb) + d) * TB + cl] = natural[(uint64_t)d * Fa + blk * TB + cl];
}

static size_t conv3_lds(int logF) { return lds_total_words_host(1u << 14, logF) * sizeof(cf); }

int fb_conv3_check(int logM)
{
  const Conv3Kernels k = conv3_kernels(logM);
  if (!k.a) return DSPSR_AMD_EINVAL;
  const int la = conv3_la(logM), lb = conv3_lb(logM);
  if (dspsr_amd_allow_lds((const void*)k.a, conv3_lds(la)) != hipSuccess || dspsr_amd_allow_lds((const void*)k.b, conv3_lds(lb)) != hipSuccess)
    return DSPSR_AMD_EHIP;
  for (int f = 0; f < 4; f++)
    if (dspsr_amd_allow_lds((const void*)k.c[f], conv3_lds(la)) != hipSuccess) return DSPSR_AMD_EHIP;
  return DSPSR_AMD_OK;
}

// one launch group: channels [0, nchan) behind `in` / `kern` / out.chan0, parts [part0, part0 + nparts); S1 and S2 hold
// nchan * nparts * 2 * M elements each
int fb_conv3_launch(dspsr_amd_ctx* ctx, int logM, const float* in, uint64_t chan_stride, uint64_t pol_stride, uint64_t in_step,
                    const cf* kern, const FbOut& out, uint32_t nchan, uint32_t nfilt_pos, uint32_t nkeep, uint64_t part0, uint32_t nparts,
                    cf* S1, cf* S2)
{
  const Conv3Kernels k = conv3_kernels(logM);
  if (!k.a) return DSPSR_AMD_EINVAL;
  const uint64_t total = ((uint64_t)nchan * nparts) << (logM - 13);
  if (total == 0) return DSPSR_AMD_OK;
  if (total >= (1ull << 31)) return DSPSR_AMD_EINVAL;
  Conv3Params p = {};
  p.in = in; p.chan_stride = chan_stride; p.pol_stride = pol_stride; p.in_step = in_step;
  p.kern = kern; p.out = out; p.S1 = S1; p.S2 = S2; p.nchan = nchan; p.nparts = nparts; p.part0 = part0;
  p.nfilt_pos = nfilt_pos; p.nkeep = nkeep;
  const int la = conv3_la(logM), lb = conv3_lb(logM);
  const uint32_t grid = (uint32_t)(total < ctx->ncu ? total : ctx->ncu);
  hipLaunchKernelGGL(k.a, dim3(grid), dim3(512), conv3_lds(la), ctx->stream, p, ctx->tw);
  hipLaunchKernelGGL(k.b, dim3(grid), dim3(512), conv3_lds(lb), ctx->stream, p, ctx->tw);
  const int of = out.kind == 2 ? (out.ndim == 4 ? 1 : out.ndim == 2 ? 2 : 3) : 0;
  hipLaunchKernelGGL(k.c[of], dim3(grid), dim3(512), conv3_lds(la), ctx->stream, p, ctx->tw);
  return hipGetLastError() == hipSuccess ? DSPSR_AMD_OK : DSPSR_AMD_EHIP;
}

}  // namespace dspsr_amd
