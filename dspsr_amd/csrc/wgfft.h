// In-workgroup batched FFT for gfx950 (wave64): F-point FFTs over T interleaved columns
// (T >= 2), F*T = 32*blockDim.x points, every thread owning PTS = 32 points per stage.
//
// Stockham auto-sort, radix-16 stages (4x4 butterflies held in registers) plus one
// radix-2/4/8 remainder stage.  The FIRST stage takes its inputs from a register array the
// caller filled (prefetched from global memory while the previous tile was computed) and the
// LAST stage hands its outputs to a caller functor (registers -> global memory); only the
// exchanges BETWEEN stages go through LDS, element (pos, col) at word lds_pad(pos*T + col).
// Stage with radix R, G = 32/R butterflies per thread, P = product of earlier radices, Q = F/(P*R):
//   butterfly u = G*tid + g = col + T*(p + P*s)   (p < P, s < Q)
//   reads   u + i*(F/R)*T                  i < R
//   twiddle W_{R*Q}^{k*s}                  (4 table loads + multiplication ladder per butterfly)
//   writes  (s*P*R + k*P + p)*T + col      k < R
// Butterflies 2j and 2j+1 of a thread are columns (col, col+1) of the same position, so every
// LDS access moves a 16-byte pair (ds_read_b128 / ds_write_b128).
// (index math validated against numpy in tests/test_wgfft_model.py)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dspsr_amd {

typedef float2 cf;

#define DEV __device__ __forceinline__

DEV cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// 4 words of padding per 64 keep the strided stage writes spread over the LDS banks
// and preserve 16-byte alignment of even word indices
DEV uint32_t lds_pad(uint32_t e) { return e + ((e >> 6) << 2); }
inline uint32_t lds_words_host(uint32_t points) { return points + ((points >> 6) << 2) + 8; }   // exchange buffer
// total dynamic LDS words of a kernel: exchange buffer followed by the stage twiddle tables
inline uint32_t ltw_entries_host(int logF)
{
  const int nq = logF / 4, rem = logF % 4, ntw = nq - (rem ? 0 : 1);
  uint32_t n = 0;
  for (int st = 0; st < ntw; st++) n += 4u << (logF - 4 * (st + 1));
  return n;
}
// (+ 16: a workgroup that works as two half-tile groups has two exchange buffers, each with its own 8 words of slack)
inline uint32_t lds_total_words_host(uint32_t points, int logF) { return lds_words_host(points) + ltw_entries_host(logF) + 8 + 16; }


constexpr int PTS = 32;      // points per thread
constexpr int LOG_PTS = 5;
constexpr int NPAIR = PTS / 2;

// A thread's two butterflies (columns col, col+1 of the same position) are carried together in SPLIT
// form: x = (re_col, re_col+1), y = (im_col, im_col+1).  Every complex operation is then a plain
// element-wise operation on 2-vectors (v_pk_add/mul/fma_f32) with no operand shuffling, and the pair
// is also the 16-byte unit moved through LDS.
typedef float v2f __attribute__((ext_vector_type(2)));
struct cx2 { v2f x, y; };

DEV cx2 make_cx2(cf a, cf b) { cx2 r; r.x = (v2f){a.x, b.x}; r.y = (v2f){a.y, b.y}; return r; }
DEV cf cx2_lo(cx2 a) { return make_float2(a.x[0], a.y[0]); }
DEV cf cx2_hi(cx2 a) { return make_float2(a.x[1], a.y[1]); }
DEV cx2 cadd(cx2 a, cx2 b) { cx2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
DEV cx2 csub(cx2 a, cx2 b) { cx2 r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
DEV cx2 cmul(cx2 a, cx2 b) { cx2 r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; }
// multiply both by the same scalar complex w
DEV cx2 cmuls(cx2 a, cf w) { cx2 r; r.x = a.x * w.x - a.y * w.y; r.y = a.x * w.y + a.y * w.x; return r; }
// multiply by SIGN*i  (forward SIGN=-1: -i ; inverse SIGN=+1: +i)
template <int SIGN> DEV cx2 mul_si(cx2 a) { cx2 r; if (SIGN < 0) { r.x = a.y; r.y = -a.x; } else { r.x = -a.y; r.y = a.x; } return r; }

// cos/sin(2*pi*k/16)
#define C16_1 0.92387953251128674f
#define S16_1 0.38268343236508977f
#define C16_2 0.70710678118654752f

template <int SIGN> DEV void fft2(cx2& a, cx2& b) { cx2 t = a; a = cadd(t, b); b = csub(t, b); }

template <int SIGN> DEV void fft4(cx2& v0, cx2& v1, cx2& v2, cx2& v3)
{
  cx2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = mul_si<SIGN>(csub(v1, v3));
  v0 = cadd(t0, t2); v2 = csub(t0, t2);
  v1 = cadd(t1, t3); v3 = csub(t1, t3);
}

// multiply by W_8^1 = exp(SIGN*i*pi/4) and W_8^3
template <int SIGN> DEV cx2 mul_w8_1(cx2 a)
{ cx2 r; if (SIGN < 0) { r.x = C16_2 * (a.x + a.y); r.y = C16_2 * (a.y - a.x); } else { r.x = C16_2 * (a.x - a.y); r.y = C16_2 * (a.x + a.y); } return r; }
template <int SIGN> DEV cx2 mul_w8_3(cx2 a)
{ cx2 r; if (SIGN < 0) { r.x = C16_2 * (a.y - a.x); r.y = -C16_2 * (a.x + a.y); } else { r.x = -C16_2 * (a.x + a.y); r.y = C16_2 * (a.x - a.y); } return r; }
template <int SIGN> DEV cx2 mul_w(cx2 a, float c, float s) { return cmuls(a, make_float2(c, SIGN < 0 ? -s : s)); }

template <int SIGN> DEV void fft8(cx2 (&v)[8])
{
  // DIT: even/odd 4-point transforms then combine
  fft4<SIGN>(v[0], v[2], v[4], v[6]);
  fft4<SIGN>(v[1], v[3], v[5], v[7]);
  cx2 o1 = mul_w8_1<SIGN>(v[3]), o2 = mul_si<SIGN>(v[5]), o3 = mul_w8_3<SIGN>(v[7]);
  cx2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

template <int SIGN> DEV void fft16(cx2 (&v)[16])
{
  // n = 4*n1 + n2 : 4-point over n1 for each n2, twiddle W_16^{n2*k1}, 4-point over n2 ; k = k1 + 4*k2
  fft4<SIGN>(v[0], v[4], v[8], v[12]);
  fft4<SIGN>(v[1], v[5], v[9], v[13]);
  fft4<SIGN>(v[2], v[6], v[10], v[14]);
  fft4<SIGN>(v[3], v[7], v[11], v[15]);
  // y[k1][n2] sits in v[4*k1 + n2]
  v[5] = mul_w<SIGN>(v[5], C16_1, S16_1);     // W^1
  v[6] = mul_w8_1<SIGN>(v[6]);                // W^2
  v[7] = mul_w<SIGN>(v[7], S16_1, C16_1);     // W^3
  v[9] = mul_w8_1<SIGN>(v[9]);                // W^2
  v[10] = mul_si<SIGN>(v[10]);                // W^4
  v[11] = mul_w8_3<SIGN>(v[11]);              // W^6
  v[13] = mul_w<SIGN>(v[13], S16_1, C16_1);   // W^3
  v[14] = mul_w8_3<SIGN>(v[14]);              // W^6
  v[15] = mul_w<SIGN>(v[15], -C16_1, -S16_1); // W^9
  fft4<SIGN>(v[0], v[1], v[2], v[3]);
  fft4<SIGN>(v[4], v[5], v[6], v[7]);
  fft4<SIGN>(v[8], v[9], v[10], v[11]);
  fft4<SIGN>(v[12], v[13], v[14], v[15]);
  // X[k1 + 4*k2] sits in v[4*k1 + k2] : transpose 4x4 into natural order
  cx2 t;
  t = v[1]; v[1] = v[4]; v[4] = t;
  t = v[2]; v[2] = v[8]; v[8] = t;
  t = v[3]; v[3] = v[12]; v[12] = t;
  t = v[6]; v[6] = v[9]; v[9] = t;
  t = v[7]; v[7] = v[13]; v[13] = t;
  t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int R, int SIGN> DEV void fftR(cx2 (&v)[R])
{
  if constexpr (R == 2) fft2<SIGN>(v[0], v[1]);
  else if constexpr (R == 4) fft4<SIGN>(v[0], v[1], v[2], v[3]);
  else if constexpr (R == 8) fft8<SIGN>(v);
  else if constexpr (R == 16) fft16<SIGN>(v);
}

// v[k] *= w1^k for k = 1..R-1 given the exact powers w1, w2, w4, w8 (those that exist for R), the same
// scalar twiddle for both columns of the pair; the remaining powers are products of at most three
// exact factors.
template <int R> DEV void apply_powers(cx2 (&v)[R], const cf w1, const cf w2, const cf w4, const cf w8)
{
  if constexpr (R >= 2) v[1] = cmuls(v[1], w1);
  if constexpr (R >= 4) {
    const cf w3 = cmul(w2, w1);
    v[2] = cmuls(v[2], w2);
    v[3] = cmuls(v[3], w3);
    if constexpr (R >= 8) {
      const cf w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
      v[4] = cmuls(v[4], w4); v[5] = cmuls(v[5], w5); v[6] = cmuls(v[6], w6); v[7] = cmuls(v[7], w7);
      if constexpr (R >= 16) {
        v[8] = cmuls(v[8], w8);
        v[9] = cmuls(v[9], cmul(w8, w1));
        v[10] = cmuls(v[10], cmul(w8, w2));
        v[11] = cmuls(v[11], cmul(w8, w3));
        v[12] = cmuls(v[12], cmul(w8, w4));
        v[13] = cmuls(v[13], cmul(w8, w5));
        v[14] = cmuls(v[14], cmul(w8, w6));
        v[15] = cmuls(v[15], cmul(w8, w7));
      }
    }
  }
}
// same with a different twiddle per column (packed powers)
template <int R> DEV void apply_powers2(cx2 (&v)[R], const cx2 w1, const cx2 w2, const cx2 w4, const cx2 w8)
{
  if constexpr (R >= 2) v[1] = cmul(v[1], w1);
  if constexpr (R >= 4) {
    const cx2 w3 = cmul(w2, w1);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    if constexpr (R >= 8) {
      const cx2 w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
      v[4] = cmul(v[4], w4); v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
      if constexpr (R >= 16) {
        v[8] = cmul(v[8], w8);
        v[9] = cmul(v[9], cmul(w8, w1));
        v[10] = cmul(v[10], cmul(w8, w2));
        v[11] = cmul(v[11], cmul(w8, w3));
        v[12] = cmul(v[12], cmul(w8, w4));
        v[13] = cmul(v[13], cmul(w8, w5));
        v[14] = cmul(v[14], cmul(w8, w6));
        v[15] = cmul(v[15], cmul(w8, w7));
      }
    }
  }
}

// twiddle table: tw[j] = exp(-2*pi*i*j/TWN), j < TWN
constexpr int LOG_TWN = 14;
constexpr int TWN = 1 << LOG_TWN;

// Stage twiddles live in LDS behind the exchange buffer, one small table per twiddled (radix-16) stage:
//   table[st][kk][s] = W_{16 Q}^{(1<<kk) s},  kk < 4, s < Q = F / 16^(st+1)
// so the transform phase issues no global loads (the in-order vmcnt would otherwise force the next
// tile's prefetch and the previous tile's stores to drain), and lanes read consecutive or identical
// entries (conflict-free).  4*(F/16 + F/256 + ...) < 0.27 F entries.
constexpr int LTW_MAX_ENTRIES = 2304;       // enough for F = 8192
template <int LOGF> DEV void ltw_fill(cf* lds, const uint32_t ltw_off, const cf* __restrict__ tw, const uint32_t tid,
                                      const uint32_t nt)
{
  constexpr int NQ = LOGF / 4, REM = LOGF % 4;
  constexpr int NTW = NQ - (REM ? 0 : 1);   // twiddled stages: every radix-16 stage except a final one
  uint32_t off = 0;
#pragma unroll
  for (int st = 0; st < NTW; st++) {
    const int logQ = LOGF - 4 * (st + 1);
    const uint32_t Q = 1u << logQ;
    for (uint32_t i = tid; i < 4 * Q; i += nt) {
      const uint32_t kk = i >> logQ, sidx = i & (Q - 1);
      lds[ltw_off + off + i] = tw[((sidx << kk) << (LOG_TWN - 4 - logQ)) & (TWN - 1)];
    }
    off += 4 * Q;
  }
  __syncthreads();
}

template <int LOGF> constexpr uint32_t ltw_entries_dev()
{
  constexpr int NQ = LOGF / 4, REM = LOGF % 4, NTW = NQ - (REM ? 0 : 1);
  uint32_t n = 0;
  for (int st = 0; st < NTW; st++) n += 4u << (LOGF - 4 * (st + 1));
  return n;
}

// radix plan of an F = 2^LOGF transform
template <int LOGF> struct FftPlan {
  static constexpr int NQ = LOGF / 4, REM = LOGF % 4;
  static constexpr int NS = NQ + (REM ? 1 : 0);
  static constexpr int LOGR1 = NS == 0 ? 0 : (NQ ? 4 : REM);   // first-stage radix
  static constexpr int R1 = 1 << LOGR1;
  static constexpr int G1 = PTS / R1;
};

// logical element index (pos*T + col) of the FIRST element of pair x[(g/2)*R1 + i] (g even) of the
// first-stage register array; the second element of the pair is the next column (index + 1)
template <int LOGF> DEV uint32_t first_stage_elem(uint32_t tid, int logT, int g, int i)
{
  typedef FftPlan<LOGF> P;
  return P::G1 * tid + g + ((uint32_t)i << (LOGF - P::LOGR1 + logT));
}

// Out: void operator()(uint32_t col, uint32_t p, uint32_t pstride, cx2 (&v)[R])
//      v[k] holds output position k*pstride + p of columns col (even, .x[0]/.y[0]) and col + 1
struct NoMid { DEV void operator()(int) const {} };
// an Out that keeps the last-stage outputs in registers is told which of the thread's butterfly pairs a call is (`h`, a
// compile-time constant after unrolling), so that its register array is indexed statically
template <class O> DEV auto out_set_h(O& o, const int h, int) -> decltype((void)(o.h = h)) { o.h = h; }
template <class O> DEV void out_set_h(O&, const int, long) {}

// Mid: called once per stage between the butterflies and the exchange (`mid(stage + 1)`; the driver calls
// `mid(0)` before the first stage).  The passes use it to issue the global loads of their NEXT tile in
// NS + 1 small groups spread over the transform instead of one burst that blocks the wave while the
// memory pipeline accepts it.
// MIRROR (last stage only, real-input transforms): the thread's butterfly pair is not (column c, c + 1) of one position but the
// positions (p, P - p) of ONE column -- p = 0 goes with P/2 --, read as two 8-byte elements.  Its outputs are then bins k*P + p in
// the low halves and k*P + (P - p) in the high halves of v[k]: the Hermitian mirror C - (k*P + p) = (R-1-k)*P + (P - p) of every
// bin lies in the SAME thread (high half of v[R-1-k]), so the real-transform post-processing (X[k] from Z[k], Z[C-k]) needs no
// further exchange.  Pair q = H*tid + h is column q % T, position pair q / T; `out(v)` is called ONCE with the thread's whole
// image cx2 v[H][R].
// ILV (the stage in FRONT of a MIRROR stage): its exchange writes the pair as (re, im) of column c, (re, im) of column c + 1 --
// whole 8-byte elements per column -- instead of the split form (re c, re c+1, im c, im c+1) every other stage reads back.
template <int LOGR, int SIGN, bool FIRST, bool LAST, bool STAGED, bool MIRROR = false, bool ILV = false, class Out, class Mid>
DEV void wgfft_stage(cf* lds, const uint32_t ltw_off, const uint32_t tid, const int logT, const int logF,
                     const int logP, cx2 (&x)[NPAIR], Out& out, Mid& mid, const int phase)
{
  static_assert(!MIRROR || (LAST && !FIRST && !STAGED), "MIRROR: a last stage fed from the exchange buffer, outputs kept in registers");
  constexpr int R = 1 << LOGR;
  constexpr int G = PTS / R;          // butterflies per thread
  constexpr int H = G / 2;            // pairs of butterflies
  const int logQ = logF - logP - LOGR;
  const uint32_t stride = 1u << (logF - LOGR + logT);
  cx2 v[H][R];
  // Padded addresses without per-element arithmetic: lds_pad(e0 + c) == lds_pad(e0) + c + ((c >> 6) << 2) whenever
  // (e0 & 63) + (c & 63) < 64.  For the reads c = i*stride with stride a multiple of 64; for the writes
  // c = k*step, step = P*T, and e0 & 63 < step because R*P*T is a multiple of 64 (uniform conditions, folded at
  // compile time for the full-size tiles).
  const bool raff = (stride & 63) == 0;
#pragma unroll
  for (int h = 0; h < H; h++) {
    const uint32_t u = G * tid + 2 * h;
    const uint32_t rbase = lds_pad(u);
#pragma unroll
    for (int i = 0; i < R; i++) {
      if constexpr (FIRST) {
        v[h][i] = x[h * R + i];
      } else if constexpr (MIRROR) {
        const uint32_t q = H * tid + h, c = q & ((1u << logT) - 1), j = q >> logT;
        const uint32_t pb = j ? (1u << logP) - j : (1u << (logP - 1));
        const uint32_t ea = (j << logT) + c, eb = (pb << logT) + c, cc = i * stride;     // (stride = P*T)
        const cf a = lds[raff ? lds_pad(ea) + cc + ((cc >> 6) << 2) : lds_pad(ea + cc)];
        const cf b = lds[raff ? lds_pad(eb) + cc + ((cc >> 6) << 2) : lds_pad(eb + cc)];
        v[h][i].x = (v2f){a.x, b.x};
        v[h][i].y = (v2f){a.y, b.y};
      } else {
        const uint32_t c = i * stride;
        const float4 pr = *(const float4*)&lds[raff ? rbase + c + ((c >> 6) << 2) : lds_pad(u + c)];
        v[h][i].x = (v2f){pr.x, pr.y};
        v[h][i].y = (v2f){pr.z, pr.w};
      }
    }
  }
#pragma unroll
  for (int h = 0; h < H; h++) {
    fftR<R, SIGN>(v[h]);
    if (R > 1 && logQ > 0) {
      // only radix-16 stages carry twiddles (a remainder stage is always last); both columns of a pair
      // share s, hence the twiddle
      const uint32_t u = G * tid + 2 * h;
      const uint32_t s = u >> (logT + logP);
      const uint32_t tb = ltw_off + s;
      cf w1 = lds[tb], w2 = lds[tb + (1u << logQ)], w4 = lds[tb + (2u << logQ)], w8 = lds[tb + (3u << logQ)];
      if (SIGN > 0) { w1.y = -w1.y; w2.y = -w2.y; w4.y = -w4.y; w8.y = -w8.y; }
      apply_powers<R>(v[h], w1, w2, w4, w8);
    }
  }
  mid(phase);
  // every read of the in-place exchange buffer (this or the previous tile) is done; a STAGED last stage
  // re-uses the buffer to reorder its outputs, so it needs the same guarantee (also when it is the only stage:
  // the previous tile's staged image may still be being read)
  if (!LAST || STAGED) __syncthreads();
#pragma unroll
  for (int h = 0; h < H; h++) {
    const uint32_t u = G * tid + 2 * h;
    const uint32_t col = u & ((1u << logT) - 1);
    const uint32_t rest = u >> logT;
    const uint32_t p = rest & ((1u << logP) - 1);
    const uint32_t s = rest >> logP;
    if constexpr (LAST && MIRROR) {
      if (h == 0) out(v);                // the whole register image at once: pair h is column (H tid + h) % T, positions ((H tid + h) / T, P - that)
    } else if constexpr (LAST) {
      out_set_h(out, h, 0);
      out(col, p, 1u << logP, v[h]);     // Q == 1, s == 0
    } else {
      const uint32_t e0 = ((((s << (logP + LOGR)) + p) << logT) | col), step = 1u << (logP + logT);
      const bool waff = LOGR + logP + logT >= 6 && logP + logT <= 6;    // step divides 64 or is a multiple of it
      const bool waff2 = waff || (step & 63) == 0;
      const uint32_t wbase = lds_pad(e0);
#pragma unroll
      for (int k = 0; k < R; k++) {
        const uint32_t c = k * step;
        *(float4*)&lds[waff2 ? wbase + c + ((c >> 6) << 2) : lds_pad(e0 + c)] =
            ILV ? make_float4(v[h][k].x[0], v[h][k].y[0], v[h][k].x[1], v[h][k].y[1])
                : make_float4(v[h][k].x[0], v[h][k].x[1], v[h][k].y[0], v[h][k].y[1]);
      }
    }
  }
  if (!LAST) __syncthreads();
}

// Runs the whole F-point transform of one tile: x feeds the first stage, `out` receives the last.
// May be called repeatedly (persistent workgroup): the barrier in front of the first LDS write
// also separates it from the previous tile's last-stage LDS reads.
// STAGED: `out` writes into the exchange buffer (the caller copies it out after a barrier).
// FROM_LDS: the caller has already written the tile into the exchange buffer -- element (pos, col) at word
// lds_pad(pos*T + col), column pairs in split form as the stages write them -- and passed a barrier; the first stage then
// reads its inputs there like every later stage (x is not used).  This is how a pass chains two transforms over different
// axes of one tile without leaving the workgroup (k_rows_inv: FFT over the rows in registers, inverse FFT over the bins here).
template <int LOGF, int SIGN, bool STAGED = false, bool FROM_LDS = false, bool MIRROR = false, class Out, class Mid>
DEV void wgfft(cf* lds, const uint32_t ltw_off, uint32_t tid, const int logT, cx2 (&x)[NPAIR], Out& out, Mid& mid)
{
  static_assert(!MIRROR || FftPlan<LOGF>::NS >= 2, "MIRROR needs a last stage behind an exchange (P >= 2)");
  typedef FftPlan<LOGF> P;
  // opaque copy: LDS addresses and twiddle indices are loop-invariant in a persistent workgroup and
  // would otherwise be hoisted out of the tile loop and spilled (hundreds of registers)
  asm volatile("" : "+v"(tid));
  mid(0);
  if constexpr (P::NS <= 1) {
    wgfft_stage<P::LOGR1, SIGN, !FROM_LDS, true, STAGED>(lds, ltw_off, tid, logT, LOGF, 0, x, out, mid, 1);
  } else {
    if constexpr (!MIRROR) {
      // (this form -- the middle stages as an unrolled loop -- is kept for every pass that does not use MIRROR: written as the
      //  constexpr chain below, the fused inverse pass k_inv_chan<12, 1, 2> came out with 28 bytes of scratch per lane and ran 15 %
      //  slower, profiles/r05_experiments.txt item 7)
      wgfft_stage<4, SIGN, !FROM_LDS, false, STAGED>(lds, ltw_off, tid, logT, LOGF, 0, x, out, mid, 1);
      int logP = 4;
      uint32_t toff = ltw_off + (4u << (LOGF - 4));
#pragma unroll
      for (int j = 1; j < P::NQ - (P::REM ? 0 : 1); j++) {
        wgfft_stage<4, SIGN, false, false, STAGED>(lds, toff, tid, logT, LOGF, logP, x, out, mid, j + 1);
        logP += 4;
        toff += 4u << (LOGF - logP);
      }
      if constexpr (P::REM != 0) wgfft_stage<P::REM, SIGN, false, true, STAGED>(lds, toff, tid, logT, LOGF, logP, x, out, mid, P::NS);
      else wgfft_stage<4, SIGN, false, true, STAGED>(lds, toff, tid, logT, LOGF, logP, x, out, mid, P::NS);
    } else {
    // radix-16 stages between the first and the last: at most two (LOGF <= 13); the one in front of a MIRROR stage writes ILV
    constexpr int NMID = P::NQ - (P::REM ? 0 : 1) - 1;
    static_assert(NMID >= 0 && NMID <= 2, "wgfft: at most four stages");
    wgfft_stage<4, SIGN, !FROM_LDS, false, STAGED, false, MIRROR && NMID == 0>(lds, ltw_off, tid, logT, LOGF, 0, x, out, mid, 1);
    int logP = 4;
    uint32_t toff = ltw_off + (4u << (LOGF - 4));
    if constexpr (NMID >= 1) {
      wgfft_stage<4, SIGN, false, false, STAGED, false, MIRROR && NMID == 1>(lds, toff, tid, logT, LOGF, logP, x, out, mid, 2);
      logP += 4;
      toff += 4u << (LOGF - logP);
    }
    if constexpr (NMID >= 2) {
      wgfft_stage<4, SIGN, false, false, STAGED, false, MIRROR>(lds, toff, tid, logT, LOGF, logP, x, out, mid, 3);
      logP += 4;
      toff += 4u << (LOGF - logP);
    }
    if constexpr (P::REM != 0) wgfft_stage<P::REM, SIGN, false, true, STAGED, MIRROR>(lds, toff, tid, logT, LOGF, logP, x, out, mid, P::NS);
    else wgfft_stage<4, SIGN, false, true, STAGED, MIRROR>(lds, toff, tid, logT, LOGF, logP, x, out, mid, P::NS);
    }
  }
}
template <int LOGF, int SIGN, bool STAGED = false, bool FROM_LDS = false, bool MIRROR = false, class Out>
DEV void wgfft(cf* lds, const uint32_t ltw_off, uint32_t tid, const int logT, cx2 (&x)[NPAIR], Out& out)
{
  NoMid mid;
  wgfft<LOGF, SIGN, STAGED, FROM_LDS, MIRROR>(lds, ltw_off, tid, logT, x, out, mid);
}

// LDS-DMA of 16 bytes per lane (a plan entry, a piece of an 8-bit block), global -> LDS without passing through registers (lane l of the wave lands at
// `lds_wave_base` + 16*l), issued from inline assembly: the compiler does not see a vector-memory operation, so it does NOT put
// `s_waitcnt vmcnt(0)` in front of the next barrier.  With __builtin_amdgcn_global_load_lds it did -- in the middle of the
// transform, where that wait also drained the whole prefetch of the next tile, issued just before (ISA of round 3's
// k_inv_chan<12,true,2>: global_load_lds_dwordx4 ... s_waitcnt vmcnt(0); s_barrier between the second and the third stage;
// the stamps of profiles/r03_experiments.txt item 4 show the transform phase 1.8k cycles longer for it).  The hardware needs
// no such wait: a barrier does not drain vector memory (MI355X_MICROARCH.md, "Two waves per SIMD" item 7); what orders a reader
// behind the DMA is the issuing wave's covering vmcnt wait plus a barrier, and the callers have both: every tile begins with an
// explicit `s_waitcnt vmcnt(0)` and the entries are read behind the tile's first exchange barrier.  An operation the compiler
// does not count only makes its own counted waits more conservative (the counter is in order).  m0 (the LDS base of the DMA)
// is saved and restored inside the block.
DEV void lds_dma_b128(const void* gsrc, const uint32_t lds_wave_base)
{
  const uint32_t sb = __builtin_amdgcn_readfirstlane(lds_wave_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(sb) : "memory");
}
DEV uint32_t lds_byte_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }

// Work distribution of a persistent grid.  Items are dealt to the 8 XCDs in runs of `run`
// consecutive items (blocks b and b+8 share an XCD under the observed round-robin placement;
// a different placement only changes speed), so neighbouring items -- which share 128-byte
// input lines (P1) or chirp rows (P3) -- are served by one XCD's L2.
// (32-bit arithmetic throughout: a launch holds fewer than 2^31 items -- the host sizes its launch groups accordingly --
// and 64-bit divisions by a run-time value cost hundreds of instructions per tile)
DEV bool persistent_item(uint32_t b, uint32_t grid, uint32_t j, uint32_t run, uint32_t total, uint32_t& item)
{
  if (grid & 7) {
    item = b + j * grid;
  } else {
    const uint32_t nxl = grid >> 3;
    const uint32_t q = j * nxl + (b >> 3);
    const uint32_t qr = q / run;
    item = qr * (8u * run) + (b & 7) * run + (q - qr * run);
  }
  return item < total;
}

}  // namespace dspsr_amd
