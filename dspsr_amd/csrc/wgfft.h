// In-workgroup batched FFT for gfx950 (wave64): F-point FFTs over T interleaved columns,
// F*T = 16*blockDim.x points resident in LDS, every thread owning 16 points per stage.
//
// Stockham auto-sort, radix-16 stages (4x4 butterflies held in registers) plus one
// radix-2/4/8 remainder stage.  Element (pos, col) lives at LDS word lds_pad(pos*T + col).
// Stage with radix R, P = product of earlier radices, Q = F/(P*R):
//   butterfly u = col + T*(p + P*s)        (p < P, s < Q; consecutive lanes -> consecutive words)
//   reads   u + i*(F/R)*T                  i < R
//   twiddle W_{R*Q}^{k*s}
//   writes  (s*P*R + k*P + p)*T + col      k < R
// (index math validated against numpy in tests/test_wgfft_model.py)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dspsr_amd {

typedef float2 cf;

#define DEV __device__ __forceinline__

DEV cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
DEV cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
DEV cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
DEV cf cconj(cf a) { return make_float2(a.x, -a.y); }
// multiply by SIGN*i  (forward SIGN=-1: -i ; inverse SIGN=+1: +i)
template <int SIGN> DEV cf mul_si(cf a) { return SIGN < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x); }

// 4 words of padding per 64 keep the strided stage writes spread over the LDS banks
// and preserve 16-byte alignment of even word indices
DEV uint32_t lds_pad(uint32_t e) { return e + ((e >> 6) << 2); }
inline uint32_t lds_words_host(uint32_t points) { return points + ((points >> 6) << 2) + 8; }

// cos/sin(2*pi*k/16)
#define C16_1 0.92387953251128674f
#define S16_1 0.38268343236508977f
#define C16_2 0.70710678118654752f

template <int SIGN> DEV void fft2(cf& a, cf& b) { cf t = a; a = cadd(t, b); b = csub(t, b); }

template <int SIGN> DEV void fft4(cf& v0, cf& v1, cf& v2, cf& v3)
{
  cf t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = mul_si<SIGN>(csub(v1, v3));
  v0 = cadd(t0, t2); v2 = csub(t0, t2);
  v1 = cadd(t1, t3); v3 = csub(t1, t3);
}

// multiply by W_8^1 = exp(SIGN*i*pi/4) and W_8^3
template <int SIGN> DEV cf mul_w8_1(cf a)
{ return SIGN < 0 ? make_float2(C16_2 * (a.x + a.y), C16_2 * (a.y - a.x)) : make_float2(C16_2 * (a.x - a.y), C16_2 * (a.x + a.y)); }
template <int SIGN> DEV cf mul_w8_3(cf a)
{ return SIGN < 0 ? make_float2(C16_2 * (a.y - a.x), -C16_2 * (a.x + a.y)) : make_float2(-C16_2 * (a.x + a.y), C16_2 * (a.x - a.y)); }
template <int SIGN> DEV cf mul_w(cf a, float c, float s) { return cmul(a, make_float2(c, SIGN < 0 ? -s : s)); }

template <int SIGN> DEV void fft8(cf (&v)[8])
{
  // DIT: even/odd 4-point transforms then combine
  fft4<SIGN>(v[0], v[2], v[4], v[6]);
  fft4<SIGN>(v[1], v[3], v[5], v[7]);
  cf o1 = mul_w8_1<SIGN>(v[3]), o2 = mul_si<SIGN>(v[5]), o3 = mul_w8_3<SIGN>(v[7]);
  cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

template <int SIGN> DEV void fft16(cf (&v)[16])
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
  cf t;
  t = v[1]; v[1] = v[4]; v[4] = t;
  t = v[2]; v[2] = v[8]; v[8] = t;
  t = v[3]; v[3] = v[12]; v[12] = t;
  t = v[6]; v[6] = v[9]; v[9] = t;
  t = v[7]; v[7] = v[13]; v[13] = t;
  t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int R, int SIGN> DEV void fftR(cf (&v)[R])
{
  if constexpr (R == 2) fft2<SIGN>(v[0], v[1]);
  else if constexpr (R == 4) fft4<SIGN>(v[0], v[1], v[2], v[3]);
  else if constexpr (R == 8) fft8<SIGN>(v);
  else fft16<SIGN>(v);
}

// twiddle table: tw[j] = exp(-2*pi*i*j/TWN), j < TWN
constexpr int LOG_TWN = 14;
constexpr int TWN = 1 << LOG_TWN;

template <int LOGR, int SIGN>
DEV void wgfft_stage(cf* lds, const uint32_t tid, const uint32_t nt, const int logT, const int logF,
                     const int logP, const cf* __restrict__ tw)
{
  constexpr int R = 1 << LOGR;
  constexpr int G = 16 / R;
  const int logQ = logF - logP - LOGR;
  const uint32_t stride = 1u << (logF - LOGR + logT);
  cf v[G][R];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const uint32_t u = tid + nt * g;
#pragma unroll
    for (int i = 0; i < R; i++) v[g][i] = lds[lds_pad(u + i * stride)];
  }
#pragma unroll
  for (int g = 0; g < G; g++) {
    fftR<R, SIGN>(v[g]);
    if (logQ > 0) {
      const uint32_t u = tid + nt * g;
      const uint32_t s = u >> (logT + logP);
      const int sh = LOG_TWN - LOGR - logQ;
#pragma unroll
      for (int k = 1; k < R; k++) {
        cf w = tw[(k * s) << sh];
        if (SIGN > 0) w.y = -w.y;
        v[g][k] = cmul(v[g][k], w);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < G; g++) {
    const uint32_t u = tid + nt * g;
    const uint32_t col = u & ((1u << logT) - 1);
    const uint32_t rest = u >> logT;
    const uint32_t p = rest & ((1u << logP) - 1);
    const uint32_t s = rest >> logP;
#pragma unroll
    for (int k = 0; k < R; k++) {
      const uint32_t pos = (s << (logP + LOGR)) + ((uint32_t)k << logP) + p;
      lds[lds_pad((pos << logT) | col)] = v[g][k];
    }
  }
  __syncthreads();
}

// Pre : element (n, col) at lds_pad(n*T + col), all threads synchronised.
// Post: element (k, col) at lds_pad(k*T + col), all threads synchronised.
template <int LOGF, int SIGN>
DEV void wgfft(cf* lds, const uint32_t tid, const uint32_t nt, const int logT, const cf* __restrict__ tw)
{
  constexpr int NQ = LOGF / 4, REM = LOGF % 4;
  int logP = 0;
#pragma unroll
  for (int j = 0; j < NQ; j++) { wgfft_stage<4, SIGN>(lds, tid, nt, logT, LOGF, logP, tw); logP += 4; }
  if constexpr (REM != 0) wgfft_stage<REM, SIGN>(lds, tid, nt, logT, LOGF, logP, tw);
}

}  // namespace dspsr_amd
