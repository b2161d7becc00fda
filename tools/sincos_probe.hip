// Accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) for FFT twiddles exp(-2*pi*i*j/2^23):
// max and rms absolute error against double precision over all j.  Build: hipcc --offload-arch=gfx950 -O2 -o sincos_probe sincos_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(float* s, float* c, int logL)
{
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const float x = (float)j * (1.0f / (float)(1u << logL));          // exact
  s[j] = __builtin_amdgcn_sinf(x);
  c[j] = __builtin_amdgcn_cosf(x);
}
// two-step form for L > 2^24 (j / L no longer exact in float): exp(-2 pi i j / L) = w(hi) * w(lo), hi = j >> s (13 bits,
// x = hi / 2^13 exact), lo = j mod 2^s (x = lo / L exact) -- what twiddles_big (csrc/filterbank.hip) does instead of two table
// look-ups; kernel writes the complex product
__global__ void k2(float2* w, int logL)
{
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = logL - 13;
  const float xh = (float)(j >> s) * (1.0f / 8192.0f), xl = (float)(j & ((1u << s) - 1)) * __uint_as_float((uint32_t)(127 - logL) << 23);
  const float ch = __builtin_amdgcn_cosf(xh), sh = __builtin_amdgcn_sinf(xh), cl = __builtin_amdgcn_cosf(xl), sl = __builtin_amdgcn_sinf(xl);
  w[j] = make_float2(ch * cl - sh * sl, -(ch * sl + sh * cl));
}
static void composite(int logL)
{
  const size_t n = (size_t)1 << logL;
  float2* dw;
  hipMalloc(&dw, n * 8);
  hipLaunchKernelGGL(k2, dim3(n / 256), dim3(256), 0, 0, dw, logL);
  std::vector<float2> w(n);
  hipMemcpy(w.data(), dw, n * 8, hipMemcpyDeviceToHost);
  double m = 0, r = 0, mt = 0;
  const int sh = logL - 14;
  for (size_t j = 0; j < n; j++) {
    const double a = 2.0 * M_PI * (double)j / (double)n;
    const double e = hypot((double)w[j].x - cos(a), (double)w[j].y + sin(a));
    if (e > m) m = e;
    r += e * e;
    if ((j & 63) == 0) {     // the table form it replaces: float(coarse) * float(fine), both correctly rounded
      const double ah = 2.0 * M_PI * (double)(j >> sh) / 16384.0, al = 2.0 * M_PI * (double)(j & (((size_t)1 << sh) - 1)) / (double)n;
      const float hc = (float)cos(ah), hs = (float)-sin(ah), lc = (float)cos(al), ls = (float)-sin(al);
      const float tx = hc * lc - hs * ls, ty = hc * ls + hs * lc;
      const double et = hypot((double)tx - cos(a), (double)ty + sin(a));
      if (et > mt) mt = et;
    }
  }
  printf("L = 2^%d two-step v_sin/v_cos: max |err| %.3e rms %.3e   (coarse x fine tables: max |err| %.3e)\n", logL, m, sqrt(r / n), mt);
  hipFree(dw);
}
int main()
{
  composite(25);
  composite(26);
  const int logL = 23;
  const size_t n = 1u << logL;
  float *ds, *dc;
  hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, ds, dc, logL);
  std::vector<float> s(n), c(n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double ms = 0, mc = 0, rs = 0, rc = 0;
  for (size_t j = 0; j < n; j++) {
    const double a = 2.0 * M_PI * (double)j / (double)n;
    const double es = fabs((double)s[j] - sin(a)), ec = fabs((double)c[j] - cos(a));
    if (es > ms) ms = es;
    if (ec > mc) mc = ec;
    rs += es * es; rc += ec * ec;
  }
  printf("v_sin_f32: max abs err %.3e rms %.3e   v_cos_f32: max abs err %.3e rms %.3e   (float eps/2 = 5.96e-08)\n", ms, sqrt(rs / n), mc,
         sqrt(rc / n));
  // correctly rounded reference for comparison
  double mr = 0;
  for (size_t j = 0; j < n; j += 7) { const double a = 2.0 * M_PI * (double)j / (double)n; const double e = fabs((double)(float)sin(a) - sin(a)); if (e > mr) mr = e; }
  printf("correctly rounded float: max abs err %.3e\n", mr);
  return 0;
}
