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
int main()
{
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
