// Probe: does a streaming read with 8-byte-per-lane loads (the inverse pass: one complex element per lane) run slower
// than the same bytes with 16-byte-per-lane loads (pass 2)?  Persistent grid, 256 x 512 threads, every wave instruction
// reads one contiguous run (512 B or 1 KB), 2 GiB footprint, 32 (16) loads in flight per thread like the passes' prefetch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <typename V, int NL>
__global__ __launch_bounds__(512) void k_read(const V* __restrict__ src, const uint64_t n, float* sink, const int wg_per_tile_order)
{
  float acc = 0.f;
  const uint64_t per_tile = (uint64_t)blockDim.x * NL;                 // elements per tile of one workgroup
  const uint64_t ntile = n / per_tile;
  for (uint64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    const V* p = src + t * per_tile + threadIdx.x;
    V v[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) v[j] = __builtin_nontemporal_load(p + (uint64_t)j * blockDim.x);
#pragma unroll
    for (int j = 0; j < NL; j++) acc += ((const float*)&v[j])[0];
  }
  if (acc == 1.2345f) sink[0] = acc;
}
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
int main()
{
  const uint64_t total = 2048ull << 20;
  void* a; float* sink;
  CHECK(hipMalloc(&a, total)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(a, 0, total));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; mode++) {
    float best = 1e9;
    for (int r = 0; r < 6; r++) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL((k_read<v2, 32>), dim3(256), dim3(512), 0, 0, (const v2*)a, total / 8, sink, 0);
      if (mode == 1) hipLaunchKernelGGL((k_read<v4, 16>), dim3(256), dim3(512), 0, 0, (const v4*)a, total / 16, sink, 0);
      if (mode == 2) hipLaunchKernelGGL((k_read<v2, 32>), dim3(512), dim3(512), 0, 0, (const v2*)a, total / 8, sink, 0);
      if (mode == 3) hipLaunchKernelGGL((k_read<v4, 16>), dim3(512), dim3(512), 0, 0, (const v4*)a, total / 16, sink, 0);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    const char* names[4] = {"8 B/lane x32, 256 WG", "16 B/lane x16, 256 WG", "8 B/lane x32, 512 WG", "16 B/lane x16, 512 WG"};
    printf("%-24s %.3f ms  %.2f TB/s\n", names[mode], best, total / best / 1e9);
  }
  return 0;
}
