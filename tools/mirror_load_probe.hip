// Probe for k_inv_a's prefetch (real input, blocked spectrum): per tile a workgroup reads NRUN runs of RUNB bytes, STRIDE bytes
// apart, from an ascending stream and as many from a descending (mirror) stream, all in one burst, then consumes them.
//   mode 0: 8 B per lane, both streams (32 loads per thread)            -- what the pass does today
//   mode 1: 16 B per lane ascending + 16 B per lane at an 8-byte-aligned address descending (16 loads)
//   mode 2: 16 B per lane, both aligned (16 loads)                      -- bound for mode 1
//   mode 3: as mode 0 with the mirror stream ascending too              -- does the direction matter?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
typedef v4 v4u __attribute__((aligned(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k_probe(const char* __restrict__ src, const uint64_t half, const uint32_t ntile, const uint64_t stride,
                                               const uint64_t tile_step, float* sink)
{
  float acc = 0.f;
  const uint32_t tid = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    const char* a = src + (uint64_t)t * tile_step;                       // ascending stream: runs of 1 KB, `stride` apart
    const char* b = src + 2 * half - (uint64_t)t * tile_step;            // mirror stream runs down from the end
    if (MODE == 0 || MODE == 3) {
      v2 va[16], vb[16];
      // 512 threads x 8 B = 4 KB = 4 runs per instruction
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint64_t e = tid + 512u * i, run = e >> 7, w = e & 127;
        va[i] = __builtin_nontemporal_load((const v2*)(a + run * stride + w * 8));
        vb[i] = MODE == 0 ? __builtin_nontemporal_load((const v2*)(b - run * stride - w * 8 - 8))
                          : __builtin_nontemporal_load((const v2*)(b - (run + 1) * stride + w * 8));
      }
#pragma unroll
      for (int i = 0; i < 16; i++) acc += va[i].x * va[i].y + vb[i].y * vb[i].x;
    } else {
      v4 va[8], vb[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint64_t e = 2 * (tid + 512u * i), run = e >> 7, w = e & 127;
        va[i] = __builtin_nontemporal_load((const v4*)(a + run * stride + w * 8));
        if (MODE == 1) vb[i] = __builtin_nontemporal_load((const v4u*)(b - run * stride - w * 8 - 8));     // elements L-k-1, L-k: 8 mod 16
        else vb[i] = __builtin_nontemporal_load((const v4*)(b - run * stride - w * 8 - 16));
      }
#pragma unroll
      for (int i = 0; i < 8; i++) acc += va[i].x * va[i].y + va[i].z * va[i].w + vb[i].w * vb[i].x + vb[i].y * vb[i].z;
    }
  }
  if (acc == 1.2345f) sink[0] = acc;
}
int main(int argc, char** argv)
{
  const uint64_t half = 1024ull << 20;                 // two streams of 1 GiB
  const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 0) : (256u << 10) + 1024;
  char* a; float* sink;
  CHECK(hipMalloc(&a, 2 * half + 4096)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(a, 0, 2 * half + 4096));
  // a tile = 64 runs per stream (64 KB + 64 KB); tiles advance by one run inside the first stride, then by 64 strides
  const uint32_t runs_per_stride = (uint32_t)(stride / 1024);
  const uint32_t ntile_max = (uint32_t)((half - 64 * stride) / 1024);
  const uint32_t ntile = ntile_max < 16384 ? ntile_max : 16384;
  (void)runs_per_stride;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char* names[4] = {"8 B/lane both (32 loads)", "16 B + 16 B at 8 mod 16 (16 loads)", "16 B both aligned (16 loads)", "8 B/lane both ascending"};
  for (int mode = 0; mode < 4; mode++) {
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(512), 0, 0, a + 2048, half, ntile, stride, 1024, sink);
      if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(256), dim3(512), 0, 0, a + 2048, half, ntile, stride, 1024, sink);
      if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(256), dim3(512), 0, 0, a + 2048, half, ntile, stride, 1024, sink);
      if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(256), dim3(512), 0, 0, a + 2048, half, ntile, stride, 1024, sink);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    printf("stride %llu  %-36s %.3f ms  %.2f TB/s  (%.0f ns per 128 KB tile and CU)\n", (unsigned long long)stride, names[mode], best,
           (double)ntile * 131072 / best / 1e9, best * 1e6 / (ntile / 256.0));
  }
  return 0;
}
