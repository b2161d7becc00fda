// Probe: streaming store / load rate on MI355X as a function of the contiguous run length, with the runs of
// one wave instruction scattered at a large stride (the access shape of the blocked scratch layouts:
// P1 writes T1*T2-element runs, P2 writes T2*T3-element runs).
//   ./runprobe <mode 0 store|1 load|2 copy> <run_bytes> <stride_bytes>
// Total footprint 512 MiB, persistent grid of 256 x 512 threads, 16 B per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// logical 16-byte unit u -> address: runs of `run_v4` units; consecutive runs are `stride_v4` apart and wrap
// around so that every unit of the buffer is touched exactly once
__device__ __forceinline__ uint64_t map(uint64_t u, uint32_t run_v4, uint64_t stride_v4, uint64_t total_v4)
{
  const uint64_t r = u / run_v4, w = u % run_v4;
  const uint64_t runs_per_stride = stride_v4 / run_v4;          // runs that fit between two strided neighbours
  const uint64_t nstrides = total_v4 / stride_v4;
  const uint64_t a = r % nstrides, b = r / nstrides;            // a-th strided position, b-th run inside it
  return a * stride_v4 + (b % runs_per_stride) * run_v4 + w;
}

__global__ __launch_bounds__(512) void k_rw(float4* dst, const float4* src, int mode, uint32_t run_v4,
                                            uint64_t stride_v4, uint64_t total_v4, float* sink)
{
  float4 acc = make_float4(0, 0, 0, 0);
  const uint64_t nthr = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t u0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u0 < total_v4; u0 += nthr * 4) {
    float4 v[4];
    uint64_t ad[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t u = u0 + j * nthr;
      ad[j] = u < total_v4 ? map(u, run_v4, stride_v4, total_v4) : 0;
    }
    if (mode >= 1) {
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = src[ad[j]];
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = make_float4(u0, j, 1.f, 2.f);
    }
    if (mode != 1) {
#pragma unroll
      for (int j = 0; j < 4; j++) dst[ad[j]] = v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) { acc.x += v[j].x; acc.y += v[j].w; }
    }
  }
  if (acc.x == 1.2345f) sink[0] = acc.y;
}

int main(int argc, char** argv)
{
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const uint32_t run = argc > 2 ? atoi(argv[2]) : 128;
  const uint64_t stride = argc > 3 ? strtoull(argv[3], 0, 10) : 65536;
  const uint64_t total = 512ull << 20;
  float4 *a, *b; float* sink;
  CHECK(hipMalloc(&a, total)); CHECK(hipMalloc(&b, total)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(a, 0, total)); CHECK(hipMemset(b, 0, total));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9;
  for (int r = 0; r < 6; r++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rw, dim3(256 * 4), dim3(512), 0, 0, a, b, mode, run / 16, stride / 16, total / 16, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best) best = ms;
  }
  const double bytes = (mode == 2 ? 2.0 : 1.0) * total;
  printf("mode %d run %u B stride %llu B: %.3f ms  %.2f TB/s\n", mode, run, (unsigned long long)stride, best, bytes / best / 1e9);
  return 0;
}
