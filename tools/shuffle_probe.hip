// Probe: the 16 x 16 transposition between two radix-16 stages (every thread holds 16 elements of 16 bytes = one packed
// pair of complex points; element k of lane j must reach lane k as its element j, inside each row of 16 lanes)
//   (a) through LDS, as wgfft.h does it: 16 ds_write_b128 + barrier + 16 ds_read_b128
//   (b) inside the wave: four butterfly steps (lane distance 1, 2, 4, 8) of DPP moves and selects -- the cheapest cross-lane
//       form of a transposition (a rotation-based one needs dynamic register indexing)
// 256 workgroups x 512 threads, NIT exchanges each, results checked against each other.  Reports ns per exchange and wave
// and the vector instructions per exchange in the ISA (hipcc -S).  BASELINE.json's north star names "wavefront-shuffle
// butterflies"; this is the measurement behind DESIGN.md section 4, item 7.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NIT = 256;

__device__ __forceinline__ float dpp_xor(float v, int d)       // value of lane ^ d inside a row of 16 lanes
{
  int i = __float_as_int(v), r;
  if (d == 1) r = __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
  else if (d == 2) r = __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  else if (d == 4) r = __builtin_amdgcn_ds_swizzle(i, 0x101F);                       // swap lanes ^4 (BitMode: xor 4)
  else r = __builtin_amdgcn_ds_swizzle(i, 0x201F);                                   // xor 8
  return __int_as_float(r);
}

template <bool SHUFFLE>
__global__ __launch_bounds__(512) void k_exchange(float4* out)
{
  extern __shared__ float4 sm[];
  const uint32_t tid = threadIdx.x, lane16 = tid & 15, row = tid >> 4;
  float4 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = make_float4(tid * 16 + k, 1.f + k, 2.f * tid, 0.5f * k);
  for (int it = 0; it < NIT; it++) {
    if (SHUFFLE) {
      // step d: lanes l and l^d exchange the elements whose index has bit d different from the lane's own bit d
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const bool up = (lane16 & d) != 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          if (k & d) continue;
          // pair (k, k|d): a lane with bit d clear keeps v[k] and receives the partner's v[k] into v[k|d]
          float4 a = v[k], b = v[k | d];
          float4 send = up ? a : b, got;
          got.x = dpp_xor(send.x, d); got.y = dpp_xor(send.y, d); got.z = dpp_xor(send.z, d); got.w = dpp_xor(send.w, d);
          v[k] = up ? got : a;
          v[k | d] = up ? b : got;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) sm[(row * 16 + k) * 17 + lane16] = v[k];          // element k of lane j at [k][j]
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = sm[(row * 16 + lane16) * 17 + k];          // lane k reads [k][.]
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 16; k++) v[k].x += 1.0f;          // keep the iterations dependent
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 16; k++) { s.x += v[k].x * (k + 1); s.y += v[k].y; s.z += v[k].z * (k + 1); s.w += v[k].w; }
  out[blockIdx.x * blockDim.x + tid] = s;
}

int main()
{
  float4 *o0, *o1;
  const size_t n = 256 * 512;
  CHECK(hipMalloc(&o0, n * sizeof(float4))); CHECK(hipMalloc(&o1, n * sizeof(float4)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t lds = 32 * 16 * 17 * sizeof(float4);
  CHECK(hipFuncSetAttribute((const void*)k_exchange<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best[2] = {1e9f, 1e9f};
  for (int r = 0; r < 5; r++)
    for (int m = 0; m < 2; m++) {
      CHECK(hipEventRecord(e0));
      if (m == 0) hipLaunchKernelGGL(k_exchange<false>, dim3(256), dim3(512), lds, 0, o0);
      else hipLaunchKernelGGL(k_exchange<true>, dim3(256), dim3(512), 0, 0, o1);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best[m]) best[m] = ms;
    }
  float4* h0 = (float4*)malloc(n * sizeof(float4)); float4* h1 = (float4*)malloc(n * sizeof(float4));
  CHECK(hipMemcpy(h0, o0, n * sizeof(float4), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1, o1, n * sizeof(float4), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) if (h0[i].x != h1[i].x || h0[i].y != h1[i].y || h0[i].z != h1[i].z || h0[i].w != h1[i].w) bad++;
  printf("transposition of 16 x 16 packed pairs inside rows of 16 lanes, %d exchanges per workgroup, 256 x 512 threads\n", NIT);
  printf("  through LDS (16 ds_write_b128 + 16 ds_read_b128 + 2 barriers): %.3f ms = %.0f ns per exchange and workgroup\n", best[0], best[0] * 1e6 / NIT);
  printf("  cross-lane (4 steps of DPP / swizzle moves and selects)       : %.3f ms = %.0f ns per exchange and workgroup\n", best[1], best[1] * 1e6 / NIT);
  printf("  results %s (%zu differences)\n", bad ? "DIFFER" : "identical", bad);
  return bad ? 1 : 0;
}
