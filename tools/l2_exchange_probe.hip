// Probe: can the workgroups of one XCD exchange a 2 MB scratch through their shared L2 inside one
// persistent launch, without the bytes crossing the fabric?  (Design question behind the fused
// filterbank passes: DESIGN.md "XCD-local exchange".)
//
//   mode 0: scratch re-used every iteration (L2-resident candidate), sc1 (L1-bypass) loads
//   mode 1: scratch advances through a large buffer every iteration (HBM / Infinity-Cache path)
//   mode 2: as mode 0 with plain loads (shows whether stale L1 lines are observed)
//   mode 3: barriers only
//   mode 4: as mode 0, output written with sc1 stores (dropped from L2 after the write)
//   mode 5: as mode 0, output written with nt stores
// argv: mode iters reps chunk_kb (scratch per XCD = 32 x chunk)
// Teams are formed from HW_REG_XCC_ID, every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o l2probe tools/l2_exchange_probe.hip && ./l2probe <mode> <iters>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NT = 512;

constexpr uint64_t SPIN_LIMIT = 4000000ull;

struct Ctl {
  unsigned team_cnt[8];
  unsigned total;
  unsigned arrive[8 * 32];    // one line per XCD
  unsigned timeout;
  unsigned bad;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, Ctl* ctl)
{
  for (uint64_t i = 0; i < SPIN_LIMIT; i++) {
    if (ld_sc1(p) >= target) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  atomicExch(&ctl->timeout, 1u);
  return false;
}

__global__ __launch_bounds__(NT) void k_probe(Ctl* ctl, float4* scratch, float4* out, uint64_t scratch_stride_v4,
                                              uint64_t out_iters, int mode, int iters, unsigned* checksum_bad, const uint32_t CHUNK_V4)
{
  extern __shared__ float4 lds[];
  __shared__ unsigned s_xcc, s_rank, s_n;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;     // HW_REG_XCC_ID
    s_xcc = xcc;
    s_rank = atomicAdd(&ctl->team_cnt[xcc], 1u);
    __threadfence();
    atomicAdd(&ctl->total, 1u);
    wait_ge(&ctl->total, gridDim.x, ctl);
    s_n = ld_sc1(&ctl->team_cnt[xcc]);
  }
  __syncthreads();
  const unsigned xcc = s_xcc, rank = s_rank, n = s_n;
  unsigned* arr = &ctl->arrive[xcc * 32];
  unsigned nbad = 0;
  for (int it = 0; it < iters; it++) {
    float4* S = scratch + (uint64_t)xcc * 32 * CHUNK_V4 + (mode == 1 ? (uint64_t)it * scratch_stride_v4 : 0);
    if (mode != 3) {
      // phase 1: write my chunk
      for (uint32_t j = 0; j < CHUNK_V4 / NT; j++) {
        const uint32_t idx = j * NT + tid;
        const float v = (float)((it * 131 + rank * 17 + idx) & 0xffff);
        S[(uint64_t)rank * CHUNK_V4 + idx] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) { atomicAdd(arr, 1u); wait_ge(arr, n * (2 * it + 1), ctl); }
    __syncthreads();
    if (ld_sc1(&ctl->timeout)) break;
    if (mode != 3) {
      // phase 2: read slice `rank` of every chunk (all-to-all), write 64 KB of output
      float4* O = out + ((uint64_t)(it % out_iters) * 256 + blockIdx.x) * CHUNK_V4;
      const uint32_t per = CHUNK_V4 / n;          // 16-byte units taken from each chunk
      for (uint32_t j0 = 0; j0 < CHUNK_V4; j0 += NT * 4) {
        float4 v[4];
        uint32_t src_idx[4], src_c[4];
        const float4* src_p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t q = j0 + u * NT + tid;
          uint32_t c = q / per, w = q % per;
          if (c >= n) { c = n - 1; w = q - c * per; if (w >= CHUNK_V4) w = CHUNK_V4 - 1; }
          src_c[u] = c; src_idx[u] = rank * per + w < CHUNK_V4 ? rank * per + w : CHUNK_V4 - 1;
          const float4* p = &S[(uint64_t)c * CHUNK_V4 + src_idx[u]];
          src_p[u] = p;
        }
        if (mode == 2) {
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = *src_p[u];
        } else {
          // loads and their wait in ONE statement: no output may be touched before the wait
          typedef float f4 __attribute__((ext_vector_type(4)));
          f4 t0, t1, t2, t3;
          asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                       "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                       : "v"(src_p[0]), "v"(src_p[1]), "v"(src_p[2]), "v"(src_p[3]) : "memory");
          v[0] = make_float4(t0[0], t0[1], t0[2], t0[3]); v[1] = make_float4(t1[0], t1[1], t1[2], t1[3]);
          v[2] = make_float4(t2[0], t2[1], t2[2], t2[3]); v[3] = make_float4(t3[0], t3[1], t3[2], t3[3]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float e = (float)((it * 131 + src_c[u] * 17 + src_idx[u]) & 0xffff);
          if (v[u].x != e || v[u].w != e + 3.f) nbad++;
          float4* op = &O[j0 + u * NT + tid];
          typedef float f4 __attribute__((ext_vector_type(4)));
          const f4 t = {v[u].x, v[u].y, v[u].z, v[u].w};
          if (mode == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(op), "v"(t) : "memory");
          else if (mode == 5) __builtin_nontemporal_store(t, (f4*)op);
          else *op = v[u];
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) { atomicAdd(arr, 1u); wait_ge(arr, n * (2 * it + 2), ctl); }
    __syncthreads();
    if (ld_sc1(&ctl->timeout)) break;
  }
  if (nbad) atomicAdd(checksum_bad, nbad);
}

int main(int argc, char** argv)
{
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const uint32_t CHUNK_V4 = (argc > 4 ? atoi(argv[4]) : 64) * 64;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount;
  Ctl* ctl; float4 *scratch, *out; unsigned* bad;
  const uint64_t scratch_stride_v4 = 8ull * 32 * CHUNK_V4;          // 16 MB per iteration (mode 1)
  const uint64_t scratch_iters = mode == 1 ? iters : 1;
  const uint64_t out_iters = 64;                                   // 64 x 16 MB = 1 GB ring
  CHECK(hipMalloc(&ctl, sizeof(Ctl)));
  CHECK(hipMalloc(&scratch, scratch_iters * scratch_stride_v4 * sizeof(float4)));
  CHECK(hipMalloc(&out, out_iters * 256 * CHUNK_V4 * sizeof(float4)));
  CHECK(hipMalloc(&bad, 4));
  CHECK(hipMemset(bad, 0, 4));
  CHECK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < reps; r++) {
    CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe, dim3(grid), dim3(NT), 128 * 1024, 0, ctl, scratch, out, scratch_stride_v4, out_iters, mode, iters, bad, CHUNK_V4);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    Ctl h; unsigned hb;
    CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("mode %d chunk %u KB iters %d grid %d: %.3f ms, %.2f us/iter, teams %u %u %u %u %u %u %u %u, timeout %u, bad words %u\n", mode, CHUNK_V4 / 64, iters, grid, ms,
           ms * 1e3 / iters, h.team_cnt[0], h.team_cnt[1], h.team_cnt[2], h.team_cnt[3], h.team_cnt[4], h.team_cnt[5],
           h.team_cnt[6], h.team_cnt[7], h.timeout, hb);
  }
  return 0;
}
