// Phase stamps (diagnostic builds only: make EXTRA=-DFB_STAMPS=<id> OUT=... OBJDIR=..., read by tools/stamps.py): where a tile of a
// persistent kernel spends its cycles -- s_memtime per phase on lane 0 of wave 0 of every workgroup.
//   ids: 1 k_fwd_cols, 2 k_fwd_rows, 3 k_inv_chan (fused fold), 4 k_inv_a, 6 k_fwd_col1q, 7 k_rows_inv, 8 k_tfp
// A kernel marks its phases in program order: FB_ST_BEGIN(id) in front of the tile loop, FB_ST(id, 0) at the top of a tile (it
// first waits for the prefetched tile, so that the wait is timed on its own), FB_ST(id, k) behind phase k, FB_ST_TILE(id, n)
// behind the last of the n phases, FB_ST_END(id) behind the loop.  Row b of the table = cycles of phases 0 .. n-1 summed over the
// tiles of workgroup b, the number of tiles in column 7.  The table lives in the translation unit of the kernel (device globals
// are per code object without -fgpu-rdc); FB_ST_READER(name) at the end of a unit exports dspsr_amd_debug_stamps_<name>.
// Without -DFB_STAMPS every macro is empty: the shipped library holds none of this.
#pragma once
#ifdef FB_STAMPS
#include <hip/hip_runtime.h>
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
static __device__ unsigned long long g_stamps[1024][8];
#define FB_ST_BEGIN(id) [[maybe_unused]] unsigned long long st_t[8] = {}, st_acc[8] = {}, st_prev = 0; do { if (FB_STAMPS == (id)) STAMP(st_prev); } while (0)
#define FB_ST(id, k) do { if (FB_STAMPS == (id)) { if ((k) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(st_t[k]); } } while (0)
#define FB_ST_TILE(id, n) do { if (FB_STAMPS == (id)) { st_acc[0] += st_t[0] - st_prev; for (int q_ = 1; q_ < (n); q_++) st_acc[q_] += st_t[q_] - st_t[q_ - 1]; st_acc[7] += 1; st_prev = st_t[(n) - 1]; } } while (0)
#define FB_ST_END(id) do { if (FB_STAMPS == (id) && threadIdx.x == 0 && blockIdx.x < 1024) for (int q_ = 0; q_ < 8; q_++) atomicAdd(&g_stamps[blockIdx.x][q_], st_acc[q_]); } while (0)
#define FB_ST_READER(name)                                                                                                   \
  extern "C" int dspsr_amd_debug_stamps_##name(unsigned long long* out_host, int zero)                                         \
  {                                                                                                                            \
    if (zero) { static unsigned long long z[1024][8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }     \
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(g_stamps));                                         \
  }
#else
#define FB_ST_BEGIN(id)
#define FB_ST(id, k)
#define FB_ST_TILE(id, n)
#define FB_ST_END(id)
#define FB_ST_READER(name)
#endif
