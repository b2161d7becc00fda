// k_inv_chan with complex / detected output + k_time_combine (freq_res = R * 2^k)
#include "fb_inv_chan.h"

namespace dspsr_amd {

// freq_res = R * 2^k, last step: y[n] = sum_r exp(+2 pi i r n / freq_res) y_r[n mod M'] for the kept samples n of every channel and
// part, from the pseudo-channels' whole transforms Y[c*R + r][pol][part][M'] (written by the inverse pass as complex rows), into
// the caller's output: complex rows (kind 1) or detected samples (kind 2; Detection.C:423-474 layouts as in k_inv_chan).
// RT = 0: the factor at run time (p.tw.R: any odd number <= ODD_MAX)
template <int RT>
__global__ __launch_bounds__(256) void k_time_combine(const TimeCombine p, const FbOut out)
{
  const uint32_t R = RT ? (uint32_t)RT : p.tw.R;
  constexpr int UNR = RT ? RT : 1;
  const uint32_t Mi = 1u << p.logMi;
  const uint64_t n = (uint64_t)p.nparts * p.C * p.nkeep;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(i % p.nkeep);
    const uint64_t pc = i / p.nkeep;
    const uint32_t c = (uint32_t)(pc % p.C), lp = (uint32_t)(pc / p.C);
    const uint32_t nn = p.nfilt_pos + t, ni = nn & (Mi - 1);
    cf a = make_float2(0.f, 0.f), b = a;
#pragma unroll UNR
    for (uint32_t r = 0; r < R; r++) {
      const cf* __restrict__ y = p.Y + (uint64_t)(c * R + r) * p.y_chan_stride + ((uint64_t)lp << p.logMi) + ni;
      cf w = make_float2(1.f, 0.f);
      if (r) {                                                   // exp(+2 pi i r n / freq_res), freq_res = R M' (r n < 127 * 127 * 2^13 < 2^32: the run-time form takes 64 bits anyway)
        if constexpr (RT != 0) w = twiddle_odd<RT ? RT : 3>(r * nn, p.logMi, p.tw);
        else w = twiddle_odd_rt((uint64_t)r * nn, p.logMi, p.tw);
        w.y = -w.y;
      }
      const cf v0 = cmul(y[0], w);
      a.x += v0.x; a.y += v0.y;
      if (p.npol == 2) { const cf v1 = cmul(y[p.y_pol_stride], w); b.x += v1.x; b.y += v1.y; }
    }
    const uint64_t part = p.part0 + (uint64_t)lp * p.part_stride;
    const uint32_t chan = out.chan0 + c;
    float* __restrict__ row = out.base + chan * out.chan_stride;
    if (out.kind == 1) {
      float2* __restrict__ o = (float2*)(row + part * out.part_step) + t;
      *o = a;
      if (p.npol == 2) *(float2*)((float*)o + out.pol_stride) = b;
    } else if (out.kind == 2) {
      float q[4];
      detect4(a, b, out.state, q);
      const uint64_t idat = part * p.nkeep + t;
      if (out.ndim == 4) ((float4*)row)[idat] = make_float4(q[0], q[1], q[2], q[3]);
      else if (out.ndim == 2) {
        ((float2*)row)[idat] = make_float2(q[0], q[1]);
        ((float2*)(row + out.pol_stride))[idat] = make_float2(q[2], q[3]);
      } else {
        row[idat] = q[0];
        row[out.pol_stride + idat] = q[1];
        row[2 * out.pol_stride + idat] = q[2];
        row[3 * out.pol_stride + idat] = q[3];
      }
    }
  }
}

template <int... I> static k3_t pick3(int logf, bool full, iseq<I...>)
{
  static const k3_t t[] = {k_inv_chan<I, 0, -1>...};
  static const k3_t f[] = {k_inv_chan<I, 0, full_logt(I)>...};
  return full ? f[logf] : t[logf];
}
k3_t fb_pick3(int logf, bool full) { return pick3(logf, full, seq_t()); }
void fb_launch_time_combine(hipStream_t stream, const TimeCombine& p, const FbOut& out, uint32_t R, uint32_t ncu)
{
  switch (R) {
    case 3: hipLaunchKernelGGL(k_time_combine<3>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 5: hipLaunchKernelGGL(k_time_combine<5>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 7: hipLaunchKernelGGL(k_time_combine<7>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 9: hipLaunchKernelGGL(k_time_combine<9>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 15: hipLaunchKernelGGL(k_time_combine<15>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    default: hipLaunchKernelGGL(k_time_combine<0>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
  }
}

}  // namespace dspsr_amd

FB_ST_READER(inv_chan)
