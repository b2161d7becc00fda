// dsp::Convolution with a response of at most 8192 points in ONE tile pass (round 5).
//
// Reference: Signal/General/Convolution.C:338-461 (GPU twins ConvolutionCUDA.cu:552-800, ConvolutionCUDASpectral.cu:369-476) -- per
// (channel, polarisation, part): forward transform of n_fft complex samples, x response of the channel (Response.C:385-444),
// backward transform, samples [nfilt_pos, nfilt_pos + nsamp_step) kept.  The filterbank object runs this with nchan_subband = 1 as
// four tile passes (two forward, two inverse: fb_four_pass.hip) -- the shape of ONE long transform.  Behind a filterbank
// (`dspsr -F N`: many channels, short responses) the whole transform of a (channel, part) sequence fits a workgroup tile together
// with its second polarisation and, for n_fft < 8192, with more parts: forward transform, response, backward transform, keep window
// and Detection then happen between ONE read of the input rows and ONE write of the output rows -- a quarter of the HBM traffic.
//
// Tile = M-point transforms over T = 2^14 / M columns; column pair (2 j, 2 j + 1) = the two polarisations of part p0 + j of ONE
// channel (a tile never straddles channels, so the response is a function of the bin alone: 16 factors per thread, kept in
// registers while the workgroup walks the tiles of a channel).  The forward transform hands its last stage to a functor that
// multiplies by the response and writes the spectrum into the exchange buffer in the order the backward transform's first stage
// reads it (wgfft STAGED -> FROM_LDS, as k_rows_inv chains its two transforms, fb_two_pass.hip); the backward transform's last
// stage stores the kept samples (or their polarisation products) straight from registers.
// Complex float32 rows with two polarisations (what Convolution::Engine::perform is handed behind a filterbank).
#include "fb_common.h"

namespace dspsr_amd {

struct Conv1Params {
  const float* in;                 // rows: in + chan * chan_stride + pol * pol_stride + part * in_step, (re, im) pairs
  uint64_t chan_stride, pol_stride, in_step;     // floats
  const cf* kern;                  // [nchan][M] or null
  FbOut out;                       // kind 0 (none), 1 (complex rows), 2 (detected)
  uint32_t nchan, nfilt_pos, nkeep;
  uint64_t npart;
  uint32_t tiles_per_chan;
};

// the response factors of a thread: one per element the forward transform's last stage hands it, in the order of those calls
template <int N> struct Conv1Fwd {
  cf* lds;
  const cf* kk;
  int logT;
  int h;
  template <int R> DEV void operator()(const uint32_t col, const uint32_t p, const uint32_t pstride, cx2 (&v)[R])
  {
#pragma unroll
    for (int k = 0; k < R; k++) {
      const cx2 q = cmuls(v[k], kk[h * R + k]);                          // Response::operate: both polarisations share the factor
      const uint32_t e = ((k * pstride + p) << logT) + col;               // element (bin, column) of the backward transform's tile
      *(float4*)&lds[lds_pad(e)] = make_float4(q.x[0], q.x[1], q.y[0], q.y[1]);
    }
  }
};

// LOGP: points per tile -- 2^13 (256 threads, two workgroups per compute unit: one's store phase under the other's transforms) up to
// n_fft = 4096, 2^14 for n_fft = 8192 (two polarisations of one part)
constexpr int conv1_log_points(int logM) { return logM <= 12 ? 13 : 14; }
template <int LOGM>
__global__ __launch_bounds__(512) void k_conv1(const Conv1Params p, const cf* __restrict__ tw)
{
  typedef FftPlan<LOGM> P;
  static_assert(P::NS >= 2 && LOGM <= 13, "k_conv1: 64 <= n_fft <= 8192");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  constexpr int LOGP = conv1_log_points(LOGM);
  constexpr uint32_t nt = 1u << (LOGP - LOG_PTS);
  constexpr int logT = LOGP - LOGM;                                        // columns of the tile
  constexpr uint32_t T = 1u << logT, Ts = T / 2;                           // Ts parts of one channel, both polarisations
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LOGM>(lds, ltw_off, tw, tid, nt);
  const uint32_t total = p.nchan * p.tiles_per_chan;
  // every workgroup takes one contiguous range of tiles: it walks the parts of a channel, so the channel's response is loaded once
  uint32_t item = (uint32_t)(((uint64_t)total * blockIdx.x) / gridDim.x);
  const uint32_t item_end = (uint32_t)(((uint64_t)total * (blockIdx.x + 1)) / gridDim.x);
  if (item >= item_end) return;
  const uint64_t last_part = p.npart - 1;
  struct Pol2 { cf a, b; };
  auto fetch = [&](const uint32_t it, Pol2 (&raw)[NPAIR]) {
    const uint32_t chan = it / p.tiles_per_chan, tl = it - chan * p.tiles_per_chan;
    const float* __restrict__ row = p.in + (uint64_t)chan * p.chan_stride;
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        const uint32_t e = first_stage_elem<LOGM>(tid, logT, g2, i);
        const uint32_t j = (e & (T - 1)) >> 1, n = e >> logT;
        uint64_t part = (uint64_t)tl * Ts + j;
        part = part < last_part ? part : last_part;                       // (ragged last tile: loaded again, never stored)
        const float* __restrict__ q = row + part * p.in_step + 2 * n;
        Pol2 r;
        r.a = *(const float2*)q;
        r.b = *(const float2*)(q + p.pol_stride);
        raw[(g2 / 2) * P::R1 + i] = r;
      }
  };
  // forward last stage: radix RL, G = 32 / RL butterflies per thread, pair h = butterflies 2h, 2h + 1: bin k * (M / RL) + pp(h)
  constexpr int LOGRL = P::REM ? P::REM : 4, RL = 1 << LOGRL, GL = PTS / RL, HL = GL / 2, logPL = LOGM - LOGRL;
  cf kk[NPAIR];
  uint32_t kk_chan = ~0u;
  auto load_response = [&](const uint32_t chan) {
    if (!p.kern) {
#pragma unroll
      for (int q = 0; q < NPAIR; q++) kk[q] = make_float2(1.f, 0.f);
      return;
    }
    const cf* __restrict__ kc = p.kern + ((uint64_t)chan << LOGM);
#pragma unroll
    for (int h = 0; h < HL; h++) {
      const uint32_t u = GL * tid + 2 * h, pp = (u >> logT) & ((1u << logPL) - 1);
#pragma unroll
      for (int k = 0; k < RL; k++) kk[h * RL + k] = kc[((uint32_t)k << logPL) + pp];
    }
  };
  Pol2 raw[NPAIR];
  fetch(item, raw);
  for (;;) {
    asm volatile("" : "+v"(tid));
    const uint32_t chan = item / p.tiles_per_chan, tl = item - chan * p.tiles_per_chan;
    if (chan != kk_chan) { load_response(chan); kk_chan = chan; }
    cx2 x[NPAIR];
#pragma unroll
    for (int h = 0; h < NPAIR; h++) x[h] = make_cx2(raw[h].a, raw[h].b);
    const uint32_t next = item + 1;
    const bool more = next < item_end;
    fetch(more ? next : item, raw);          // unconditional: a conditional prefetch is waited for inside its block (fb_inv_chan.h)

    Conv1Fwd<NPAIR> fwd = {lds, kk, logT, 0};
    wgfft<LOGM, -1, true>(lds, ltw_off, tid, logT, x, fwd);
    __syncthreads();                          // the spectrum x response lies in the exchange buffer, element (bin, column)

    const FbOut& out = p.out;
    auto store = [&](const uint32_t col, const uint32_t pos, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if (out.kind == 0) return;
      const uint64_t part = (uint64_t)tl * Ts + (col >> 1);
      if (part > last_part) return;
      float* __restrict__ row = out.base + (uint64_t)(out.chan0 + chan) * out.chan_stride;
      const int32_t t0 = (int32_t)pos - (int32_t)p.nfilt_pos;
      if (out.kind == 1) {
        float2* __restrict__ o2 = (float2*)(row + part * out.part_step) + t0;
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= p.nkeep) continue;
          float2* o = o2 + k * pstride;
          st_stream(o, cx2_lo(v[k]));
          st_stream((float2*)((float*)o + out.pol_stride), cx2_hi(v[k]));
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t ts = t0 + (int32_t)(k * pstride);
          if ((uint32_t)ts >= p.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          const uint64_t idat = part * p.nkeep + (uint32_t)ts;
          if (out.ndim == 4) st_stream(&((float4*)row)[idat], make_float4(r[0], r[1], r[2], r[3]));
          else if (out.ndim == 2) {
            st_stream(&((float2*)row)[idat], make_float2(r[0], r[1]));
            st_stream(&((float2*)(row + out.pol_stride))[idat], make_float2(r[2], r[3]));
          } else {
            row[idat] = r[0];
            row[out.pol_stride + idat] = r[1];
            row[2 * out.pol_stride + idat] = r[2];
            row[3 * out.pol_stride + idat] = r[3];
          }
        }
      }
    };
    wgfft<LOGM, +1, false, true>(lds, ltw_off, tid, logT, x, store);
    if (!more) break;
    item = next;
    // (the next tile's first exchange write sits behind a barrier of its own inside wgfft: every wave has then read this tile's
    //  last exchange)
  }
}

typedef void (*kconv1_t)(Conv1Params, const cf*);
template <int... I> static kconv1_t pick_conv1(int logm, iseq<I...>)
{
  static const kconv1_t t[] = {k_conv1<I + 6>...};
  return logm >= 6 && logm < 6 + (int)sizeof...(I) ? t[logm - 6] : nullptr;
}

int fb_conv1_check(int logM, size_t* lds_bytes)
{
  kconv1_t k = pick_conv1(logM, mkseq<8>::type());                      // n_fft = 64 ... 8192
  if (!k) return DSPSR_AMD_EINVAL;
  const size_t lds = lds_total_words_host(1u << conv1_log_points(logM), logM) * sizeof(cf);
  if (lds_bytes) *lds_bytes = lds;
  return dspsr_amd_allow_lds((const void*)k, lds) == hipSuccess ? DSPSR_AMD_OK : DSPSR_AMD_EHIP;
}

int fb_conv1_launch(dspsr_amd_ctx* ctx, int logM, const float* in, uint64_t chan_stride, uint64_t pol_stride, uint64_t in_step,
                    const cf* kern, const FbOut& out, uint32_t nchan, uint32_t nfilt_pos, uint32_t nkeep, uint64_t npart)
{
  kconv1_t k = pick_conv1(logM, mkseq<8>::type());
  if (!k) return DSPSR_AMD_EINVAL;
  Conv1Params p = {};
  p.in = in; p.chan_stride = chan_stride; p.pol_stride = pol_stride; p.in_step = in_step;
  p.kern = kern; p.out = out; p.nchan = nchan; p.nfilt_pos = nfilt_pos; p.nkeep = nkeep; p.npart = npart;
  const int lp = conv1_log_points(logM);
  const uint32_t Ts = (1u << (lp - logM)) / 2;
  const uint64_t tpc = (npart + Ts - 1) / Ts;
  if (tpc * nchan >= (1ull << 31)) return DSPSR_AMD_EINVAL;
  p.tiles_per_chan = (uint32_t)tpc;
  const uint64_t total = tpc * nchan;
  const size_t lds = lds_total_words_host(1u << lp, logM) * sizeof(cf);
  const uint32_t wgs = ctx->ncu * (2 * lds + 1024 <= 160 * 1024 ? 2u : 1u);
  const uint32_t grid = (uint32_t)(total < wgs ? total : wgs);
  hipLaunchKernelGGL(k, dim3(grid), dim3(1u << (lp - LOG_PTS)), lds, ctx->stream, p, ctx->tw);
  return hipGetLastError() == hipSuccess ? DSPSR_AMD_OK : DSPSR_AMD_EHIP;
}

}  // namespace dspsr_amd
