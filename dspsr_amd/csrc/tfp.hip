// Search-mode front end (digifil): non-convolving filterbank + square-law detection + time scrunch,
// fused into one launch.
//
// Reference chain (Signal/General/LoadToFil.C:195-315):
//   dsp::TFPFilterbank::filterbank   TFPFilterbank.C:27-101   forward FFT of nsamp_fft = 2*nchan real samples
//                                    per pol and part, power Re^2 + Im^2 of bins 0..nchan-1, TFP order,
//                                    optional pol sum (pscrunch)
//   dsp::TScrunch::tfp_tscrunch      TScrunch.C:180-206       out = in[0]; out += in[1] ... in[sfactor-1]
// Here: both polarisations of one part form one complex sequence w = x0 + i*x1 of L = 2*nchan points
// (the interleaved 8-bit bytes are w), T = 16384/L consecutive parts are the columns of one workgroup
// transform, the Hermitian split + |.|^2 is applied while reading the transform back from LDS, and each
// thread keeps the running time-scrunch sums of the bins it owns in registers (added in time order).
#include "engine_internal.h"

namespace dspsr_amd {

struct TfpParams {
  const uint8_t* raw;      // generic 8-bit order, real, 2 pols, 1 channel: byte 2*t + p
  float* out;              // [nout][nchan][npol_out]
  uint64_t npart;          // parts available
  uint32_t sfactor;        // time scrunch factor (>= 1)
  uint32_t pscrunch;       // 1: sum the two polarisations (Intensity), 0: PPQQ
  float scale;
  int logT;
  int caspsr;
};

// (pol0, pol1) bytes of sample t: issued as independent loads and combined only when the tile is decoded, so that the
// words of the NEXT tile stay in flight during the transform (combining here would wait for them at the prefetch)
template <bool CASPSR> struct TfpRaw { uint32_t w[CASPSR ? 2 : 1]; };
template <bool CASPSR> DEV TfpRaw<CASPSR> tfp_fetch(const TfpParams& p, const uint64_t t, const bool valid)
{
  TfpRaw<CASPSR> r;
  if constexpr (CASPSR) {
    const uint8_t* b = p.raw + (t >> 2) * 8 + (t & 3);
    r.w[0] = valid ? b[0] : 0x80u;
    r.w[1] = valid ? b[4] : 0x80u;
  } else {
    r.w[0] = valid ? *(const uint16_t*)(p.raw + 2 * t) : 0x8080u;
  }
  return r;
}
template <bool CASPSR> DEV uint32_t tfp_pair(const TfpRaw<CASPSR>& r)
{
  if constexpr (CASPSR) return (r.w[0] & 0xffu) | ((r.w[1] & 0xffu) << 8);
  else return r.w[0] & 0xffffu;
}

// COAL: the T parts of a group are ONE contiguous range of T*L*2 bytes.  Loading it in first-stage order means 32 two-byte
// loads per thread (128 bytes per wave instruction: 13.7 us per 16384-point group, the pass was bound by the issue of its
// loads); instead every thread loads 16-byte pieces of the range (prefetched one group ahead, 4 registers per piece), the
// bytes go through the -- at that moment idle -- exchange buffer, and the thread picks its 32 samples from LDS.
// Needs a 16-byte aligned block (the host checks; otherwise the element-wise kernel runs).
template <int LOGF, bool CASPSR, bool COAL>
__global__ __launch_bounds__(512) void k_tfp(const TfpParams p, const cf* __restrict__ tw)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const uint32_t nt = blockDim.x;
  const int logT = p.logT;
  const uint32_t T = 1u << logT, L = 1u << LOGF, nchan = L >> 1;
  const uint32_t npol_out = p.pscrunch ? 1 : 2;
  // one item = the parts of one output sample, rounded up to whole column groups
  const uint64_t nout = p.npart / p.sfactor;
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, nt);
  constexpr int NB = (LOGF - 1 > 9) ? (1 << (LOGF - 1 - 9)) : 1;   // bins per thread at 512 threads

  // work items: groups of T consecutive parts, dealt in order so that a workgroup owns whole output samples
  // when sfactor >= T, or several whole output samples when sfactor < T (host guarantees sfactor % T == 0
  // or T % sfactor == 0)
  const uint32_t groups_per_out = p.sfactor > T ? p.sfactor >> logT : 1;
  const uint64_t nitem = p.sfactor > T ? nout : (nout * p.sfactor + T - 1) >> logT;

  auto fetch = [&](const uint64_t group, TfpRaw<CASPSR> (&ra)[NPAIR], TfpRaw<CASPSR> (&rb)[NPAIR]) {
    const uint64_t part0 = group << logT;
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        const uint32_t e = first_stage_elem<LOGF>(tid, logT, g2, i);
        const uint32_t col = e & (T - 1), n = e >> logT;
        const uint64_t pa = part0 + col, pb = pa + 1;
        ra[(g2 / 2) * P::R1 + i] = tfp_fetch<CASPSR>(p, pa * L + n, pa < p.npart);
        rb[(g2 / 2) * P::R1 + i] = tfp_fetch<CASPSR>(p, pb * L + n, pb < p.npart);
      }
  };

  // the 8-bit samples of the next group are requested while the current one is transformed (register prefetch,
  // as in the filterbank passes); past the end the loads are skipped by the part bound inside fetch()
  TfpRaw<CASPSR> ra[NPAIR], rb[NPAIR];
  constexpr uint32_t NCH = (PTS * 2) / 16;                 // 16-byte pieces per thread: 32 points x 2 bytes
  uint4 piece[NCH];
  const uint64_t raw_bytes = p.npart * (uint64_t)L * 2;
  auto fetch_pieces = [&](const uint64_t group) {
    const uint64_t base = (group << logT) * (uint64_t)L * 2;
#pragma unroll
    for (uint32_t r = 0; r < NCH; r++) {
      const uint64_t off = base + 16ull * (tid + r * nt);
      piece[r] = off + 16 <= raw_bytes ? *(const uint4*)(p.raw + off) : make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    }
  };
  if (blockIdx.x < nitem) {
    if constexpr (COAL) fetch_pieces((uint64_t)blockIdx.x * groups_per_out);
    else fetch((uint64_t)blockIdx.x * groups_per_out, ra, rb);
  }
  for (uint64_t item = blockIdx.x; item < nitem; item += gridDim.x) {
    float acc[NB][2];
#pragma unroll
    for (int j = 0; j < NB; j++) acc[j][0] = acc[j][1] = 0.f;
    for (uint32_t gi = 0; gi < groups_per_out; gi++) {
      const uint64_t group = item * groups_per_out + gi;
      asm volatile("" : "+v"(tid));
      cx2 x[NPAIR];
      if constexpr (COAL) {
        // the group's bytes, in file order, into the exchange buffer (the previous group's read-back ended with a barrier)
        uint8_t* img = (uint8_t*)lds;
#pragma unroll
        for (uint32_t r = 0; r < NCH; r++) *(uint4*)(img + 16u * (tid + r * nt)) = piece[r];
        __syncthreads();
#pragma unroll
        for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
          for (int i = 0; i < P::R1; i++) {
            const uint32_t e = first_stage_elem<LOGF>(tid, logT, g2, i);
            const uint32_t col = e & (T - 1), n = e >> logT;
            const uint32_t sa = col * L + n, sb = sa + L;           // columns col and col + 1 (T >= 2)
            uint32_t wa, wb;
            if constexpr (CASPSR) {
              const uint8_t* a = img + (sa >> 2) * 8 + (sa & 3);
              const uint8_t* b = img + (sb >> 2) * 8 + (sb & 3);
              wa = (uint32_t)a[0] | ((uint32_t)a[4] << 8);
              wb = (uint32_t)b[0] | ((uint32_t)b[4] << 8);
            } else {
              wa = *(const uint16_t*)(img + 2 * sa);
              wb = *(const uint16_t*)(img + 2 * sb);
            }
            ra[(g2 / 2) * P::R1 + i].w[0] = wa & 0xffu;
            rb[(g2 / 2) * P::R1 + i].w[0] = wb & 0xffu;
            if constexpr (CASPSR) {
              ra[(g2 / 2) * P::R1 + i].w[1] = wa >> 8;
              rb[(g2 / 2) * P::R1 + i].w[1] = wb >> 8;
            } else {
              ra[(g2 / 2) * P::R1 + i].w[0] = wa;
              rb[(g2 / 2) * P::R1 + i].w[0] = wb;
            }
          }
      }
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        const uint32_t w = tfp_pair<CASPSR>(ra[h]) | (tfp_pair<CASPSR>(rb[h]) << 16);
        x[h] = make_cx2(make_float2(((float)(int8_t)(w & 0xff) + 0.5f) * p.scale, ((float)(int8_t)((w >> 8) & 0xff) + 0.5f) * p.scale),
                        make_float2(((float)(int8_t)((w >> 16) & 0xff) + 0.5f) * p.scale, ((float)(int8_t)(w >> 24) + 0.5f) * p.scale));
      }
      {
        const uint64_t next = gi + 1 < groups_per_out ? group + 1 : (item + gridDim.x) * groups_per_out;
        if (gi + 1 < groups_per_out || item + gridDim.x < nitem) {
          if constexpr (COAL) fetch_pieces(next);
          else fetch(next, ra, rb);
        }
      }
      if constexpr (COAL) __syncthreads();      // every thread has taken its samples: the exchanges may overwrite the image
      auto store = [&](const uint32_t col, const uint32_t pp, const uint32_t pstride, auto& v) {
        constexpr int R = sizeof(v) / sizeof(v[0]);
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t pos = k * pstride + pp;
          *(float4*)&lds[lds_pad((pos << logT) | col)] = make_float4(v[k].x[0], v[k].y[0], v[k].x[1], v[k].y[1]);
        }
      };
      wgfft<LOGF, -1, true>(lds, ltw_off, tid, logT, x, store);
      __syncthreads();
      // Hermitian split, power, time scrunch (columns = consecutive parts, added in time order)
      const uint64_t part_first = group << logT, part_end = nout * p.sfactor;
      const uint32_t phase_first = (uint32_t)(part_first % p.sfactor);       // wave-uniform: no division per bin
      const uint64_t out_first = part_first / p.sfactor;
      // Whole group inside one output sample (tscrunch >= T, all parts present): no per-column bookkeeping, the columns of
      // a bin are read two at a time (16 bytes).  0 + p0 == p0, so starting an output sample from zero gives the sums of
      // TScrunch.C:193-200 bit for bit.
      const bool whole = p.sfactor >= T && T >= 2 && part_first + T <= part_end;
      if (whole) {
        const bool emit = phase_first + T == p.sfactor;
#pragma unroll
        for (int j = 0; j < NB; j++) {
          const uint32_t k = tid + j * nt;
          if (k < nchan) {
            const uint32_t km = (L - k) & (L - 1);
            float s0 = phase_first == 0 ? 0.f : acc[j][0], s1 = phase_first == 0 ? 0.f : acc[j][1];
            const uint32_t ia = lds_pad(k << logT), ib = lds_pad(km << logT);
            for (uint32_t c = 0; c < T; c += 2) {
              const uint32_t ca = c + ((c >> 6) << 2);          // lds_pad(e0 + c) for e0 a multiple of 64 (T >= 64) or c < 64
              const float4 a2 = *(const float4*)&lds[T >= 64 ? ia + ca : ia + c];
              const float4 b2 = *(const float4*)&lds[T >= 64 ? ib + ca : ib + c];
#pragma unroll
              for (int h = 0; h < 2; h++) {
                const float ax = h ? a2.z : a2.x, ay = h ? a2.w : a2.y, bx = h ? b2.z : b2.x, by = h ? b2.w : b2.y;
                const float x0r = 0.5f * (ax + bx), x0i = 0.5f * (ay - by);
                const float x1r = 0.5f * (ay + by), x1i = 0.5f * (bx - ax);
                float p0 = x0r * x0r; p0 += x0i * x0i;            // TFPFilterbank.C:56-59
                float p1 = x1r * x1r; p1 += x1i * x1i;
                if (p.pscrunch) { p0 += p1; p1 = 0.f; }
                s0 += p0; s1 += p1;
              }
            }
            acc[j][0] = s0; acc[j][1] = s1;
            if (emit) {
              float* o = p.out + (out_first * nchan + k) * npol_out;
              o[0] = s0;
              if (!p.pscrunch) o[1] = s1;
            }
          }
        }
      } else
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const uint32_t k = tid + j * nt;
        if (k < nchan) {
          const uint32_t km = (L - k) & (L - 1);
          uint32_t phase = phase_first;
          uint64_t oidx = out_first;
          for (uint32_t c = 0; c < T; c++) {
            const uint64_t part = part_first + c;
            if (part >= part_end) break;
            const cf a = lds[lds_pad((k << logT) | c)], b = lds[lds_pad((km << logT) | c)];
            const float x0r = 0.5f * (a.x + b.x), x0i = 0.5f * (a.y - b.y);
            const float x1r = 0.5f * (a.y + b.y), x1i = 0.5f * (b.x - a.x);
            float p0 = x0r * x0r; p0 += x0i * x0i;            // TFPFilterbank.C:56-59
            float p1 = x1r * x1r; p1 += x1i * x1i;
            if (p.pscrunch) { p0 += p1; p1 = 0.f; }                 // TFPFilterbank.C:79-80: pol sum BEFORE the time sum
            if (phase == 0) { acc[j][0] = p0; acc[j][1] = p1; }     // TScrunch.C:193-194
            else { acc[j][0] += p0; acc[j][1] += p1; }               // TScrunch.C:199-200
            if (++phase == p.sfactor) {
              float* o = p.out + (oidx * nchan + k) * npol_out;
              o[0] = acc[j][0];
              if (!p.pscrunch) o[1] = acc[j][1];
              phase = 0;
              oidx++;
            }
          }
        }
      }
      __syncthreads();    // LDS is overwritten by the next group's exchanges
    }
  }
}

typedef void (*ktfp_t)(TfpParams, const cf*);
template <int... I> struct iseq_t {};
template <int N, int... I> struct mkseq_t : mkseq_t<N - 1, N - 1, I...> {};
template <int... I> struct mkseq_t<0, I...> { typedef iseq_t<I...> type; };
template <int... I> static ktfp_t pick_tfp(int logf, bool caspsr, bool coal, iseq_t<I...>)
{
  static const ktfp_t t[] = {k_tfp<I, false, false>...};
  static const ktfp_t c[] = {k_tfp<I, true, false>...};
  static const ktfp_t tc[] = {k_tfp<I, false, true>...};
  static const ktfp_t cc[] = {k_tfp<I, true, true>...};
  return coal ? (caspsr ? cc[logf] : tc[logf]) : (caspsr ? c[logf] : t[logf]);
}

}  // namespace dspsr_amd

using namespace dspsr_amd;

extern "C" int dspsr_amd_tfp_filterbank(dspsr_amd_ctx* ctx, const dspsr_amd_tfp_config* cfg, const int8_t* raw_dev,
                                        int raw_layout, float scale, float* out_dev, uint64_t npart)
{
  if (!ctx || !cfg || !raw_dev || !out_dev) return DSPSR_AMD_EINVAL;
  const uint32_t nchan = cfg->nchan;
  if (nchan < 16 || (nchan & (nchan - 1)) || nchan > 4096)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tfp_filterbank: nchan=%u must be a power of two in [16, 4096]", nchan);
  if (cfg->npol != 2)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tfp_filterbank: only real dual-polarisation 8-bit input is built (npol=%u)", cfg->npol);
  const uint32_t sf = cfg->tscrunch ? cfg->tscrunch : 1;
  int logF = 0;
  while ((1u << logF) < 2 * nchan) logF++;
  const int logT = 14 - logF;
  const uint32_t T = 1u << logT;
  if (!((sf % T) == 0 || (T % sf) == 0))
    return ctx_fail(ctx, DSPSR_AMD_EINVAL,
                    "dspsr_amd_tfp_filterbank: tscrunch=%u must divide or be a multiple of %u parts per workgroup", sf, T);
  if (raw_layout != DSPSR_AMD_RAW_GENERIC && raw_layout != DSPSR_AMD_RAW_CASPSR)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tfp_filterbank: unknown raw layout %d", raw_layout);
  if (raw_layout == DSPSR_AMD_RAW_GENERIC && ((uintptr_t)raw_dev & 1))
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tfp_filterbank: raw pointer must be 2-byte aligned");
  const uint64_t nout = npart / sf;
  if (nout == 0) return DSPSR_AMD_OK;
  TfpParams p;
  p.raw = (const uint8_t*)raw_dev; p.out = out_dev; p.npart = npart; p.sfactor = sf; p.pscrunch = cfg->pscrunch ? 1 : 0;
  p.scale = scale; p.logT = logT; p.caspsr = raw_layout == DSPSR_AMD_RAW_CASPSR;
  // whole-range 16-byte loads need an aligned block, at least two columns per group (T >= 2) and full-size workgroups
  const bool coal = ((uintptr_t)raw_dev & 15) == 0 && logT >= 1;
  ktfp_t k = pick_tfp(logF, p.caspsr != 0, coal, mkseq_t<14>::type());
  const size_t lds = lds_total_words_host(16384, logF) * sizeof(cf);
  hipError_t e = dspsr_amd_allow_lds((const void*)k, lds);      // raised once per kernel, not per call
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_tfp_filterbank: %s", hipGetErrorString(e));
  const uint64_t nitem = sf > T ? nout : (nout * sf + T - 1) >> logT;
  const uint32_t ncu = ctx->ncu;
  const uint32_t grid = (uint32_t)(nitem < ncu ? nitem : ncu);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, ctx->stream, p, ctx->tw);
  e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_tfp_filterbank: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}
