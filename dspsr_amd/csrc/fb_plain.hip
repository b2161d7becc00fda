// Non-convolving filterbank: dsp::Filterbank with freq_res = 1 (`dspsr -F N`, Filterbank::Config::After / Never).
//
// Reference: Signal/General/Filterbank.C:561-662 with the `freq_res == 1` branch :614-623 -- per input channel, part and
// polarisation a forward transform of nsamp_fft = 2 * nchan_subband real samples (frc1d, first nchan_subband bins kept) or
// nchan_subband complex samples (fcc1d), optionally Response::operate (Response.C:385-444: one factor per channel here), and
// bin k stored as THE output sample of channel k for this part (nkeep = 1, no overlap, no backward transform).  The reference's
// GPU engine does the same with plan_bwd == NULL and k_ncopy of one sample per channel (FilterbankCUDA.cu:92-116,258-304).
//
// Here: one workgroup tile = T columns x C = nchan_subband points.  Real input: the 2C real samples of one polarisation and part
// are C complex points z[n] = x[2n] + i x[2n+1] (the form k_tfp uses, tfp.hip), X[k] = A + w^k B from (Z[k], Z[C-k]) while the
// staged transform is read back; complex input: the column is the part's C samples.  Columns = (part, polarisation) pairs (two
// polarisations) or consecutive parts (one), so a tile covers T/2 or T consecutive parts and the read-back walks the PARTS with
// consecutive lanes: every store instruction writes runs of consecutive output samples of a channel row (FPT order).
// Every input form of the convolving filterbank is taken (float32 rows, generic 8-bit real / complex with any number of input
// channels, CASPSR, 16-bit UWB) through the same fetch / decode helpers (fb_common.h).
#include "fb_common.h"

namespace dspsr_amd {

struct PlainParams {
  FbGeom g;                 // (real_input, npol: what fetch_pair / decode_pair read)
  FbIn in;
  FbOut out;                // kind 0 (none), 1 (complex rows), 2 (detected)
  const cf* kern;           // [input_nchan][C] or null
  uint64_t in_chan_stride;  // float rows: floats from one input channel's rows to the next
  uint64_t npart;
  uint32_t input_nchan;
  int logT;                 // columns per tile
};

// one complex sample of polarisation `seq` at time t (complex input): up to two raw words
DEV void plain_fetch1(const FbGeom& g, const FbIn& in, const uint32_t seq, const uint64_t t, uint32_t& w0, uint32_t& w1)
{
  w0 = w1 = 0u;
  if (in.kind == 0) {
    const float* x = (const float*)in.base + seq * in.pol_stride + 2 * t;
    w0 = __float_as_uint(x[0]); w1 = __float_as_uint(x[1]);
  } else if (in.kind == 4) {                              // UWB: word (block*npol + pol)*2048 + t%2048 = (re, im) int16
    w0 = ((const uint32_t*)in.base)[((t >> 11) * g.npol + seq) * 2048 + (t & 2047)];
  } else {                                                // generic 8-bit complex: ((t*nchan+c)*npol+p)*2+d
    const uint8_t* b = (const uint8_t*)in.base + ((t * in.nchan + in.ichan) * g.npol + seq) * 2;
    if ((((uintptr_t)in.base) & 1) == 0) w0 = *(const uint16_t*)b;
    else w0 = (uint32_t)b[0] | ((uint32_t)b[1] << 8);
  }
}
DEV cf plain_decode1(const FbIn& in, const uint32_t w0, const uint32_t w1)
{
  if (in.kind == 0) return make_float2(__uint_as_float(w0), __uint_as_float(w1));
  if (in.kind == 4) return make_float2((float)(int16_t)((w0 & 0xffff) ^ 0x8000) * in.scale, (float)(int16_t)((w0 >> 16) ^ 0x8000) * in.scale);
  return make_float2(cvt8((int8_t)(w0 & 0xff), in.scale), cvt8((int8_t)((w0 >> 8) & 0xff), in.scale));
}
// real single-polarisation input: samples t, t + 1 as two raw words
DEV void plain_fetch_r1(const FbGeom& g, const FbIn& in, const uint64_t t, uint32_t& w0, uint32_t& w1)
{
  if (in.kind == 0) {
    const float* x = (const float*)in.base + t;
    w0 = __float_as_uint(x[0]); w1 = __float_as_uint(x[1]);
  } else {                                                // generic 8-bit real, one polarisation: byte t*nchan + c
    const uint8_t* b = (const uint8_t*)in.base + t * in.nchan + in.ichan;
    w0 = b[0]; w1 = b[in.nchan];
  }
}
DEV float plain_decode_r1(const FbIn& in, const uint32_t w) { return in.kind == 0 ? __uint_as_float(w) : cvt8((int8_t)(w & 0xff), in.scale); }

// MODE 0: real input, two polarisations (column pair = the two polarisations of one part)
//      1: real input, one polarisation  (column pair = two consecutive parts)
//      2: complex input                 (column pair = the two polarisations of one part, or two consecutive parts)
template <int LOGF, int MODE>
__global__ __launch_bounds__(512) void k_fb_plain(const PlainParams p, const cf* __restrict__ tw)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const uint32_t nt = blockDim.x;
  const int logT = p.logT, loghT = logT - 1;
  constexpr uint32_t C = 1u << LOGF;
  const uint32_t T = 1u << logT, hT = T >> 1;
  const bool two_pol = p.g.npol == 2;                                     // uniform
  const uint32_t tile_parts = two_pol ? hT : T;
  const uint32_t ntile = (uint32_t)((p.npart + tile_parts - 1) / tile_parts);
  const uint32_t total = ntile * p.input_nchan;
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, nt);
  // time samples (real: of the 2C-sample part; complex: of the C-sample part) between parts
  const uint64_t step = p.in.part_step;

  auto in_of = [&](const uint32_t ichan) {
    FbIn ci = p.in;
    ci.ichan = ichan;
    if (ci.kind == 0) ci.base = (const float*)p.in.base + (uint64_t)ichan * p.in_chan_stride;
    return ci;
  };
  auto fetch = [&](const uint32_t item, Raw4 (&raw)[NPAIR]) {
    const uint32_t tile = item % ntile, ichan = item / ntile;
    const FbIn ci = in_of(ichan);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        const uint32_t e = first_stage_elem<LOGF>(tid, logT, g2, i);
        const uint32_t col = e & (T - 1), n = e >> logT;
        Raw4& r = raw[(g2 / 2) * P::R1 + i];
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0u;
        if (two_pol) {
          const uint64_t part = (uint64_t)tile * hT + (col >> 1);
          if (part >= p.npart) continue;
          if constexpr (MODE == 0) r = fetch_pair<4>(p.g, ci, 0, part * step + 2ull * n);
          else if constexpr (MODE == 2) {
            plain_fetch1(p.g, ci, 0, part * step + n, r.w[0], r.w[1]);
            plain_fetch1(p.g, ci, 1, part * step + n, r.w[2], r.w[3]);
          }
        } else {
          const uint64_t pa = (uint64_t)tile * T + col;
          if constexpr (MODE == 1) {
            if (pa < p.npart) plain_fetch_r1(p.g, ci, pa * step + 2ull * n, r.w[0], r.w[1]);
            if (pa + 1 < p.npart) plain_fetch_r1(p.g, ci, (pa + 1) * step + 2ull * n, r.w[2], r.w[3]);
          } else if constexpr (MODE == 2) {
            if (pa < p.npart) plain_fetch1(p.g, ci, 0, pa * step + n, r.w[0], r.w[1]);
            if (pa + 1 < p.npart) plain_fetch1(p.g, ci, 0, (pa + 1) * step + n, r.w[2], r.w[3]);
          }
        }
      }
  };

  uint32_t item, next;
  uint32_t jrun = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, jrun, 8u, total, item)) return;
  Raw4 raw[NPAIR];
  fetch(item, raw);
  // staged transform: one plane of C float4 per column pair, (Re c0, Re c1, Im c0, Im c1) at bin k; planes one float4 further
  // apart than C so that consecutive pairs -- the lanes of a read-back -- fall on different banks (C < 16: the padding would not fit)
  float4* const stg = (float4*)lds;
  const uint32_t plane = C >= 16 ? C + 1 : C;
  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
    {
      const FbIn ci = in_of(item / ntile);
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        if constexpr (MODE == 0) {
          cf a, b;
          decode_pair<4>(p.g, ci, raw[h], a, b, 0);             // a = (x0[2n], x1[2n]), b = (x0[2n+1], x1[2n+1])
          x[h].x = (v2f){a.x, a.y};
          x[h].y = (v2f){b.x, b.y};
        } else if constexpr (MODE == 1) {
          x[h].x = (v2f){plain_decode_r1(ci, raw[h].w[0]), plain_decode_r1(ci, raw[h].w[2])};
          x[h].y = (v2f){plain_decode_r1(ci, raw[h].w[1]), plain_decode_r1(ci, raw[h].w[3])};
        } else {
          const cf a = plain_decode1(ci, raw[h].w[0], raw[h].w[1]), b = plain_decode1(ci, raw[h].w[2], raw[h].w[3]);
          x[h] = make_cx2(a, b);
        }
      }
    }
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++jrun, 8u, total, next);
    if (more) fetch(next, raw);

    auto store = [&](const uint32_t col, const uint32_t pp, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      float4* const d = stg + (col >> 1) * plane + pp;
#pragma unroll
      for (int k = 0; k < R; k++) d[k * pstride] = make_float4(v[k].x[0], v[k].x[1], v[k].y[0], v[k].y[1]);
    };
    wgfft<LOGF, -1, true>(lds, ltw_off, tid, logT, x, store);
    __syncthreads();

    // read-back: consecutive lanes = consecutive column pairs (parts) of one bin (pair)
    const uint32_t tile = item % ntile, ichan = item / ntile;
    const uint32_t chan0 = p.out.chan0 + ichan * C;
    const cf* __restrict__ kern = p.kern ? p.kern + (uint64_t)ichan * C : nullptr;
    const FbOut& out = p.out;
    // both halves of a staged pair at output channel (bin) k: half h = polarisation h of part tile*hT + j (two polarisations) or
    // part tile*T + 2j + h (one)
    auto emit = [&](const uint32_t k, const uint32_t j, cf a, cf b) {
      if (kern) { const cf w = kern[k]; a = cmul(a, w); b = cmul(b, w); }            // Response::operate, Response.C:385-444
      if (out.kind == 0) return;
      float* __restrict__ row = out.base + (uint64_t)(chan0 + k) * out.chan_stride;
      if (two_pol) {
        const uint64_t part = (uint64_t)tile * hT + j;
        if (part >= p.npart) return;
        if (out.kind == 1) {
          float2* o = (float2*)(row + part * out.part_step);
          *o = a;
          *(float2*)((float*)o + out.pol_stride) = b;
        } else {
          float q[4];
          detect4(a, b, out.state, q);
          if (out.ndim == 4) ((float4*)row)[part] = make_float4(q[0], q[1], q[2], q[3]);
          else if (out.ndim == 2) {
            ((float2*)row)[part] = make_float2(q[0], q[1]);
            ((float2*)(row + out.pol_stride))[part] = make_float2(q[2], q[3]);
          } else {
            row[part] = q[0];
            row[out.pol_stride + part] = q[1];
            row[2 * out.pol_stride + part] = q[2];
            row[3 * out.pol_stride + part] = q[3];
          }
        }
      } else {
        const uint64_t pa = (uint64_t)tile * T + 2 * j;
        if (pa < p.npart) *(float2*)(row + pa * out.part_step) = a;
        if (pa + 1 < p.npart) *(float2*)(row + (pa + 1) * out.part_step) = b;
      }
    };
    if constexpr (MODE == 2) {
      const uint32_t nitem = C << loghT;
      for (uint32_t idx = tid; idx < nitem; idx += nt) {
        const uint32_t j = idx & (hT - 1), k = idx >> loghT;
        const float4 z = stg[j * plane + k];
        emit(k, j, make_float2(z.x, z.z), make_float2(z.y, z.w));
      }
    } else {
      // X[k] = A + w^k B, X[C-k] = conj(A - w^k B);  A = (Z[k] + conj Z[C-k]) / 2, B = (Z[k] - conj Z[C-k]) / 2i, w = exp(-i pi / C)
      // both columns of the pair packed (.x = column 0, .y = column 1)
      auto split = [&](const float4 zk, const float4 zm, const float c, const float sn, v2f& xr, v2f& xi, v2f& yr, v2f& yi) {
        const v2f zr = {zk.x, zk.y}, zi = {zk.z, zk.w}, mr = {zm.x, zm.y}, mi = {zm.z, zm.w};
        const v2f ar = 0.5f * (zr + mr), ai = 0.5f * (zi - mi);
        const v2f br = 0.5f * (zi + mi), bi = 0.5f * (mr - zr);
        const v2f wr = c * br + sn * bi, wi = c * bi - sn * br;            // w^k = (c, -sn)
        xr = ar + wr; xi = ai + wi; yr = ar - wr; yi = wi - ai;
      };
      const uint32_t nitem = (C / 2) << loghT;
      for (uint32_t idx = tid; idx < nitem; idx += nt) {
        const uint32_t j = idx & (hT - 1), kp = idx >> loghT;
        v2f xr, xi, yr, yi;
        if (kp == 0) {               // bins 0 and C/2 are their own mirrors: X[0] from Z[0] (w = 1), X[C/2] from Z[C/2] (w = -i)
          const float4 z0 = stg[j * plane], zh = stg[j * plane + C / 2];
          v2f ur, ui;
          split(z0, z0, 1.0f, 0.0f, xr, xi, ur, ui);
          emit(0, j, make_float2(xr[0], xi[0]), make_float2(xr[1], xi[1]));
          split(zh, zh, 0.0f, 1.0f, yr, yi, ur, ui);
          emit(C / 2, j, make_float2(yr[0], yi[0]), make_float2(yr[1], yi[1]));
        } else {
          const float xa = (float)kp * __uint_as_float((uint32_t)(127 - (LOGF + 1)) << 23);       // k / 2C revolutions, exact
          split(stg[j * plane + kp], stg[j * plane + (C - kp)], __builtin_amdgcn_cosf(xa), __builtin_amdgcn_sinf(xa), xr, xi, yr, yi);
          emit(kp, j, make_float2(xr[0], xi[0]), make_float2(xr[1], xi[1]));
          emit(C - kp, j, make_float2(yr[0], yi[0]), make_float2(yr[1], yi[1]));
        }
      }
    }
    if (!more) break;
    item = next;
    // (the next tile's first LDS write sits behind a barrier of its own inside wgfft)
  }
}

typedef void (*kplain_t)(PlainParams, const cf*);
template <int... I> static kplain_t pick_plain(int logf, int mode, iseq<I...>)
{
  static const kplain_t m0[] = {k_fb_plain<I + 1, 0>...};
  static const kplain_t m1[] = {k_fb_plain<I + 1, 1>...};
  static const kplain_t m2[] = {k_fb_plain<I + 1, 2>...};
  if (logf < 1 || logf > (int)sizeof...(I)) return nullptr;
  return mode == 0 ? m0[logf - 1] : mode == 1 ? m1[logf - 1] : m2[logf - 1];
}

// points per workgroup tile: 2^13 (256 threads, two workgroups per compute unit) up to 1024 channels, 2^14 above (longer runs of
// output samples per channel row); at most 512 columns (the staged image must fit the exchange buffer)
static int plain_log_points(int logC) { const int lp = logC <= 10 ? 13 : 14; return lp < logC + 9 ? lp : logC + 9; }

int fb_plain_check(dspsr_amd_ctx* ctx, int logC, bool real_input, uint32_t npol, size_t* lds_bytes)
{
  const int mode = real_input ? (npol == 2 ? 0 : 1) : 2;
  kplain_t k = pick_plain(logC, mode, mkseq<MAX_LOGF>::type());
  if (!k) return DSPSR_AMD_EINVAL;
  const int lp = plain_log_points(logC);
  const size_t lds = lds_total_words_host(1u << lp, logC) * sizeof(cf);
  if (lds_bytes) *lds_bytes = lds;
  const hipError_t e = dspsr_amd_allow_lds((const void*)k, lds);
  return e == hipSuccess ? DSPSR_AMD_OK : DSPSR_AMD_EHIP;
}

int fb_plain_launch(dspsr_amd_ctx* ctx, int logC, bool real_input, uint32_t npol, uint32_t input_nchan, const cf* kern,
                    const FbIn& in, const FbOut& out, uint64_t in_chan_stride, uint64_t npart)
{
  const int mode = real_input ? (npol == 2 ? 0 : 1) : 2;
  kplain_t k = pick_plain(logC, mode, mkseq<MAX_LOGF>::type());
  if (!k) return DSPSR_AMD_EINVAL;
  const int lp = plain_log_points(logC);
  PlainParams p = {};
  p.g.real_input = real_input ? 1 : 0;
  p.g.npol = (int)npol;
  p.g.C = 1u << logC;
  p.g.nkeep = 1;
  p.in = in;
  p.out = out;
  p.kern = kern;
  p.in_chan_stride = in_chan_stride;
  p.npart = npart;
  p.input_nchan = input_nchan;
  p.logT = lp - logC;
  const uint32_t tile_parts = npol == 2 ? (1u << (p.logT - 1)) : (1u << p.logT);
  const uint64_t total = ((npart + tile_parts - 1) / tile_parts) * input_nchan;
  if (total >= (1ull << 31)) return DSPSR_AMD_EINVAL;
  const size_t lds = lds_total_words_host(1u << lp, logC) * sizeof(cf);
  const uint32_t wgs = ctx->ncu * (2 * lds + 1024 <= 160 * 1024 ? 2u : 1u);
  uint64_t grid = total < wgs ? total : wgs;
  if (grid >= 8) grid &= ~7ull;
  hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(1u << (lp - LOG_PTS)), lds, ctx->stream, p, ctx->tw);
  return hipGetLastError() == hipSuccess ? DSPSR_AMD_OK : DSPSR_AMD_EHIP;
}

}  // namespace dspsr_amd
