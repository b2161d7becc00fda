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
// channels, CASPSR, 16-bit UWB); the form is decided once per kernel (see F_WORD ... F_HALF below).
#include "fb_common.h"

namespace dspsr_amd {

struct PlainParams {
  FbGeom g;                 // (real_input, npol)
  FbIn in;
  FbOut out;                // kind 0 (none), 1 (complex rows), 2 (detected)
  const cf* kern;           // [input_nchan][C] or null
  uint64_t in_chan_stride;  // float rows: floats from one input channel's rows to the next
  uint64_t npart;
  uint32_t input_nchan;
};

// Input forms, decided ONCE per kernel (uniform) so that the loads of a tile are one straight-line burst: with the form looked up
// per element (fetch_pair's chain of tests) every load sat behind a branch and a `s_waitcnt vmcnt(0)` -- sixteen dependent
// round trips per tile, 13 us per tile where the transform needs 3 (profiles/r05_experiments.txt item 8).
//   real, two polarisations (MODE 0):  F_WORD   generic 8-bit, one input channel, 4-byte aligned block: (p0, p1)[t], (p0, p1)[t+1] = one word
//                                      F_CASPSR 4 B pol0 | 4 B pol1: two half words
//                                      F_FLOAT  float32 rows
//                                      F_BYTES  generic 8-bit, any channel count / alignment: four bytes
//   real, one polarisation (MODE 1):   F_FLOAT, F_BYTES
//   complex (MODE 2):                  F_FLOAT (re, im) pairs, F_UWB 16-bit offset binary, F_HALF generic 8-bit (re, im) as one half word
//                                      (2-byte aligned block), F_BYTES
enum { F_WORD = 0, F_CASPSR, F_FLOAT, F_BYTES, F_UWB, F_HALF };

// points per workgroup tile: 2^13 (256 threads, two workgroups per compute unit) up to 256 channels, 2^14 above (runs of 128 bytes
// per channel row at 512 channels, 64 at 1024: measured 2.26 -> see profiles/r05_experiments.txt item 8); at most 512 columns (the staged image must fit the exchange buffer).  A function of the channel
// count alone, so the column count is a compile-time constant of every instantiation (LDS addresses fold into immediates).
constexpr int plain_log_points(int logC, int mode = 0)
{
  (void)mode;       // (2^13-point tiles of one polarisation at 1024 channels, two workgroups per compute unit: 2.55 against 1.95 ms)
  const int lp = logC <= 8 ? 13 : 14;
  return lp < logC + 9 ? lp : logC + 9;
}

// MODE 0: real input, two polarisations (column pair = the two polarisations of one part)
//      1: real input, one polarisation  (column pair = two consecutive parts)
//      2: complex input                 (column pair = the two polarisations of one part, or two consecutive parts)
//      3: real input, two polarisations in the data, ONE per tile (column pair = two consecutive parts of that polarisation; the
//         tile of the other polarisation is the next work item): twice the parts per tile, i.e. twice the bytes a tile writes to a
//         channel row -- for channel counts whose rows would otherwise receive less than a cache line per tile (complex rows only:
//         Detection needs both polarisations in one thread)
template <int LOGF, int MODE>
__global__ __launch_bounds__(512) void k_fb_plain(const PlainParams p, const cf* __restrict__ tw)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  constexpr int logT = plain_log_points(LOGF, MODE) - LOGF, loghT = logT - 1;
  constexpr uint32_t nt = 1u << (plain_log_points(LOGF, MODE) - LOG_PTS);
  constexpr uint32_t C = 1u << LOGF;
  constexpr uint32_t T = 1u << logT, hT = T >> 1;
  const bool two_pol = MODE != 3 && p.g.npol == 2;                        // uniform: both polarisations in a tile
  constexpr uint32_t NPS = MODE == 3 ? 2u : 1u;                           // polarisation tiles per time range
  const uint32_t tile_parts = two_pol ? hT : T;
  const uint32_t ntile = (uint32_t)((p.npart + tile_parts - 1) / tile_parts);
  const uint32_t total = ntile * p.input_nchan * NPS;
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, nt);
  // time samples (real: of the 2C-sample part; complex: of the C-sample part) between parts
  const uint64_t step = p.in.part_step;

  // the input form (uniform)
  int form;
  if (MODE == 0 || MODE == 3) form = p.in.kind == 0 ? F_FLOAT : p.in.kind == 2 ? F_CASPSR : (p.in.nchan == 1 && (((uintptr_t)p.in.base) & 3) == 0) ? F_WORD : F_BYTES;
  else if (MODE == 1) form = p.in.kind == 0 ? F_FLOAT : F_BYTES;
  else form = p.in.kind == 0 ? F_FLOAT : p.in.kind == 4 ? F_UWB : (((uintptr_t)p.in.base) & 1) == 0 ? F_HALF : F_BYTES;
  const uint64_t last_part = p.npart - 1;
  // columns of element (g2, i) of this thread: pair index, position; the parts of a ragged last tile are clamped to the last part
  // (loaded again, never stored), so no load is conditional
  auto fetch = [&](const uint32_t item, Raw4 (&raw)[NPAIR]) {
    const uint32_t pol = item % NPS, tile = (item / NPS) % ntile, ichan = item / (NPS * ntile);
    // first input sample of the two columns of pair h (computed where it is used: kept in arrays, the sixteen 64-bit pairs spilled)
    auto samp = [&](const int h, uint64_t& sa, uint64_t& sb) {
      const int g2 = 2 * (h / P::R1), i = h % P::R1;
      const uint32_t e = first_stage_elem<LOGF>(tid, logT, g2, i);
      const uint32_t col = e & (T - 1), n = e >> logT;
      const uint32_t nn = MODE == 2 ? n : 2 * n;
      if (two_pol) {
        uint64_t part = (uint64_t)tile * hT + (col >> 1);
        part = part < last_part ? part : last_part;
        sa = sb = part * step + nn;
      } else {
        const uint64_t part = (uint64_t)tile * T + col;
        const uint64_t p0 = part < last_part ? part : last_part, p1 = part + 1 < last_part ? part + 1 : last_part;
        sa = p0 * step + nn;
        sb = p1 * step + nn;
      }
    };
    if constexpr (MODE == 0) {
      if (form == F_WORD) {
        const uint8_t* b = (const uint8_t*)p.in.base;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) { uint64_t sa, sb; samp(h, sa, sb); (void)sb; raw[h].w[0] = *(const uint32_t*)(b + 2 * sa); }
      } else if (form == F_CASPSR) {
        const uint8_t* b = (const uint8_t*)p.in.base;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          const uint8_t* q = b + (sa >> 2) * 8 + (sa & 3);
          raw[h].w[0] = *(const uint16_t*)q;
          raw[h].w[1] = *(const uint16_t*)(q + 4);
        }
      } else if (form == F_FLOAT) {
        const float* x = (const float*)p.in.base + (uint64_t)ichan * p.in_chan_stride;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          const float* q = x + sa;
          raw[h].w[0] = __float_as_uint(q[0]); raw[h].w[1] = __float_as_uint(q[1]);
          raw[h].w[2] = __float_as_uint(q[p.in.pol_stride]); raw[h].w[3] = __float_as_uint(q[p.in.pol_stride + 1]);
        }
      } else {
        const uint64_t skip = (uint64_t)p.in.nchan * 2;
        const uint8_t* b = (const uint8_t*)p.in.base + (uint64_t)ichan * 2;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          const uint8_t* q = b + sa * skip;
          raw[h].w[0] = q[0]; raw[h].w[1] = q[1]; raw[h].w[2] = q[skip]; raw[h].w[3] = q[skip + 1];
        }
      }
    } else if constexpr (MODE == 1 || MODE == 3) {
      // one polarisation per tile: MODE 1 the data's only one, MODE 3 polarisation `pol` of two
      if (form == F_FLOAT) {
        const float* x = (const float*)p.in.base + (uint64_t)ichan * p.in_chan_stride + (uint64_t)pol * p.in.pol_stride;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          raw[h].w[0] = __float_as_uint(x[sa]); raw[h].w[1] = __float_as_uint(x[sa + 1]);
          raw[h].w[2] = __float_as_uint(x[sb]); raw[h].w[3] = __float_as_uint(x[sb + 1]);
        }
      } else if (MODE == 3 && form == F_WORD) {             // (p0[2n], p1[2n], p0[2n+1], p1[2n+1]) of the pair's two parts
        const uint8_t* b = (const uint8_t*)p.in.base;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          raw[h].w[0] = *(const uint32_t*)(b + 2 * sa);
          raw[h].w[1] = *(const uint32_t*)(b + 2 * sb);
        }
      } else if (MODE == 3 && form == F_CASPSR) {           // (x[2n], x[2n+1]) of this polarisation: one half word per part
        const uint8_t* b = (const uint8_t*)p.in.base + 4 * pol;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          raw[h].w[0] = *(const uint16_t*)(b + (sa >> 2) * 8 + (sa & 3));
          raw[h].w[1] = *(const uint16_t*)(b + (sb >> 2) * 8 + (sb & 3));
        }
      } else {
        const uint64_t skip = (uint64_t)p.in.nchan * NPS;
        const uint8_t* b = (const uint8_t*)p.in.base + (uint64_t)ichan * NPS + pol;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          raw[h].w[0] = b[sa * skip]; raw[h].w[1] = b[(sa + 1) * skip];
          raw[h].w[2] = b[sb * skip]; raw[h].w[3] = b[(sb + 1) * skip];
        }
      }
    } else {
      // complex input: the pair's columns are (pol 0, pol 1) of one part, or pol 0 of two parts
      const uint32_t spol = two_pol ? 1u : 0u;
      if (form == F_FLOAT) {
        const float* x = (const float*)p.in.base + (uint64_t)ichan * p.in_chan_stride;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          const float* qa = x + 2 * sa;
          const float* qb = x + spol * p.in.pol_stride + 2 * sb;
          raw[h].w[0] = __float_as_uint(qa[0]); raw[h].w[1] = __float_as_uint(qa[1]);
          raw[h].w[2] = __float_as_uint(qb[0]); raw[h].w[3] = __float_as_uint(qb[1]);
        }
      } else if (form == F_UWB) {                          // word (block * npol + pol) * 2048 + t % 2048 = (re, im) int16
        const uint32_t* b = (const uint32_t*)p.in.base;
        const uint32_t np = (uint32_t)p.g.npol;
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
          raw[h].w[0] = b[((sa >> 11) * np) * 2048 + (sa & 2047)];
          raw[h].w[2] = b[((sb >> 11) * np + spol) * 2048 + (sb & 2047)];
        }
      } else {                                              // generic 8-bit complex: ((t * nchan + c) * npol + p) * 2 + d
        const uint64_t skip = (uint64_t)p.in.nchan * p.g.npol * 2;
        const uint8_t* b = (const uint8_t*)p.in.base + (uint64_t)ichan * p.g.npol * 2;
        if (form == F_HALF) {
#pragma unroll
          for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
            raw[h].w[0] = *(const uint16_t*)(b + sa * skip);
            raw[h].w[2] = *(const uint16_t*)(b + sb * skip + 2 * spol);
          }
        } else {
#pragma unroll
          for (int h = 0; h < NPAIR; h++) {
          uint64_t sa, sb; samp(h, sa, sb);
            const uint8_t* qa = b + sa * skip;
            const uint8_t* qb = b + sb * skip + 2 * spol;
            raw[h].w[0] = (uint32_t)qa[0] | ((uint32_t)qa[1] << 8);
            raw[h].w[2] = (uint32_t)qb[0] | ((uint32_t)qb[1] << 8);
          }
        }
      }
    }
  };
  const float scale = p.in.scale;
  auto decode = [&](const Raw4 (&raw)[NPAIR], cx2 (&x)[NPAIR], [[maybe_unused]] const uint32_t pol) {
    if constexpr (MODE == 0) {
      if (form == F_WORD) {
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          const uint32_t w = raw[h].w[0];                     // (p0[2n], p1[2n], p0[2n+1], p1[2n+1]) = (Re z0, Re z1, Im z0, Im z1)
          x[h].x = (v2f){cvt8((int8_t)(w & 0xff), scale), cvt8((int8_t)((w >> 8) & 0xff), scale)};
          x[h].y = (v2f){cvt8((int8_t)((w >> 16) & 0xff), scale), cvt8((int8_t)(w >> 24), scale)};
        }
      } else if (form == F_CASPSR) {
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          const uint32_t w0 = raw[h].w[0], w1 = raw[h].w[1];  // (p0[2n], p0[2n+1]), (p1[2n], p1[2n+1])
          x[h].x = (v2f){cvt8((int8_t)(w0 & 0xff), scale), cvt8((int8_t)(w1 & 0xff), scale)};
          x[h].y = (v2f){cvt8((int8_t)((w0 >> 8) & 0xff), scale), cvt8((int8_t)((w1 >> 8) & 0xff), scale)};
        }
      } else if (form == F_FLOAT) {
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {
          x[h].x = (v2f){__uint_as_float(raw[h].w[0]), __uint_as_float(raw[h].w[2])};
          x[h].y = (v2f){__uint_as_float(raw[h].w[1]), __uint_as_float(raw[h].w[3])};
        }
      } else {
#pragma unroll
        for (int h = 0; h < NPAIR; h++) {                     // bytes (p0[2n], p1[2n], p0[2n+1], p1[2n+1])
          x[h].x = (v2f){cvt8((int8_t)raw[h].w[0], scale), cvt8((int8_t)raw[h].w[1], scale)};
          x[h].y = (v2f){cvt8((int8_t)raw[h].w[2], scale), cvt8((int8_t)raw[h].w[3], scale)};
        }
      }
    } else if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {                       // (xa[2n], xa[2n+1], xb[2n], xb[2n+1]) of parts a, b
        if (form == F_FLOAT) {
          x[h].x = (v2f){__uint_as_float(raw[h].w[0]), __uint_as_float(raw[h].w[2])};
          x[h].y = (v2f){__uint_as_float(raw[h].w[1]), __uint_as_float(raw[h].w[3])};
        } else if (MODE == 3 && form == F_WORD) {             // byte `pol` = x[2n], byte 2 + `pol` = x[2n+1] of each part's word
          const uint32_t wa = raw[h].w[0] >> (8 * pol), wb = raw[h].w[1] >> (8 * pol);
          x[h].x = (v2f){cvt8((int8_t)(wa & 0xff), scale), cvt8((int8_t)(wb & 0xff), scale)};
          x[h].y = (v2f){cvt8((int8_t)((wa >> 16) & 0xff), scale), cvt8((int8_t)((wb >> 16) & 0xff), scale)};
        } else if (MODE == 3 && form == F_CASPSR) {
          const uint32_t wa = raw[h].w[0], wb = raw[h].w[1];
          x[h].x = (v2f){cvt8((int8_t)(wa & 0xff), scale), cvt8((int8_t)(wb & 0xff), scale)};
          x[h].y = (v2f){cvt8((int8_t)((wa >> 8) & 0xff), scale), cvt8((int8_t)((wb >> 8) & 0xff), scale)};
        } else {
          x[h].x = (v2f){cvt8((int8_t)raw[h].w[0], scale), cvt8((int8_t)raw[h].w[2], scale)};
          x[h].y = (v2f){cvt8((int8_t)raw[h].w[1], scale), cvt8((int8_t)raw[h].w[3], scale)};
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        cf a, b;
        if (form == F_FLOAT) {
          a = make_float2(__uint_as_float(raw[h].w[0]), __uint_as_float(raw[h].w[1]));
          b = make_float2(__uint_as_float(raw[h].w[2]), __uint_as_float(raw[h].w[3]));
        } else if (form == F_UWB) {                           // convert_offset_binary, UWBUnpackerCUDA.cu:24
          const uint32_t wa = raw[h].w[0], wb = raw[h].w[2];
          a = make_float2((float)(int16_t)((wa & 0xffff) ^ 0x8000) * scale, (float)(int16_t)((wa >> 16) ^ 0x8000) * scale);
          b = make_float2((float)(int16_t)((wb & 0xffff) ^ 0x8000) * scale, (float)(int16_t)((wb >> 16) ^ 0x8000) * scale);
        } else {
          const uint32_t wa = raw[h].w[0], wb = raw[h].w[2];
          a = make_float2(cvt8((int8_t)(wa & 0xff), scale), cvt8((int8_t)((wa >> 8) & 0xff), scale));
          b = make_float2(cvt8((int8_t)(wb & 0xff), scale), cvt8((int8_t)((wb >> 8) & 0xff), scale));
        }
        x[h] = make_cx2(a, b);
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
    decode(raw, x, item % NPS);
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++jrun, 8u, total, next);
    fetch(more ? next : item, raw);      // unconditional: a conditional prefetch is waited for where it is issued (fb_inv_chan.h)

    auto store = [&](const uint32_t col, const uint32_t pp, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      float4* const d = stg + (col >> 1) * plane + pp;
#pragma unroll
      for (int k = 0; k < R; k++) d[k * pstride] = make_float4(v[k].x[0], v[k].x[1], v[k].y[0], v[k].y[1]);
    };
    wgfft<LOGF, -1, true>(lds, ltw_off, tid, logT, x, store);
    __syncthreads();

    // read-back: consecutive lanes = consecutive column pairs (parts) of one bin (pair)
    const uint32_t pol = item % NPS, tile = (item / NPS) % ntile, ichan = item / (NPS * ntile);
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
        float* __restrict__ prow = row + (uint64_t)pol * out.pol_stride;                   // (MODE 3: this tile's polarisation)
        if (pa < p.npart) *(float2*)(prow + pa * out.part_step) = a;
        if (pa + 1 < p.npart) *(float2*)(prow + (pa + 1) * out.part_step) = b;
      }
    };
    if constexpr (MODE == 2) {
      constexpr int NIT = (int)((C << loghT) / nt);                       // 16 bins of a pair per thread, eight LDS reads at a time
#pragma unroll
      for (int i0 = 0; i0 < NIT; i0 += 8) {
        float4 z[8];
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const uint32_t idx = tid + (i0 + it) * nt;
          z[it] = stg[(idx & (hT - 1)) * plane + (idx >> loghT)];
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const uint32_t idx = tid + (i0 + it) * nt;
          emit(idx >> loghT, idx & (hT - 1), make_float2(z[it].x, z[it].z), make_float2(z[it].y, z[it].w));
        }
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
      // (C/2 * hT bin pairs on nt = C * T / 32 threads: eight per thread -- eight LDS reads, then the arithmetic of four pairs, twice:
      //  all sixteen reads at once spilled 64 bytes per lane)
      constexpr int NIT = (int)(((C / 2) << loghT) / nt);
      static_assert(NIT == 8, "eight bin pairs per thread");
#pragma unroll
      for (int i0 = 0; i0 < NIT; i0 += 4) {
      float4 zk[4], zm[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t idx = tid + (i0 + q) * nt, j = idx & (hT - 1), kp = idx >> loghT;
        zk[q] = stg[j * plane + kp];
        zm[q] = stg[j * plane + (kp ? C - kp : C / 2)];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int it = q;
        const uint32_t idx = tid + (i0 + q) * nt, j = idx & (hT - 1), kp = idx >> loghT;
        v2f xr, xi, yr, yi;
        if (kp == 0) {               // bins 0 and C/2 are their own mirrors: X[0] from Z[0] (w = 1), X[C/2] from Z[C/2] (w = -i)
          v2f ur, ui;
          split(zk[it], zk[it], 1.0f, 0.0f, xr, xi, ur, ui);
          emit(0, j, make_float2(xr[0], xi[0]), make_float2(xr[1], xi[1]));
          split(zm[it], zm[it], 0.0f, 1.0f, yr, yi, ur, ui);
          emit(C / 2, j, make_float2(yr[0], yi[0]), make_float2(yr[1], yi[1]));
        } else {
          const float xa = (float)kp * __uint_as_float((uint32_t)(127 - (LOGF + 1)) << 23);       // k / 2C revolutions, exact
          split(zk[it], zm[it], __builtin_amdgcn_cosf(xa), __builtin_amdgcn_sinf(xa), xr, xi, yr, yi);
          emit(kp, j, make_float2(xr[0], xi[0]), make_float2(xr[1], xi[1]));
          emit(C - kp, j, make_float2(yr[0], yi[0]), make_float2(yr[1], yi[1]));
        }
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
  static const kplain_t m3[] = {k_fb_plain<I + 1, 3>...};
  if (logf < 1 || logf > (int)sizeof...(I)) return nullptr;
  return mode == 0 ? m0[logf - 1] : mode == 1 ? m1[logf - 1] : mode == 2 ? m2[logf - 1] : m3[logf - 1];
}

// Polarisation-split tiles (MODE 3) where a two-polarisation tile gives a channel row 32 bytes or less: from 2048 channels on
// (2^14-point tiles hold 4 parts of two polarisations there, 8 parts of one), complex rows only.  Measured per 2^29 samples, two-
// polarisation / split tiles: 1024 channels 1.95 / 2.00 ms (64 against 128 bytes per row: not what bounds it), 2048 channels 3.34 / 2.59,
// 8192 channels 11.9 / 8.3 (profiles/r05_experiments.txt item 8)
static bool plain_pol_split(int logC, bool real_input, uint32_t npol, int out_kind)
{
  return real_input && npol == 2 && logC >= 11 && (out_kind == 0 || out_kind == 1);
}

int fb_plain_check(dspsr_amd_ctx* ctx, int logC, bool real_input, uint32_t npol, size_t* lds_bytes)
{
  const int mode = real_input ? (npol == 2 ? 0 : 1) : 2;
  kplain_t k = pick_plain(logC, mode, mkseq<MAX_LOGF>::type());
  if (!k) return DSPSR_AMD_EINVAL;
  const int lp = plain_log_points(logC);
  const size_t lds = lds_total_words_host(1u << lp, logC) * sizeof(cf);
  if (lds_bytes) *lds_bytes = lds;
  hipError_t e = dspsr_amd_allow_lds((const void*)k, lds);
  if (e == hipSuccess && plain_pol_split(logC, real_input, npol, 1))
    e = dspsr_amd_allow_lds((const void*)pick_plain(logC, 3, mkseq<MAX_LOGF>::type()),
                            lds_total_words_host(1u << plain_log_points(logC, 3), logC) * sizeof(cf));
  return e == hipSuccess ? DSPSR_AMD_OK : DSPSR_AMD_EHIP;
}

int fb_plain_launch(dspsr_amd_ctx* ctx, int logC, bool real_input, uint32_t npol, uint32_t input_nchan, const cf* kern,
                    const FbIn& in, const FbOut& out, uint64_t in_chan_stride, uint64_t npart)
{
  const bool split = plain_pol_split(logC, real_input, npol, out.kind);
  const int mode = split ? 3 : real_input ? (npol == 2 ? 0 : 1) : 2;
  kplain_t k = pick_plain(logC, mode, mkseq<MAX_LOGF>::type());
  if (!k) return DSPSR_AMD_EINVAL;
  const int lp = plain_log_points(logC, mode);
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
  const int logT = lp - logC;                        // columns per tile: (part, polarisation) pairs, or parts
  const uint32_t tile_parts = (npol == 2 && !split) ? (1u << (logT - 1)) : (1u << logT);
  const uint64_t total = ((npart + tile_parts - 1) / tile_parts) * input_nchan * (split ? 2 : 1);
  if (total >= (1ull << 31)) return DSPSR_AMD_EINVAL;
  const size_t lds = lds_total_words_host(1u << lp, logC) * sizeof(cf);
  const uint32_t wgs = ctx->ncu * (2 * lds + 1024 <= 160 * 1024 ? 2u : 1u);
  uint64_t grid = total < wgs ? total : wgs;
  if (grid >= 8) grid &= ~7ull;
  hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(1u << (lp - LOG_PTS)), lds, ctx->stream, p, ctx->tw);
  return hipGetLastError() == hipSuccess ? DSPSR_AMD_OK : DSPSR_AMD_EHIP;
}

}  // namespace dspsr_amd
