// Convolving filterbank (dsp::Filterbank -F N:D) for gfx950: shared declarations of the pass kernels (fb_*.hip) and the host
// dispatch (filterbank.hip).
//
// Reference algorithm (Signal/General/Filterbank.C:561-662, FilterbankCUDA.cu:181-304):
//   forward FFT of nsamp_fft samples per pol -> multiply first N bins by the response
//   (Response.C:385-444) -> nchan_subband backward FFTs of freq_res -> keep [nfilt_pos, +nkeep).
//
// MI355X formulation (DESIGN.md "Kernels"):
//   real dual-pol input is transformed as ONE complex sequence w = x0 + i*x1 of L = 2N points
//   (for 8-bit generic DADA data the interleaved (pol0,pol1) bytes ARE w); complex input as
//   npol sequences of L = N points.  L = M * Rr with M = freq_res, Rr = L/M spectrum rows.
//     P1 k_fwd_cols : M-point FFTs down the stride-Rr columns (+ int8 load + twiddle W_L^{nb*ka})
//     P2 k_fwd_rows : Rr-point FFTs along contiguous rows -> spectrum rows s' = k_b, bin m = k_a
//     P3 k_inv_chan : rows s and Rr-1-s -> X_pol0, X_pol1 (Hermitian split) -> x chirp
//                     -> inverse M-point FFTs -> keep window -> complex output or fused detection
//   Scratch between passes is stored blocked so every global access is a >=128-byte run:
//     A[(ka/T2)][nb][ka%T2]   (written by P1 as T1*T2-element runs, read contiguously by P2)
//     X[(s'/T3)][m][s'%T3]    (written by P2 as T2*T3-element runs, read contiguously by P3)
#pragma once
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "engine_internal.h"
#include "fold_internal.h"
#include "stamps.h"

namespace dspsr_amd {

struct FbGeom {
  int logM, logR, logT1, logT2, logT3;   // four-pass mode: logM/logR are the forward factors Fa/Fb (L = Fa*Fb), logT3 = 0
  int logX3;                             // channels per block of the X layout (>= tile channels 2^logT3 of pass 3)
  int four_pass;                         // freq_res handled by a two-pass inverse (k_inv_a + k_inv_b)
  int xblocked;                          // four-pass mode: spectrum element k = ka + Fa*kb lies at X[(ka >> logT2)*xblock + (kb << logT2
                                         //   | (ka & (T2-1)))] -- every pass-2 tile is one contiguous block (see k_fwd_rows / k_inv_a)
  uint32_t xblock;                       // elements from one block to the next: 2^(logR+logT2) + padding (power-of-two strides
                                         //   between the pieces a k_inv_a tile reads would all fall on the same memory channels)
  uint32_t kblock;                       // the same for the chirp on the device (k < N only, no padding): (N >> logM) << logT2
  uint64_t xstride;                      // elements from one spectrum (sequence) to the next in X: L, or the padded size
  int logMf, logMa, logMb, logTm, logTt; // freq_res = Ma*Mb ; m2 columns per k_inv_a tile ; t1 columns per k_inv_b tile
  int real_input, npol;
  uint32_t nsub;                         // 1, or 3 / 5: nchan_subband = nsub * 2^k -- the forward transform of L = nsub * L' points as nsub
                                         //   interleaved sub-sequences of L' = M << logR points each (passes 0-2 on the power-of-two geometry,
                                         //   k_sub_combine), the inverse pass on nsub << logR spectrum rows
  int logFb2, logFa2;                    // two-pass path (fb_two_pass.hip): L = 2^logFa2 * 2^logFb2 (Fa <= 2^14), the inverse tile holds
                                         // 2^logFb2 channels x 2 pols
  uint32_t C, nfilt_pos, nkeep;
  const float2* tw_lo;   // exp(-2*pi*i*j/L), j < L/TWN : fine part of the pass-1 twiddle (L > TWN)
  const float2* tw_lo_m; // exp(-2*pi*i*j/freq_res), j < freq_res/TWN : same for the inverse twiddle (four-pass mode)
};

struct FbIn {
  int kind;  // 0: float32 rows, 1: int8 generic, 2: int8 caspsr, 3: (pol0,pol1) byte pairs pre-transposed per tile,
             // 4: 16-bit offset-binary complex in 2048-sample blocks per polarisation (UWB)
             // 5: float32 pairs pre-transposed per tile ((pol0, pol1) of real input or (re, im) of one polarisation)
  const void* base;
  uint64_t pol_stride;  // float32: floats between pol rows
  uint64_t part_step;   // time samples between parts
  uint32_t nchan, ichan;
  float scale;
  // channel-batched convolution (float32 complex rows of several input channels in ONE launch group, filterbank.hip fb_run_batched):
  // the "parts" of pass 1 are then virtual sequences vp = (part * npol + pol) * batch + c -- channel c of the group lies
  // chan_stride_c complex samples behind `base`; 0: off
  uint32_t batch;
  uint64_t chan_stride_c;
};

struct FbOut {
  int kind;  // 0: none (benchmark), 1: complex filterbank rows, 2: detected, 3: detected and folded in the same
             //    kernel (base = device profile [chan][nbin] float4, ndim 4; plan per part, see fold_internal.h)
             // 4: four-pass geometry, wide phase bins: k_inv_b reduces the detected samples of its tile to the sums of the
             //    Tt-sample segments it holds (base = segment sums [chan][part][tile][t2][2] float4; pstart = the
             //    time-ordered interval offsets of the block's bin plan, blk_first = their index per 1024 samples,
             //    nparts_plan = parts of the block); fold_segment_combine adds them to the profile in time order
  float* base;
  uint64_t chan_stride, pol_stride, part_step;  // floats
  int state;                                    // detected: coherence / stokes
  uint32_t ndim, chan0;
  uint32_t nbin;                                // kind 3
  uint64_t prof_span4;                          // kind 3: float4 between consecutive channel rows of the profile
  uint32_t prof_planes;                         // kind 3: 1 = one float4 (PP, QQ, Re, Im) per bin (npol 1, ndim 4); 2 = two rows of
                                                //         float2 per channel, (PP, QQ) and (Re, Im) (npol 2, ndim 2: the layout
                                                //         the reference's GPU pipeline folds, LoadToFold1.C:1105-1109)
  uint64_t plane_stride;                        // kind 3, prof_planes 2: floats from the (PP, QQ) row to the (Re, Im) row
  float* part;                                  // kind 3, nseg > 1: partial profiles of part segments 1 .. nseg-1 for the
                                                //         nchan_subband channels of this launch, packed
                                                //         [seg-1][chan - chan0][nbin] float4, zeroed before the launch
  uint32_t nchan_prof;                          // kind 3: channel rows of the whole profile
  uint32_t nseg;                                // kind 3: part segments of a launch folded by different workgroups (0/1: one)
  dspsr_amd_fold* fold;                         // kind 3 (host side only): the engine whose profile `base` is
  const uint32_t* pstart;                       // kind 3: per-part active-bin plan (fold_internal.h), nparts_plan parts
  uint32_t nparts_plan;
  uint32_t plan_cap;                            // kind 3: plan entries per LDS buffer (two buffers behind the twiddles)
  const Interval* piv;                          // kind 3: intervals (offset within the part, hits), time ordered per bin
  const uint32_t* blk_first;                    // kind 4: interval that holds sample 1024*i of the block
  const uint32_t* bin_start;                    // kind 4 (host side only): the intervals bucketed by phase bin (with piv)
  // kind 5: search mode (digifil -F N:D): square-law detection (state = DSPSR_AMD_INTENSITY | DSPSR_AMD_PPQQ) and the time
  // scrunch of the detected STREAM inside the inverse pass; base = FPT rows [chan][npol_out] of scrunched samples, chan_stride /
  // pol_stride in floats.  See ts_part / the search epilogue below.
  uint32_t ts_sf;                               //   scrunch factor
  uint32_t ts_magic;                            //   ceil(2^32 / sf): t / sf = umulhi(t, magic) for t * sf < 2^32 (host checks)
  uint32_t ts_phase0;                           //   samples of output 0 already summed into the carry when the call begins
  uint32_t ts_G;                                //   groups per staged row (odd, >= the most groups a part can touch)
  float* ts_carry;                              //   [chan][npol_out] partial sums of the output sample still open
};

// Search-mode epilogue (FbOut kind 5): the detected samples of a channel form a stream over the parts of a call (and over calls:
// the carry).  Output sample o is the sum, in time order, of stream samples [o*sf, (o+1)*sf) (TScrunch.C:148-178).  A part's nkeep
// samples start at stream index s = phase0 + part * nkeep: they touch groups ofirst .. ofirst + ng - 1, the first one from its
// element phi on, the last one up to element rlast - 1.
struct TsPart { uint32_t phi, ng, rlast, ofirst; };
DEV TsPart ts_part(const FbOut& out, const uint32_t part_in_call, const uint32_t nkeep)
{
  const uint32_t s = out.ts_phase0 + part_in_call * nkeep;       // (< 2^32: the host bounds npart * nkeep)
  TsPart p;
  p.ofirst = s / out.ts_sf;
  p.phi = s - p.ofirst * out.ts_sf;
  const uint32_t e = p.phi + nkeep;
  p.ng = (e + out.ts_sf - 1) / out.ts_sf;
  p.rlast = e - (p.ng - 1) * out.ts_sf;
  return p;
}
// Square-law detection as the reference's host code rounds it (Detection.C:273-279: `*out = re*re; *out += im*im;` -- two
// roundings, no fused multiply-add on its x86 builds)
DEV float sqld(const cf a) { return __fadd_rn(__fmul_rn(a.x, a.x), __fmul_rn(a.y, a.y)); }
// The tile's detected samples are staged as [channel * npol_out + q][element r of the group][group] floats (G groups per row, G
// odd): the scrunch then reads element r of consecutive groups from consecutive words (conflict free), and the last stage's writes
// -- consecutive samples of a lane pair's channels -- fall G words apart.
DEV void ts_stage(float* __restrict__ stg, const FbOut& out, const TsPart& tp, const uint32_t slo, const uint32_t t, const cf a, const cf b)
{
  const uint32_t tr = t + tp.phi, gq = out.ts_sf == 1 ? tr : __umulhi(tr, out.ts_magic), r = tr - gq * out.ts_sf;
  const float pp = sqld(a), qq = sqld(b);
  if (out.state == DSPSR_AMD_PPQQ) {
    const uint32_t base = ((2 * slo) * out.ts_sf + r) * out.ts_G + gq;
    stg[base] = pp;
    stg[base + out.ts_sf * out.ts_G] = qq;
  } else {
    stg[(slo * out.ts_sf + r) * out.ts_G + gq] = __fadd_rn(pp, qq);          // Detection.C:285-300: *p0 += *p1
  }
}
// carry of row `row` (channel * npol_out + q), read past the L1 (the previous part's store of this workgroup went to L2)
DEV float ts_carry_load(const FbOut& out, const uint32_t row) { return __hip_atomic_load(out.ts_carry + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The scrunch of one staged tile: nrow = channels * npol_out staged rows; chan_of(row / npo) = output channel (without chan0).
// Thread w < nrow takes the FIRST group of row w (the one that may continue the carry, pre-loaded into `carry_pre` by that same
// thread), the others the remaining (row, group) items.  Sums are sequential in time: out = in[0]; out += in[1]; ...
template <class ChanOf>
DEV void ts_reduce(const float* __restrict__ stg, const FbOut& out, const TsPart& tp, const uint32_t nrow, const float carry_pre,
                   const bool have_pre, const uint32_t tid, const uint32_t nthr, ChanOf&& chan_of)
{
  const uint32_t sf = out.ts_sf, G = out.ts_G, npo = out.state == DSPSR_AMD_PPQQ ? 2u : 1u;
  const uint32_t ng1 = tp.ng - 1, nitem = nrow * tp.ng;
  for (uint32_t w = tid; w < nitem; w += nthr) {
    uint32_t row, gq;
    if (w < nrow) { row = w; gq = 0; }
    else { const uint32_t x = w - nrow; row = x / ng1; gq = 1 + (x - row * ng1); }
    const uint32_t r1 = gq == ng1 ? tp.rlast : sf;
    uint32_t r = gq == 0 ? tp.phi : 0;
    const float* __restrict__ src = stg + (row * sf) * G + gq;
    float acc;
    if (gq == 0 && tp.phi) acc = (have_pre && w == tid) ? carry_pre : ts_carry_load(out, (out.chan0 + chan_of(row / npo)) * npo + row % npo);
    else { acc = src[r * G]; r++; }
    for (; r + 4 <= r1; r += 4) {                                  // loads ahead, adds in time order
      const float a0 = src[r * G], a1 = src[(r + 1) * G], a2 = src[(r + 2) * G], a3 = src[(r + 3) * G];
      acc = __fadd_rn(acc, a0); acc = __fadd_rn(acc, a1); acc = __fadd_rn(acc, a2); acc = __fadd_rn(acc, a3);
    }
    for (; r < r1; r++) acc = __fadd_rn(acc, src[r * G]);
    const uint32_t chan = out.chan0 + chan_of(row / npo), q = row % npo;
    if (r1 < sf) out.ts_carry[chan * npo + q] = acc;                // the group is still open: the next part (or call) continues it
    else out.base[chan * out.chan_stride + q * out.pol_stride + tp.ofirst + gq] = acc;
  }
}

// nchan_subband = 3 * 2^k / 5 * 2^k: arguments of k_sub_split (see the section in front of pass 2)
struct SubSplit {
  int kind;                   // FbIn::kind of the source: 0 float rows, 1 generic 8-bit, 2 CASPSR
  const void* base;
  uint64_t chan_off;          // float: floats to this input channel's rows
  uint64_t pol_stride;        // float: floats between polarisation rows
  uint32_t nchan, ichan, npol, ndim;
  uint64_t t_first;           // first sample of the group
  uint64_t nper;              // samples per sub-sequence (all windows)
  uint32_t R;
  uint64_t sub_stride;        // bytes from one sub-block to the next
  // windows: the group's parts as nwin separate windows of wlen samples per sub-sequence, win_step input samples apart (parts that
  // do not follow each other: the sub-groups of a misaligned part step); nwin = 1, wlen = nper: one contiguous range
  uint32_t nwin;
  uint64_t wlen, win_step;
};
// Twiddles of the radix-R steps outside the tiles (R = 3, 5, 7, 9, 15): exp(-2 pi i t / (R 2^logP)), t < R 2^logP.  With
// t = a 2^logP + b the twiddle is W_R^a -- R values, built in double on the host and handed over in this table -- times
// (cos, -sin)(b / 2^logP / R revolutions): b / 2^logP is exact in float for logP <= 24 and the one division by R leaves
// an argument error below 2e-8 revolutions, so the product is as accurate as the hi / lo split of twiddles_big (2.4e-7; the
// single-argument form it replaces, (a + b / 2^logP) / R, rounded the whole angle: up to 4e-7, and the time-domain step's
// t * (1 / freq_res) up to 7.5e-7).
constexpr uint32_t ODD_MAX = 127;                     // largest odd factor of a transform length
struct OddTw { float2 w[ODD_MAX + 1]; uint32_t R; };   // w[a] = exp(-2 pi i a / R), a < R
template <int R> DEV cf twiddle_odd(const uint32_t t, const int logP, const OddTw& tab)
{
  const uint32_t a = (t >> logP) % (uint32_t)R, b = t & ((1u << logP) - 1);
  const float x = (float)b * __uint_as_float((uint32_t)(127 - logP) << 23) / (float)R;
  return cmul(tab.w[a], make_float2(__builtin_amdgcn_cosf(x), -__builtin_amdgcn_sinf(x)));
}
// the same for a factor known only at run time (R any odd number <= ODD_MAX; t may need 64 bits)
DEV cf twiddle_odd_rt(const uint64_t t, const int logP, const OddTw& tab)
{
  const uint32_t a = (uint32_t)((t >> logP) % tab.R), b = (uint32_t)(t & ((1ull << logP) - 1));
  const float x = (float)b * __uint_as_float((uint32_t)(127 - logP) << 23) / (float)tab.R;
  return cmul(tab.w[a], make_float2(__builtin_amdgcn_cosf(x), -__builtin_amdgcn_sinf(x)));
}
inline OddTw make_odd_tw(const uint32_t R)
{
  OddTw t;
  t.R = R;
  for (uint32_t a = 0; a <= ODD_MAX; a++) {
    const double ang = -2.0 * M_PI * (double)(a % R) / (double)R;
    t.w[a] = make_float2((float)cos(ang), (float)sin(ang));
  }
  return t;
}
// parameters of k_time_combine (freq_res = 3 * 2^k / 5 * 2^k)
struct TimeCombine {
  const cf* Y;
  uint64_t y_chan_stride, y_pol_stride;     // complex elements; parts M' apart
  uint32_t logMi, mo, nfilt_pos, nkeep, C, npol;
  uint64_t part0;                           // part of the call that local part lp is: part0 + lp * part_stride
  uint32_t nparts, part_stride;
  OddTw tw;                                 // W_R^a of the radix-R step in time
};


[[maybe_unused]] constexpr uint32_t FB_PSL_MAX = 128;   // fused fold: offsets of the parts a workgroup walks (its run of a launch), kept in LDS

// (int8 + 0.5) * scale (GenericEightBitUnpackerCUDA.cu:45).  int8 + 0.5 is exact in float, so the one rounding of the product
// is the rounding of the exact value (v + 0.5)*scale -- which fma(v, scale, scale/2) rounds likewise (scale/2 is exact):
// bit-identical, one instruction less per pair of samples
DEV float cvt8(int v, float scale) { return __builtin_fmaf((float)v, scale, 0.5f * scale); }

// streaming accesses: scratch and output data are written once and read once by another pass.  The loads carry the
// non-temporal hint (measured: pass 2 -7 %, pass 3 -6 %, profiles/r01c_experiments.txt); the stores are plain (non-temporal
// stores measured slower: the next pass finds part of a plain-stored tile in the Infinity Cache)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
DEV void st_stream(float4* p, const float4 v)
{
  *p = v;
}
DEV void st_stream(float2* p, const float2 v)
{
  *p = v;
}
DEV float4 ld_stream(const float4* p)
{
  const f4v t = __builtin_nontemporal_load((const f4v*)p);
  return make_float4(t[0], t[1], t[2], t[3]);
}
DEV float2 ld_stream(const float2* p)
{
  const f2v t = __builtin_nontemporal_load((const f2v*)p);
  return make_float2(t[0], t[1]);
}

// ---- input: two time-adjacent samples (columns col, col+1 of a tile) per request ---------------
// The load is split in two so that a persistent workgroup can issue the loads of its NEXT tile
// before computing the current one and only convert them afterwards:
//   fetch_pair  : issues the global loads, result = up to 4 raw 32-bit words
//   decode_pair : raw words -> two complex float samples  ((int8 + 0.5) * scale for 8-bit data,
//                 GenericEightBitUnpackerCUDA.cu:45)
template <int W> struct RawW { uint32_t w[W]; };
typedef RawW<4> Raw4;

template <int W> DEV RawW<W> fetch_pair(const FbGeom& g, const FbIn& in, const uint32_t seq, const uint64_t t)
{
  RawW<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = 0u;
  if constexpr (W == 1) {
    // one 32-bit word per pair: 8-bit real dual-pol, single input channel, 4-byte aligned (generic order) or
    // the pre-transposed copy; t is the byte-pair index
    r.w[0] = *(const uint32_t*)((const uint8_t*)in.base + 2 * t);
    return r;
  } else {
  if (in.kind == 5) {                                   // regrouped float32 pairs: columns t, t+1 are 16 contiguous bytes
    const uint4 v = *(const uint4*)((const cf*)in.base + t);
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  } else if (in.kind == 0) {                            // float32 rows
    if (g.real_input) {
      const float* x = (const float*)in.base + t;
      r.w[0] = __float_as_uint(x[0]); r.w[1] = __float_as_uint(x[1]);
      if (g.npol == 2) { r.w[2] = __float_as_uint(x[in.pol_stride]); r.w[3] = __float_as_uint(x[in.pol_stride + 1]); }
    } else {
      const float* x = (const float*)in.base + seq * in.pol_stride + 2 * t;
      r.w[0] = __float_as_uint(x[0]); r.w[1] = __float_as_uint(x[1]);
      r.w[2] = __float_as_uint(x[2]); r.w[3] = __float_as_uint(x[3]);
    }
  } else if (in.kind == 2) {                            // CASPSR: 4 B pol0, 4 B pol1 (t even)
    const uint8_t* b = (const uint8_t*)in.base + (t >> 2) * 8 + (t & 3);
    r.w[0] = *(const uint16_t*)b;
    r.w[1] = *(const uint16_t*)(b + 4);
  } else if (in.kind == 4) {                            // UWB: word (block*npol + pol)*2048 + t%2048 = (re, im) int16
    const uint32_t* b = (const uint32_t*)in.base;
    const uint64_t t1 = t + 1;
    r.w[0] = b[((t >> 11) * g.npol + seq) * 2048 + (t & 2047)];
    r.w[1] = b[((t1 >> 11) * g.npol + seq) * 2048 + (t1 & 2047)];
  } else if (g.real_input) {                            // generic 8-bit, byte (t*nchan + c)*npol + p
    const uint64_t skip = (uint64_t)in.nchan * g.npol;
    const uint8_t* b = (const uint8_t*)in.base + t * skip + (uint64_t)in.ichan * g.npol;
    // (loads are never combined here: the words stay in flight until decode_pair, see the complex case)
    if (g.npol == 2) {
      if (in.nchan == 1 && (((uintptr_t)in.base) & 3) == 0) {
        r.w[0] = *(const uint32_t*)b;                   // (p0,p1)[t], (p0,p1)[t+1]   (t is even)
      } else if ((((uintptr_t)in.base) & 1) == 0) {
        r.w[0] = *(const uint16_t*)b;
        r.w[1] = *(const uint16_t*)(b + skip);
      } else {
        r.w[0] = b[0]; r.w[2] = b[1]; r.w[1] = b[skip]; r.w[3] = b[skip + 1];
      }
    } else {
      r.w[0] = b[0];
      r.w[1] = b[skip];
    }
  } else {                                              // generic 8-bit complex: ((t*nchan+c)*npol+p)*2+d
    const uint64_t skip = (uint64_t)in.nchan * g.npol * 2;
    const uint8_t* b = (const uint8_t*)in.base + t * skip + ((uint64_t)in.ichan * g.npol + seq) * 2;
    // two independent 16-bit loads, combined only in decode_pair: the words stay in flight while the previous tile
    // is transformed (combining them here would wait for the loads at the prefetch)
    if (in.nchan == 1 && g.npol == 2 && (((uintptr_t)in.base) & 7) == 0) {
      // single channel, two polarisations: samples t, t+1 (t even) are one aligned 8-byte group holding both
      // polarisations; one coalesced load, the polarisation is picked in decode_pair
      const uint2 v = *(const uint2*)((const uint8_t*)in.base + t * 4);
      r.w[0] = v.x; r.w[1] = v.y;
    } else if ((((uintptr_t)in.base) & 1) == 0) {
      r.w[0] = *(const uint16_t*)b;
      r.w[1] = *(const uint16_t*)(b + skip);
    } else {
      r.w[0] = b[0]; r.w[2] = b[1]; r.w[1] = b[skip]; r.w[3] = b[skip + 1];
    }
  }
  return r;
  }
}

template <int W> DEV void decode_pair(const FbGeom& g, const FbIn& in, const RawW<W>& r, cf& a, cf& b, const uint32_t seq = 0)
{
  if constexpr (W == 1) {
    a = make_float2(cvt8((int8_t)(r.w[0] & 0xff), in.scale), cvt8((int8_t)((r.w[0] >> 8) & 0xff), in.scale));
    b = make_float2(cvt8((int8_t)((r.w[0] >> 16) & 0xff), in.scale), cvt8((int8_t)(r.w[0] >> 24), in.scale));
    return;
  } else {
  if (in.kind == 5) {
    a = make_float2(__uint_as_float(r.w[0]), __uint_as_float(r.w[1]));
    b = make_float2(__uint_as_float(r.w[2]), __uint_as_float(r.w[3]));
  } else if (in.kind == 0) {
    if (g.real_input) {
      a = make_float2(__uint_as_float(r.w[0]), g.npol == 2 ? __uint_as_float(r.w[2]) : 0.0f);
      b = make_float2(__uint_as_float(r.w[1]), g.npol == 2 ? __uint_as_float(r.w[3]) : 0.0f);
    } else {
      a = make_float2(__uint_as_float(r.w[0]), __uint_as_float(r.w[1]));
      b = make_float2(__uint_as_float(r.w[2]), __uint_as_float(r.w[3]));
    }
  } else if (in.kind == 4) {                            // convert_offset_binary, UWBUnpackerCUDA.cu:24
    a = make_float2((float)(int16_t)((r.w[0] & 0xffff) ^ 0x8000) * in.scale, (float)(int16_t)((r.w[0] >> 16) ^ 0x8000) * in.scale);
    b = make_float2((float)(int16_t)((r.w[1] & 0xffff) ^ 0x8000) * in.scale, (float)(int16_t)((r.w[1] >> 16) ^ 0x8000) * in.scale);
  } else if (in.kind == 2) {
    a = make_float2(cvt8((int8_t)(r.w[0] & 0xff), in.scale), cvt8((int8_t)(r.w[1] & 0xff), in.scale));
    b = make_float2(cvt8((int8_t)((r.w[0] >> 8) & 0xff), in.scale), cvt8((int8_t)((r.w[1] >> 8) & 0xff), in.scale));
  } else {
    if (g.real_input && g.npol == 1) {
      a = make_float2(cvt8((int8_t)(r.w[0] & 0xff), in.scale), 0.0f);
      b = make_float2(cvt8((int8_t)(r.w[1] & 0xff), in.scale), 0.0f);
    } else if (g.real_input && in.nchan == 1 && (((uintptr_t)in.base) & 3) == 0) {     // one word: (p0,p1)[t], (p0,p1)[t+1]
      a = make_float2(cvt8((int8_t)(r.w[0] & 0xff), in.scale), cvt8((int8_t)((r.w[0] >> 8) & 0xff), in.scale));
      b = make_float2(cvt8((int8_t)((r.w[0] >> 16) & 0xff), in.scale), cvt8((int8_t)(r.w[0] >> 24), in.scale));
    } else {                  // byte pair of sample t in w[0] (| w[2] << 8), of sample t+1 in w[1] (| w[3] << 8)
      uint32_t w0 = r.w[0] | (r.w[2] << 8), w1 = r.w[1] | (r.w[3] << 8);
      if (!g.real_input && in.nchan == 1 && g.npol == 2 && (((uintptr_t)in.base) & 7) == 0) {   // whole samples were loaded
        w0 = r.w[0] >> (16 * seq);
        w1 = r.w[1] >> (16 * seq);
      }
      a = make_float2(cvt8((int8_t)(w0 & 0xff), in.scale), cvt8((int8_t)((w0 >> 8) & 0xff), in.scale));
      b = make_float2(cvt8((int8_t)(w1 & 0xff), in.scale), cvt8((int8_t)((w1 >> 8) & 0xff), in.scale));
    }
  }
  }
}

// Pass twiddles exp(-2*pi*i*j/2^logL), j < 2^logL: a coarse table (2*pi/TWN steps, built in double) times a fine
// table (the remaining low bits of j), both correctly rounded -> about 1.2e-7 relative error.
// NT twiddles exp(-2*pi*i*j[q]/2^logL) at once: all table loads are issued back to back (one memory round trip)
// and only then combined -- evaluating them one by one costs a dependent L1/L2 round trip each
// tile of k_float_transpose: rows x columns of 8-byte elements through LDS (32 x 128 and 16 x 256 measured no faster,
// profiles/r04_experiments.txt item 8)
constexpr uint32_t FB_FT_ROWS = 64, FB_FT_COLS = 64;
template <int NT, typename IDX> DEV void twiddles_big(cf (&t)[NT], const IDX (&j)[NT], const int logL, const cf* __restrict__ tw,
                                                      const cf* __restrict__ tw_lo)
{
  if (logL <= 24) {      // uniform
    // v_cos_f32 / v_sin_f32 take their argument in revolutions: j / 2^logL is exact in float, and the measured
    // error over all j of 2^23 (tools/sincos_probe.hip) is 1.25e-7 max, 3.5e-8 rms -- the same as the product of the
    // coarse and fine table entries, without their loads and the memory round trip in front of the ladder
    const float sc = __uint_as_float((uint32_t)(127 - logL) << 23);
#pragma unroll
    for (int q = 0; q < NT; q++) {
      const float x = (float)(uint32_t)j[q] * sc;
      t[q] = make_float2(__builtin_amdgcn_cosf(x), -__builtin_amdgcn_sinf(x));
    }
  } else if (logL <= 32) {   // uniform (32-bit indices)
    // L > 2^24: j / L is no longer exact in float -- two exact arguments instead, hi = j >> s (13 bits) in revolutions of 2^13
    // and lo = j mod 2^s in revolutions of L, and one product: 2.4e-7 max against double (tools/sincos_probe.hip; coarse x fine
    // tables 1.3e-7).  The table form cost pass 2 ten dependent L2 round trips at the top of EVERY tile (the compiler sinks each
    // load to its use: `global_load; s_waitcnt vmcnt(0)` chains in the listing), with nothing else in flight: cfg1opt's
    // k_fwd_rows 821 -> see profiles/r04_experiments.txt item 13.
    const int s = logL - 13;
    const float scl = __uint_as_float((uint32_t)(127 - logL) << 23);
#pragma unroll
    for (int q = 0; q < NT; q++) {
      const uint32_t jq = (uint32_t)j[q];
      const float xh = (float)(jq >> s) * (1.0f / 8192.0f), xl = (float)(jq & ((1u << s) - 1)) * scl;
      const float ch = __builtin_amdgcn_cosf(xh), sh = __builtin_amdgcn_sinf(xh), cl = __builtin_amdgcn_cosf(xl), sl = __builtin_amdgcn_sinf(xl);
      t[q] = make_float2(ch * cl - sh * sl, -(ch * sl + sh * cl));
    }
  } else if (logL <= LOG_TWN) {                // uniform
#pragma unroll
    for (int q = 0; q < NT; q++) t[q] = tw[j[q] << (LOG_TWN - logL)];
  } else {
    const int sh = logL - LOG_TWN;
    cf lo[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) { t[q] = tw[j[q] >> sh]; lo[q] = tw_lo[j[q] & ((1u << sh) - 1)]; }
#pragma unroll
    for (int q = 0; q < NT; q++) t[q] = cmul(t[q], lo[q]);
  }
}

// v[k] *= W_L^{nb*(k*pstride + p)} for the column pair (nb, nb+1), k < R : base and the powers 1,2,4,8 of
// the step from the (coarse x fine) tables, the rest by the ladder
template <int R> DEV void apply_pass_twiddle(cx2 (&v)[R], const uint32_t nb, const uint32_t p, const uint32_t pstride,
                                             const int logL, const cf* __restrict__ tw, const cf* __restrict__ tw_lo)
{
  // 32-bit index arithmetic: nb < Fb and k*pstride + p < Fa with both factors <= 2^MAX_LOGF = 2^13, so every product is
  // below 2^26 and its multiples up to 8 below 2^29
  const uint32_t Lm = (uint32_t)((1ull << logL) - 1);
  const uint32_t a0 = (nb * p) & Lm, d0 = (nb * pstride) & Lm;
  const uint32_t a1 = (a0 + p) & Lm, d1 = (d0 + pstride) & Lm;             // column nb + 1
  constexpr int NP = R >= 16 ? 4 : R >= 8 ? 3 : R >= 4 ? 2 : R >= 2 ? 1 : 0;   // powers 1, 2, 4, 8 of the step
  uint32_t j[2 + 2 * (NP ? NP : 1)];
  cf t[2 + 2 * (NP ? NP : 1)];
  j[0] = a0; j[1] = a1;
#pragma unroll
  for (int q = 0; q < (NP ? NP : 1); q++) { j[2 + 2 * q] = (d0 << q) & Lm; j[3 + 2 * q] = (d1 << q) & Lm; }
  twiddles_big(t, j, logL, tw, tw_lo);
  const cx2 wa = make_cx2(t[0], t[1]);
  if constexpr (R == 1) {
    v[0] = cmul(v[0], wa);
  } else {
    // u[k] = wa * w1^k by a ladder that starts from wa (15 products for R = 16) instead of w1^k (11 products) followed by
    // a separate multiplication of every element by wa (16 more): 31 packed complex products per call instead of 42
    const cx2 w1 = make_cx2(t[2], t[3]);
    const cx2 w2 = NP >= 2 ? make_cx2(t[2 + 2 * (NP >= 2 ? 1 : 0)], t[3 + 2 * (NP >= 2 ? 1 : 0)]) : w1;
    const cx2 w4 = NP >= 3 ? make_cx2(t[2 + 2 * (NP >= 3 ? 2 : 0)], t[3 + 2 * (NP >= 3 ? 2 : 0)]) : w1;
    const cx2 w8 = NP >= 4 ? make_cx2(t[2 + 2 * (NP >= 4 ? 3 : 0)], t[3 + 2 * (NP >= 4 ? 3 : 0)]) : w1;
    cx2 u[R];
    u[0] = wa;
    u[1] = cmul(wa, w1);
    if constexpr (R >= 4) { u[2] = cmul(wa, w2); u[3] = cmul(u[1], w2); }
    if constexpr (R >= 8) {
#pragma unroll
      for (int k = 0; k < 4; k++) u[4 + k] = cmul(u[k], w4);
    }
    if constexpr (R >= 16) {
#pragma unroll
      for (int k = 0; k < 8; k++) u[8 + k] = cmul(u[k], w8);
    }
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = cmul(v[k], u[k]);
  }
}

// v[k] *= conj(W_L^{nb*(k*pstride + p)}) for BOTH columns of the pair (the two polarisations of one column
// nb), k < R : the inter-pass twiddle of the two-pass inverse transform
template <int R> DEV void apply_pass_twiddle_inv(cx2 (&v)[R], const uint32_t nb, const uint32_t p, const uint32_t pstride,
                                                 const int logL, const cf* __restrict__ tw, const cf* __restrict__ tw_lo)
{
  const uint32_t Lm = (uint32_t)((1ull << logL) - 1);
  const uint32_t a0 = (nb * p) & Lm, d0 = (nb * pstride) & Lm;      // (factors <= 2^13 each: see apply_pass_twiddle)
  constexpr int NP = R >= 16 ? 4 : R >= 8 ? 3 : R >= 4 ? 2 : R >= 2 ? 1 : 0;
  uint32_t j[1 + (NP ? NP : 1)];
  cf t[1 + (NP ? NP : 1)];
  j[0] = a0;
#pragma unroll
  for (int q = 0; q < (NP ? NP : 1); q++) j[1 + q] = (d0 << q) & Lm;
  twiddles_big(t, j, logL, tw, tw_lo);
#pragma unroll
  for (int q = 0; q < 1 + (NP ? NP : 1); q++) t[q].y = -t[q].y;          // conjugate: inverse transform
  if constexpr (R > 1) {
    const cf w1 = t[1];
    const cf w2 = NP >= 2 ? t[1 + (NP >= 2 ? 1 : 0)] : w1, w4 = NP >= 3 ? t[1 + (NP >= 3 ? 2 : 0)] : w1,
             w8 = NP >= 4 ? t[1 + (NP >= 4 ? 3 : 0)] : w1;
    apply_powers<R>(v, w1, w2, w4, w8);
  }
#pragma unroll
  for (int k = 0; k < R; k++) v[k] = cmuls(v[k], t[0]);
}

DEV void detect4(const cf p, const cf q, const int state, float (&r)[4])
{
  // cross_detect.ic:23-43 / stokes_detect.ic:21-44
  const float pp = p.x * p.x + p.y * p.y;
  const float qq = q.x * q.x + q.y * q.y;
  const float re = p.x * q.x + p.y * q.y;
  const float im = p.x * q.y - p.y * q.x;
  if (state == DSPSR_AMD_STOKES) { r[0] = pp + qq; r[1] = pp - qq; r[2] = 2.0f * re; r[3] = 2.0f * im; }
  else { r[0] = pp; r[1] = qq; r[2] = re; r[3] = im; }
}

// ------------------------------------------------------------------------------------ host
typedef void (*k1_t)(FbGeom, FbIn, cf*, const cf*, uint64_t, uint32_t, uint32_t, uint32_t);
typedef void (*k2_t)(FbGeom, const cf*, cf*, const cf*, uint32_t, uint32_t, uint32_t);
typedef void (*k3_t)(FbGeom, const cf*, const cf*, FbOut, const cf*, uint64_t, uint32_t, uint32_t);
typedef void (*k3a_t)(FbGeom, const cf*, const cf*, cf*, const cf*, uint32_t, uint32_t);
typedef void (*k3b_t)(FbGeom, const cf*, FbOut, const cf*, uint64_t, uint32_t, uint32_t);

template <int... I> struct iseq {};
template <int N, int... I> struct mkseq : mkseq<N - 1, N - 1, I...> {};
template <int... I> struct mkseq<0, I...> { typedef iseq<I...> type; };

// full-size tiles (2^14 points) have 2^(14 - LOGF) columns: instantiated with that as a compile-time constant
constexpr int full_logt(int logf) { return 14 - logf >= 1 ? 14 - logf : -1; }
constexpr int MAX_LOGF = 13;    // every pass keeps >= 2 columns per workgroup
typedef mkseq<MAX_LOGF + 1>::type seq_t;
// kernel tables live in the translation unit that instantiates the kernels
k1_t fb_pick1(int logf, int raww, bool full);
k1_t fb_pick1_dual(int raww);      // pass 1 on pairs of two-column tiles (2^13-point columns), or null
k2_t fb_pick2(int logf, bool full);
k3_t fb_pick3(int logf, bool full);       // plain
k3_t fb_pick3f(int logf, bool full);      // fused fold
k3_t fb_pick3s(int logf, bool full);      // search mode (detection + time scrunch)
k3a_t fb_pick3a(int logf, bool blocked, bool real, bool full);
k3b_t fb_pick3b(int logf, bool foldb, bool full);
// two-pass path (fb_two_pass.hip): pass 1 on whole columns, rows + inverse pass (M = 2^logm, Fb = 2^(13 - logm)), the 8-bit regroup
typedef void (*k1c_t)(FbGeom, FbIn, cf*, const cf*, uint32_t, uint32_t, uint32_t);
k1c_t fb_pick_col1();
k3_t fb_pick_rinv(int logm, int epi);      // epilogue: 0 output written, 1 fused fold, 2 search mode
void fb_launch_raw_cols(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0);
void fb_launch_sub_split(hipStream_t stream, const SubSplit& p, uint8_t* out, uint32_t ncu);
// radix-nsub step on the sub-spectra in X.  Xout == nullptr: in place where a kernel is instantiated for the factor (3, 5, 7, 9,
// 15), else into Xalt; returns the buffer that holds the combined spectrum.  Xout != nullptr (freq_res with an odd factor): the
// spectrum in pseudo-channel order into Xout.
cf* fb_launch_sub_combine(hipStream_t stream, const FbGeom& g, cf* X, uint32_t nseqs, uint32_t ncu, cf* Xalt, cf* Xout = nullptr, uint32_t mo = 0,
                          uint32_t rm = 1);
void fb_launch_time_combine(hipStream_t stream, const TimeCombine& p, const FbOut& out, uint32_t R, uint32_t ncu);
void fb_launch_raw_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0);
void fb_launch_float_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, cf* Rt, uint64_t part0);
// dsp::Convolution with n_fft <= 8192 in one tile pass (fb_conv1.hip): complex float32 rows, two polarisations
int fb_conv1_check(int logM, size_t* lds_bytes);
int fb_conv1_launch(dspsr_amd_ctx* ctx, int logM, const float* in, uint64_t chan_stride, uint64_t pol_stride, uint64_t in_step,
                    const cf* kern, const FbOut& out, uint32_t nchan, uint32_t nfilt_pos, uint32_t nkeep, uint64_t npart);
// non-convolving filterbank, freq_res = 1 (fb_plain.hip): kernel choice + dynamic-LDS limit at create time, one launch per call
// dsp::Convolution in three tile passes (fb_conv3.hip): n_fft = 2^14 ... 2^21 on complex float32 rows with two polarisations
constexpr int CONV3_MIN_LOGM = 14, CONV3_MAX_LOGM = 21;
int fb_conv3_check(int logM);
void fb_conv3_response_order(int logM, const cf* natural, cf* ordered);      // one channel's response, M bins
int fb_conv3_launch(dspsr_amd_ctx* ctx, int logM, const float* in, uint64_t chan_stride, uint64_t pol_stride, uint64_t in_step,
                    const cf* kern, const FbOut& out, uint32_t nchan, uint32_t nfilt_pos, uint32_t nkeep, uint64_t part0, uint32_t nparts,
                    cf* S1, cf* S2);
int fb_plain_check(dspsr_amd_ctx* ctx, int logC, bool real_input, uint32_t npol, size_t* lds_bytes);
int fb_plain_launch(dspsr_amd_ctx* ctx, int logC, bool real_input, uint32_t npol, uint32_t input_nchan, const cf* kern,
                    const FbIn& in, const FbOut& out, uint64_t in_chan_stride, uint64_t npart);

}  // namespace dspsr_amd
