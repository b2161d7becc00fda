// Convolving filterbank (dsp::Filterbank -F N:D) for gfx950: three passes per overlap-save part.
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
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "engine_internal.h"
#include "fold_internal.h"

// Experiment switches (ablation bits that make results wrong, geometry overrides) exist only in builds made with
// -DDSPSR_AMD_EXPERIMENT (tools/build_variant.sh).  The shipped library never reads the environment: its behaviour
// depends on the configuration structs of the C-ABI alone.
#ifdef DSPSR_AMD_EXPERIMENT
#define FB_DBG(g) ((g).dbg)
#define FB_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define FB_ENV_SET(name) (getenv(name) != nullptr)
#else
#define FB_DBG(g) 0
#define FB_ENV_INT(name, dflt) (dflt)
#define FB_ENV_SET(name) false
#endif

// The file is compiled as ONE translation unit (no FB_PART: experiment builds) or, by the Makefile, as several in
// parallel: FB_PART 1 = P0+P1 kernels, 2 = P2, 3 = P3, 5 = P3 with the fused fold, 4 = the two-pass inverse, 6 = the two-pass
// path of short responses (k_raw_cols, k_fwd_col1, k_rows_inv), 0 = host.
#ifdef FB_PART
#define FB_HAS(n) (FB_PART == (n))
#else
#define FB_HAS(n) 1
#endif

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
  int logFb2, logFa2;                    // two-pass path (FB_HAS(6)): L = 2^logFa2 * 2^logFb2 (Fa <= 2^14), the inverse tile holds
                                         // 2^logFb2 channels x 2 pols
  uint32_t C, nfilt_pos, nkeep;
  int dbg;   // DSPSR_AMD_DEBUG ablation bits (timing experiments only; results are wrong when set)
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
};

// nchan_subband = 3 * 2^k / 5 * 2^k: arguments of k_sub_split (see the section in front of pass 2)
struct SubSplit {
  int kind;                   // FbIn::kind of the source: 0 float rows, 1 generic 8-bit, 2 CASPSR
  const void* base;
  uint64_t chan_off;          // float: floats to this input channel's rows
  uint64_t pol_stride;        // float: floats between polarisation rows
  uint32_t nchan, ichan, npol, ndim;
  uint64_t t_first;           // first sample of the group
  uint64_t nper;              // samples per sub-sequence
  uint32_t R;
  uint64_t sub_stride;        // bytes from one sub-block to the next
};
// parameters of k_time_combine (freq_res = 3 * 2^k / 5 * 2^k)
struct TimeCombine {
  const cf* Y;
  uint64_t y_chan_stride, y_pol_stride;     // complex elements; parts M' apart
  uint32_t logMi, mo, nfilt_pos, nkeep, C, npol;
  uint64_t part0;
  uint32_t nparts;
};

#ifdef FB_STAMPS   // diagnostic build only (-DFB_STAMPS=1|2|3|4|6|7: pass to instrument -- 4 = k_inv_a, 6 = k_fwd_col1, 7 = k_rows_inv): where a
                   // tile spends its cycles (s_memtime per phase, lane 0 of wave 0 of every workgroup).  The counters live in
                   // the translation unit of the instrumented kernel (single-TU builds, or the FB_PART that holds it)
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FB_STAMPS_PART (FB_STAMPS == 1 ? 1 : FB_STAMPS == 2 ? 2 : FB_STAMPS == 3 ? 5 : FB_STAMPS == 4 ? 4 : 6)
static __device__ unsigned long long g_stamps[1024][8];     // (one per translation unit; the exported reader sees FB_STAMPS_PART's)
#if !defined(FB_PART) || FB_PART == FB_STAMPS_PART
extern "C" int dspsr_amd_debug_stamps(unsigned long long* out_host, int zero)
{
  if (zero) { static unsigned long long z[1024][8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#endif
#endif

// Experiment (-DFB_STAGGER=n): the persistent workgroups of a launch start n*8128 cycles apart in four phases, so that
// the compute units are not all in their load / store phases at the same time
// Experiment (-DFB_SETPRIO=1): static priority for the second-dispatched half of an 8-wave workgroup (waves 4-7 lose the
// VALU arbitration against their older SIMD partners, MI355X_MICROARCH.md "Two waves per SIMD" item 4)
DEV void fb_setprio()
{
#ifdef FB_SETPRIO
  if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(FB_SETPRIO);
#endif
}

DEV void fb_stagger()
{
#ifdef FB_STAGGER
  const uint32_t ph = (blockIdx.x >> 3) & 3u;
  for (uint32_t i = 0; i < ph * FB_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
#endif
}

[[maybe_unused]] constexpr uint32_t FB_PSL_MAX = 128;   // fused fold: offsets of the parts a workgroup walks (its run of a launch), kept in LDS

// (int8 + 0.5) * scale (GenericEightBitUnpackerCUDA.cu:45).  int8 + 0.5 is exact in float, so the one rounding of the product
// is the rounding of the exact value (v + 0.5)*scale -- which fma(v, scale, scale/2) rounds likewise (scale/2 is exact):
// bit-identical, one instruction less per pair of samples
DEV float cvt8(int v, float scale) { return __builtin_fmaf((float)v, scale, 0.5f * scale); }

// streaming accesses: scratch and output data are written once and read once by another pass, so the
// stores/loads may carry the non-temporal hint (build-time experiment switches FB_NT_STORE / FB_NT_LOAD)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
#ifndef FB_NT_STORE
#define FB_NT_STORE 0
#endif
#ifndef FB_NT_LOAD
#define FB_NT_LOAD 1     // measured: P2 -7 %, P3 -6 % (profiles/r01c_experiments.txt)
#endif
DEV void st_stream(float4* p, const float4 v)
{
#if FB_NT_STORE
  const f4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, (f4v*)p);
#else
  *p = v;
#endif
}
DEV void st_stream(float2* p, const float2 v)
{
#if FB_NT_STORE
  const f2v t = {v.x, v.y};
  __builtin_nontemporal_store(t, (f2v*)p);
#else
  *p = v;
#endif
}
DEV bool getenv_pair16_off(const FbGeom& g) { return (FB_DBG(g) & 64) != 0; }   // DSPSR_AMD_DEBUG bit 64: 8-byte loads in the inverse pass
DEV float4 ld_stream(const float4* p)
{
#if FB_NT_LOAD
  const f4v t = __builtin_nontemporal_load((const f4v*)p);
  return make_float4(t[0], t[1], t[2], t[3]);
#else
  return *p;
#endif
}
DEV float2 ld_stream(const float2* p)
{
#if FB_NT_LOAD
  const f2v t = __builtin_nontemporal_load((const f2v*)p);
  return make_float2(t[0], t[1]);
#else
  return *p;
#endif
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
  } else if constexpr (W == 2) {
    // tiles of 4 columns read straight from the 8-bit stream: the 4 samples x 2 polarisations of a row are one
    // aligned 8-byte group (CASPSR: 4 B pol0, 4 B pol1; generic: (p0,p1) x 4); both column pairs load the group
    const uint2 v = *(const uint2*)((const uint8_t*)in.base + (t >> 2) * 8);
    r.w[0] = v.x; r.w[1] = v.y;
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
  } else if constexpr (W == 2) {                        // `seq` carries the first column of the pair (0 or 2)
    if (in.kind == 2) {
      const uint32_t p0 = r.w[0] >> (8 * seq), p1 = r.w[1] >> (8 * seq);
      a = make_float2(cvt8((int8_t)(p0 & 0xff), in.scale), cvt8((int8_t)(p1 & 0xff), in.scale));
      b = make_float2(cvt8((int8_t)((p0 >> 8) & 0xff), in.scale), cvt8((int8_t)((p1 >> 8) & 0xff), in.scale));
    } else {
      const uint32_t w = seq ? r.w[1] : r.w[0];
      a = make_float2(cvt8((int8_t)(w & 0xff), in.scale), cvt8((int8_t)((w >> 8) & 0xff), in.scale));
      b = make_float2(cvt8((int8_t)((w >> 16) & 0xff), in.scale), cvt8((int8_t)(w >> 24), in.scale));
    }
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
#ifndef FB_FT_ROWS             // tile of k_float_transpose (rows x columns of 8-byte elements through LDS).  64 x 64 is the shipped
                               // one; 32 x 128 and 16 x 256 run too (r04_experiments.txt item 8: what round 3 saw as a device fault
                               // was a refused launch -- grid.x = Rr / COLS = 0 for Rr < COLS -- under an LD_PRELOAD of two libraries)
#define FB_FT_ROWS 64
#define FB_FT_COLS 64
#endif
#ifndef FB_DEFER_CO
#define FB_DEFER_CO 0          // 1: pass 1 copies a staged tile out at the top of the NEXT tile, in front of its register-only
                               //    first stage (measured: 540 -> 598 us per 32 parts, profiles/r03_experiments.txt item 8; off)
#endif
#ifndef FB_SPLIT
#define FB_SPLIT 0             // 1: pass 1 with two staggered four-wave groups per workgroup (see k_fwd_cols): correct (the whole
#endif                         //    GPU suite passes with it) and exactly as fast -- profiles/r03_experiments.txt item 5; off
#ifndef FB_TWIDDLE_IN_P2
#define FB_TWIDDLE_IN_P2 1     // 0: the inter-pass twiddle on pass 1's outputs (rounds 1-2a; A/B builds)
#endif
#ifndef FB_TABLE_TWIDDLES
#define FB_TABLE_TWIDDLES 0   // 1: pass twiddles from the (coarse x fine) tables for every length (comparison builds)
#endif
template <int NT, typename IDX> DEV void twiddles_big(cf (&t)[NT], const IDX (&j)[NT], const int logL, const cf* __restrict__ tw,
                                                      const cf* __restrict__ tw_lo)
{
  if (logL <= 24 && !FB_TABLE_TWIDDLES) {      // uniform
    // v_cos_f32 / v_sin_f32 take their argument in revolutions: j / 2^logL is exact in float, and the measured
    // error over all j of 2^23 (tools/sincos_probe.hip) is 1.25e-7 max, 3.5e-8 rms -- the same as the product of the
    // coarse and fine table entries, without their loads and the memory round trip in front of the ladder
    const float sc = __uint_as_float((uint32_t)(127 - logL) << 23);
#pragma unroll
    for (int q = 0; q < NT; q++) {
      const float x = (float)(uint32_t)j[q] * sc;
      t[q] = make_float2(__builtin_amdgcn_cosf(x), -__builtin_amdgcn_sinf(x));
    }
  } else if (logL <= 32 && !FB_TABLE_TWIDDLES) {   // uniform (32-bit indices)
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

#if FB_HAS(1)
// ------------------------------------------------------------------------------------ P0
// 8-bit pre-transposition: P1 needs, for every na (stride Rr samples apart), the T1 adjacent samples of
// its tile -- 2*T1 bytes per 2*Rr-byte row.  Reading those straight from the block costs one 128-byte line
// per 8 useful bytes and per lane, so for 8-bit real dual-pol input the window of each part is first
// regrouped (2 bytes per sample pair, coalesced both ways through LDS) into
//   Rt[part][tile][na][T1]  (pol0,pol1) byte pairs
// Both the generic order and the CASPSR 4-sample interleave are accepted.
__global__ __launch_bounds__(256) void k_raw_transpose(const FbGeom g, const FbIn in, uint16_t* __restrict__ Rt,
                                                       const uint64_t part0)
{
  // block: 64 rows (na) x 256 columns (nb) of byte pairs; rows are read as 16-byte pieces (8 samples),
  // written as T-sample (2T-byte) pieces of 64 consecutive rows = 128*T contiguous bytes per tile
  constexpr uint32_t ROWS = 64, COLS = 256, PITCH = COLS / 2 + 1;       // 32-bit words per LDS row (+1: bank skew)
  __shared__ uint32_t sm[ROWS * PITCH];
  const uint32_t tid = threadIdx.x;
  const uint32_t M = 1u << g.logM, Rr = 1u << g.logR;
  const int logT = g.logT1;
  const uint32_t nb0 = blockIdx.x * COLS, na0 = blockIdx.y * ROWS;
  // complex dual-pol input (generic order, 4 bytes per sample: p0 re, p0 im, p1 re, p1 im): one polarisation = one
  // sequence per blockIdx.z, its (re, im) byte pairs take the place of the (pol0, pol1) pairs of real input
  const uint32_t nsq = g.real_input ? 1u : g.npol;
  const uint64_t part = blockIdx.z / nsq;
  const uint32_t seq = blockIdx.z % nsq;
  const uint64_t t0 = (part0 + part) * in.part_step;
  const uint32_t ncol = Rr - nb0 < COLS ? Rr - nb0 : COLS, nrow = M - na0 < ROWS ? M - na0 : ROWS;
  if (ncol % 8 == 0) {
    for (uint32_t q = tid; q < nrow * (ncol / 8); q += 256) {       // 8 samples (16 bytes) per thread and step
      const uint32_t r = q / (ncol / 8), c8 = (q % (ncol / 8)) * 8;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c8;   // multiple of 4 (8 unless t0 is odd*4)
      uint32_t w[4];
      if (!g.real_input) {                                             // 8 samples x 4 bytes, keep this polarisation
        const uint4* p = (const uint4*)((const uint8_t*)in.base + 4 * t);
        const uint4 s0 = p[0], s1 = p[1];
        const int sh = 16 * seq;
        w[0] = ((s0.x >> sh) & 0xffffu) | (((s0.y >> sh) & 0xffffu) << 16);
        w[1] = ((s0.z >> sh) & 0xffffu) | (((s0.w >> sh) & 0xffffu) << 16);
        w[2] = ((s1.x >> sh) & 0xffffu) | (((s1.y >> sh) & 0xffffu) << 16);
        w[3] = ((s1.z >> sh) & 0xffffu) | (((s1.w >> sh) & 0xffffu) << 16);
      } else if (in.kind == 2) {                                       // CASPSR: 4 B pol0 | 4 B pol1
        const uint32_t* p = (const uint32_t*)((const uint8_t*)in.base + (t >> 2) * 8);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const uint32_t p0 = p[2 * h], p1 = p[2 * h + 1];
          w[2 * h] = (p0 & 0xff) | ((p1 & 0xff) << 8) | ((p0 & 0xff00) << 8) | ((p1 & 0xff00) << 16);
          w[2 * h + 1] = ((p0 >> 16) & 0xff) | (((p1 >> 16) & 0xff) << 8) | ((p0 >> 24) << 16) | ((p1 >> 24) << 24);
        }
      } else {
        const uint32_t* p = (const uint32_t*)((const uint8_t*)in.base + 2 * t);
        w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; w[3] = p[3];
      }
#pragma unroll
      for (int h = 0; h < 4; h++) sm[r * PITCH + c8 / 2 + h] = w[h];
    }
  } else {                                                              // narrow problems: 2 samples per step
    for (uint32_t q = tid; q < nrow * (ncol / 2); q += 256) {
      const uint32_t r = q / (ncol / 2), c2 = (q % (ncol / 2)) * 2;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c2;
      uint32_t w;
      if (!g.real_input) {
        const uint32_t* p = (const uint32_t*)((const uint8_t*)in.base + 4 * t);
        w = ((p[0] >> (16 * seq)) & 0xffffu) | (((p[1] >> (16 * seq)) & 0xffffu) << 16);
      } else if (in.kind == 2) {
        const uint8_t* b = (const uint8_t*)in.base + (t >> 2) * 8 + (t & 3);
        w = (uint32_t)b[0] | ((uint32_t)b[4] << 8) | ((uint32_t)b[1] << 16) | ((uint32_t)b[5] << 24);
      } else {
        w = *(const uint32_t*)((const uint8_t*)in.base + 2 * t);
      }
      sm[r * PITCH + c2 / 2] = w;
    }
  }
  __syncthreads();
  uint32_t* __restrict__ dst = (uint32_t*)(Rt + (part * nsq + seq) * ((uint64_t)M << g.logR));
  const int logW = logT - 1;                        // 32-bit words per (row, tile) piece
  const uint32_t ntl = ncol >> logT, W = 1u << logW;
  if (logW == 1 && (nrow & 1) == 0) {
    // 4-column tiles (the headline geometry): two rows of a tile are 16 contiguous bytes of the output -- one
    // 16-byte store per lane instead of four 4-byte ones (narrow per-lane accesses stream slower on this chip,
    // tools/load_width_probe.hip)
    const uint32_t nr2 = nrow >> 1;
    for (uint32_t q = tid; q < ntl * nr2; q += 256) {
      const uint32_t r = (q % nr2) * 2, tl = q / nr2;
      const uint32_t* s0 = &sm[r * PITCH + 2 * tl];
      const uint32_t* s1 = s0 + PITCH;
      *(uint4*)&dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << 1)] = make_uint4(s0[0], s0[1], s1[0], s1[1]);
    }
    return;
  }
  if (logW >= 2) {                                                   // tiles of >= 8 columns: 16 bytes of a row piece per lane
    const int logV = logW - 2;
    for (uint32_t q = tid; q < (ntl * nrow) << logV; q += 256) {
      const uint32_t v4 = q & ((1u << logV) - 1), r = (q >> logV) % nrow, tl = (q >> logV) / nrow;
      const uint32_t* s0 = &sm[r * PITCH + (tl << logW) + 4 * v4];
      *(uint4*)&dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << logW) + 4 * v4] = make_uint4(s0[0], s0[1], s0[2], s0[3]);
    }
    return;
  }
  for (uint32_t q = tid; q < ntl * nrow * W; q += 256) {           // [tile][row][word]: runs of nrow*T pairs
    const uint32_t wd = q & (W - 1), r = (q >> logW) % nrow, tl = (q >> logW) / nrow;
    dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << logW) + wd] = sm[r * PITCH + (tl << logW) + wd];
  }
}

// The same regrouping for float32 input -- what dsp::Filterbank::Engine::perform is handed by DSPSR (the input is unpacked
// before the boundary): T1 adjacent samples of a row are 4*T1 bytes per polarisation row, 16-byte pieces 8 KB apart at the
// headline geometry, and pass 1 reading them in place ran four times slower than from 8-bit data (2223 against 562 us per
// 32 parts).  Elements are 8 bytes: (pol0, pol1) of a real sample pair, or (re, im) of one polarisation of complex input
//   Rt[part][seq][tile][na][T1]   (lives in the X scratch, which is idle until pass 2 writes it)
__global__ __launch_bounds__(256) void k_float_transpose(const FbGeom g, const FbIn in, cf* __restrict__ Rt, const uint64_t part0)
{
  constexpr uint32_t ROWS = FB_FT_ROWS, COLS = FB_FT_COLS, PITCH = COLS + 1;
  __shared__ cf sm[ROWS * PITCH];
  const uint32_t tid = threadIdx.x;
  const uint32_t M = 1u << g.logM, Rr = 1u << g.logR;
  const int logT = g.logT1;
  const uint32_t nb0 = blockIdx.x * COLS, na0 = blockIdx.y * ROWS;
  const uint32_t nsq = g.real_input ? 1u : g.npol;
  const uint64_t part = blockIdx.z / nsq;
  const uint32_t seq = blockIdx.z % nsq;
  const uint64_t t0 = (part0 + part) * in.part_step;
  const uint32_t ncol = Rr - nb0 < COLS ? Rr - nb0 : COLS, nrow = M - na0 < ROWS ? M - na0 : ROWS;   // ncol % 4 == 0 (host)
  const float* __restrict__ x = (const float*)in.base;
  if (g.real_input) {
    for (uint32_t q = tid; q < nrow * (ncol / 4); q += 256) {            // 4 samples of both polarisations per step
      const uint32_t r = q / (ncol / 4), c4 = (q % (ncol / 4)) * 4;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c4;
      const float4 p0 = ld_stream((const float4*)(x + t)), p1 = ld_stream((const float4*)(x + in.pol_stride + t));
      cf* d = &sm[r * PITCH + c4];
      d[0] = make_float2(p0.x, p1.x); d[1] = make_float2(p0.y, p1.y); d[2] = make_float2(p0.z, p1.z); d[3] = make_float2(p0.w, p1.w);
    }
  } else {
    for (uint32_t q = tid; q < nrow * (ncol / 2); q += 256) {            // 2 complex samples per step
      const uint32_t r = q / (ncol / 2), c2 = (q % (ncol / 2)) * 2;
      const uint64_t t = t0 + (((uint64_t)(na0 + r)) << g.logR) + nb0 + c2;
      const float4 v = ld_stream((const float4*)(x + seq * in.pol_stride + 2 * t));
      cf* d = &sm[r * PITCH + c2];
      d[0] = make_float2(v.x, v.y); d[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();
  cf* __restrict__ dst = Rt + (part * nsq + seq) * ((uint64_t)M << g.logR);
  const uint32_t ntl = ncol >> logT, T = 1u << logT;                     // T >= 2: two elements (16 bytes) per lane
  for (uint32_t q = tid; q < ntl * nrow * (T / 2); q += 256) {
    const uint32_t h = q % (T / 2), r = (q / (T / 2)) % nrow, tl = q / ((T / 2) * nrow);
    const cf* s0 = &sm[r * PITCH + (tl << logT) + 2 * h];
    st_stream((float4*)&dst[((((uint64_t)((nb0 >> logT) + tl) << g.logM) + na0 + r) << logT) + 2 * h],
              make_float4(s0[0].x, s0[0].y, s0[1].x, s0[1].y));
  }
}

// ------------------------------------------------------------------------------------ P1
// M-point forward FFTs down T1 adjacent stride-Rr columns of one sequence of one part.
//   in : sample n = na*Rr + nb (8-bit or float32, converted on load), nb = tile*T1 + col
//   out: A[ka/T2][nb][ka%T2] = W_L^{nb*ka} * sum_na w[na*Rr+nb] W_M^{na*ka}
// Persistent: each workgroup walks its items (tile fastest, then sequence, then part) and
// prefetches the raw samples of the next item while transforming the current one.
// LOGT >= 0: the number of columns per tile (2^LOGT) is a compile-time constant (the usual full-size tile,
// LOGT = 14 - LOGF), so every LDS address and stride folds into immediates; LOGT = -1: taken from the geometry.
template <int LOGF, int RAWW, int LOGT>
__global__ __launch_bounds__(512) void k_fwd_cols(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                  const cf* __restrict__ tw, const uint64_t part0,
                                                  const uint32_t nparts, const uint32_t nseq, const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  // SPLIT (full-size tiles of >= 4 columns): the 512 threads work as TWO GROUPS of four waves (one wave per SIMD each), each
  // transforming half the columns of the tile in its own half of the exchange buffer, one barrier phase apart.  Every
  // s_barrier is still the whole workgroup's, but between two barriers one group runs a butterfly phase (vector unit)
  // while the other runs an exchange-write phase (LDS store path): the phases of a transform alternate V, L, V, L, V, L.
  // With all eight waves in lockstep the two kinds of phase ran one after the other -- about 6.3k cycles of butterflies
  // plus 6.4k cycles of LDS transfers per 2^14-point tile.
  constexpr bool SPLIT = FB_SPLIT && LOGT >= 2;
  const uint32_t grp = SPLIT ? threadIdx.x >> 8 : 0u;                         // wave-uniform
  uint32_t tid = SPLIT ? (threadIdx.x & 255u) : threadIdx.x;                 // thread of the group
  const int logT = LOGT >= 0 ? LOGT : g.logT1, logT2 = g.logT2;
  const int logTw = SPLIT ? logT - 1 : logT;                                  // columns a group transforms
  const uint32_t T = 1u << logT, T2 = 1u << logT2, Tw = 1u << logTw;
  const uint32_t cofs = grp << logTw;                                         // first column of the group inside the tile
  const int logL = LOGF + g.logR;          // g.logM == LOGF
  const uint64_t L = 1ull << logL;
  const uint32_t ntile = 1u << (g.logR - logT);
  const uint32_t total = ntile * nseq * nparts;
  const int logNt = g.logR - logT;          // ntile = 2^logNt ; nseq is 1 or 2
  auto seq_of = [&](const uint32_t rest) { return nseq == 2 ? (rest & 1u) : 0u; };
  auto part_of = [&](const uint32_t rest) { return nseq == 2 ? (rest >> 1) : rest; };

  auto fetch = [&](const uint32_t item, RawW<RAWW> (&raw)[PTS / 2]) {
    const uint32_t tile = item & (ntile - 1);
    const uint32_t rest = item >> logNt;
    const uint32_t seq = seq_of(rest);
    const bool pret = in.kind == 3 || in.kind == 5;   // pre-transposed: [part][tile][na][T] pairs, contiguous per tile
    const uint64_t t0 = pret ? ((uint64_t)rest * ntile + tile) * ((uint64_t)T << LOGF)       // rest = part*nseq + seq
                             : (part0 + part_of(rest)) * in.part_step + tile * T;
    if (FB_DBG(g) & 2) {     // ablation only; hoisted so that the real path has no per-load branch
#pragma unroll
      for (int i = 0; i < NPAIR; i++) { RawW<RAWW> z; z.w[0] = tid + i; raw[i] = z; }
      return;
    }
    // element i of a thread's first-stage butterfly is row na = nab + i*MS of one column pair: sample index =
    // base + i*step with a wave-uniform step (no per-element index arithmetic or branches between the loads)
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const uint64_t step = pret ? ((uint64_t)MS << logT) : ((uint64_t)MS << g.logR);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;               // element of the group's half tile: row eb >> logTw, column eb % Tw
      const uint64_t tb = t0 + cofs + (eb & (Tw - 1)) + (pret ? (uint64_t)((eb >> logTw) << logT) : (((uint64_t)(eb >> logTw)) << g.logR));
#pragma unroll
      for (int i = 0; i < P::R1; i++) raw[(g2 / 2) * P::R1 + i] = fetch_pair<RAWW>(g, in, seq, tb + i * step);
    }
  };

  // exchange buffer(s), then the stage twiddle tables (16-byte aligned)
  const uint32_t nthr_w = SPLIT ? 256u : blockDim.x;             // threads of a group
  const uint32_t ldsH = lds_pad(PTS * nthr_w) + 8;               // words of one exchange buffer
  const uint32_t ltw_off = SPLIT ? 2 * ldsH : ldsH;
  ltw_fill<LOGF>(lds, ltw_off, tw, threadIdx.x, blockDim.x);
  cf* const lw = lds + grp * ldsH;                               // this group's exchange buffer
  const uint32_t ltw_w = ltw_off - grp * ldsH;                   // the shared tables, relative to it
  fb_stagger();
  fb_setprio();
  // copy-out of the staged tile (see the end of the tile loop): thread part of the addresses, once per kernel
  const uint32_t co_swz = (PTS * blockDim.x) >= 256 ? 1u : 0u;
  const uint32_t co_l0 = 2 * (SPLIT ? (threadIdx.x & 255u) : threadIdx.x);
  const uint32_t co_n2 = SPLIT ? 512u : LOGT >= 0 ? (2u << (LOGF + LOGT - LOG_PTS)) : 2 * blockDim.x;   // full tiles: a constant
  const int co_sh = logTw + logT2;
  const bool co_fast = (co_n2 & 63) == 0 && (co_n2 >> co_sh) != 0 && (co_n2 & ((1u << co_sh) - 1)) == 0;   // uniform
  const uint32_t co_lds = lds_pad(co_l0 ^ (((co_l0 >> 4) & co_swz) << 3)), co_lstep = co_n2 + ((co_n2 >> 6) << 2);
  const uint32_t co_goff = (uint32_t)(((((uint64_t)(co_l0 >> co_sh) << g.logR) << logT2) + (co_l0 & ((1u << co_sh) - 1))) * sizeof(cf));
  const uint64_t co_gstep = ((uint64_t)(co_n2 >> co_sh) << g.logR) << logT2;       // elements of A per pair step
  // copy-out of a staged tile (all of the group's threads; `tid` is the thread of the group)
  auto copy_out = [&](const uint32_t tile, cf* __restrict__ Aseq) {
    const uint32_t swz = co_swz;
    if (!(FB_DBG(g) & 1)) {
    const uint32_t nthr = nthr_w;
    if (co_fast) {
      // pair jj of a thread is pair 0 plus jj*2*nthr elements: a constant step in the padded image (co_lstep) and a
      // uniform step in A (co_gstep) -- one LDS address and one 32-bit global offset per THREAD, computed before the
      // tile loop; the per-pair part is an immediate / a scalar-register base (this loop issued 23 % of the pass's
      // vector instructions as per-pair address arithmetic, 64-bit shifts included)
      const char* __restrict__ gb = (const char*)(Aseq + ((uint64_t)(tile * T + cofs) << logT2));
#pragma unroll
      for (int j4 = 0; j4 < PTS / 2; j4 += 4) {                  // four LDS reads in flight, then their stores
        float4 pr[4];
#pragma unroll
        for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lw[co_lds + (j4 + q) * co_lstep];
        __builtin_amdgcn_sched_barrier(0);                         // (the min-register scheduler would pair every read with its store)
#pragma unroll
        for (int q = 0; q < 4; q++) st_stream((float4*)(gb + (uint64_t)(j4 + q) * co_gstep * sizeof(cf) + co_goff), pr[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 4
      for (int jj = 0; jj < PTS / 2; jj++) {
        const uint32_t l = 2 * (tid + jj * nthr);                  // element index inside the staged image
        const uint32_t blkA = l >> (logTw + logT2), within = l & ((1u << (logTw + logT2)) - 1);
        const float4 pr = *(const float4*)&lw[lds_pad(l ^ (((l >> 4) & swz) << 3))];
        st_stream((float4*)&Aseq[((((uint64_t)blkA << g.logR) + tile * T + cofs) << logT2) + within], pr);
      }
    }
  }
  };
  [[maybe_unused]] uint32_t co_tile = 0;
  [[maybe_unused]] cf* co_Aseq = nullptr;
  [[maybe_unused]] bool co_pending = false;
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  RawW<RAWW> raw[PTS / 2];
  fetch(item, raw);
#if defined(FB_STAMPS) && FB_STAMPS == 1
  unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, acc_s[6] = {0, 0, 0, 0, 0, 0};
  STAMP(ts5);
#endif
  if (SPLIT && grp == 1) __builtin_amdgcn_s_barrier();         // the second group runs one barrier phase behind the first
  for (;;) {
    asm volatile("" : "+v"(tid));   // per-tile index math stays inside the loop (see wgfft)
    cx2 x[NPAIR];
#if defined(FB_STAMPS) && FB_STAMPS == 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(ts0);
#endif
    const uint32_t seq_cur = seq_of(item >> logNt);
#pragma unroll
    for (int h = 0; h < NPAIR; h++) {
      cf a, b;
      decode_pair<RAWW>(g, in, raw[h], a, b,
                        RAWW == 2 ? (cofs + ((P::G1 * tid + 2 * (h / P::R1)) & (Tw - 1))) : seq_cur);
      x[h] = make_cx2(a, b);
    }
#if defined(FB_STAMPS) && FB_STAMPS == 1
    STAMP(ts1);
#endif
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    if (more) fetch(next, raw);
#if FB_DEFER_CO
    // The PREVIOUS tile is copied out here, in front of this tile's first butterfly stage -- which works in registers
    // until its exchange barrier -- so that the 128 KB of stores drain while the vector unit computes, instead of in a phase
    // of their own at the end of the tile with the vector unit idle (the staged image is not touched before that barrier)
    if (co_pending) copy_out(co_tile, co_Aseq);
#endif
#if defined(FB_STAMPS) && FB_STAMPS == 1
    STAMP(ts2);
#endif

    const uint32_t tile = item & (ntile - 1);
    cf* __restrict__ Aseq = A + (uint64_t)(item >> logNt) * L;                 // sequence part*nseq + seq
    // last-stage outputs go to LDS in A-layout order [ka/T2][col][ka%T2]; after a barrier the tile is
    // written out as whole runs of T*T2 elements with 16-byte-per-lane stores.  The image is XOR-swizzled
    // (bit 3 ^= bit 4; pairs of elements stay together) so that the 8-byte scatter of a wave spreads over all
    // banks (17 % of this pass's LDS cycles were bank conflicts, profiles/r01d_lds_conflicts.txt)
    const uint32_t swz = (PTS * blockDim.x) >= 256 ? 1u : 0u;
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
#if !FB_TWIDDLE_IN_P2
      const uint32_t nb = tile * T + cofs + col;
      if (!(FB_DBG(g) & 8)) apply_pass_twiddle<R>(v, nb, p, pstride, logL, tw, g.tw_lo);
#endif
      // image (of the group's columns) index of element k: l0 + k*(pstride << logTw) (pstride is a multiple of T2), so when
      // that step is a multiple of 64 the swizzle and the padding of l0 carry over: one address per column, constant offsets
      auto img = [&](const uint32_t l) { return lds_pad(l ^ (((l >> 4) & swz) << 3)); };
      const uint32_t l0 = ((((p >> logT2) << logTw) + col) << logT2) | (p & (T2 - 1));
      const uint32_t step = pstride << logTw;
      const bool aff = (step & 63) == 0 && (pstride & (T2 - 1)) == 0;
      const uint32_t b0 = img(l0), b1 = img(l0 + T2), sp = step + (step >> 4);
      if (aff) {                                       // uniform
#pragma unroll
        for (int k = 0; k < R; k++) {
          float* __restrict__ d0 = (float*)&lw[b0 + k * sp];
          float* __restrict__ d1 = (float*)&lw[b1 + k * sp];
          d0[0] = v[k].x[0]; d0[1] = v[k].y[0];       // (re, im) of column col   (two dwords: no register shuffling)
          d1[0] = v[k].x[1]; d1[1] = v[k].y[1];       // column col + 1
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t ka = k * pstride + p;
          const uint32_t l = ((((ka >> logT2) << logTw) + col) << logT2) | (ka & (T2 - 1));
          lw[img(l)] = cx2_lo(v[k]);
          lw[img(l + T2)] = cx2_hi(v[k]);
        }
      }
    };
#ifdef FB_P1_DIRECT   // experiment: the last stage stores its outputs straight from registers (8 bytes per lane, 64-byte runs),
                      // no staging exchange through LDS
    auto store_direct = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      cf* __restrict__ ab = Aseq + ((uint64_t)(tile * T + col) << logT2);
#pragma unroll
      for (int k = 0; k < R; k++) {
        const uint32_t ka = k * pstride + p;
        cf* o = ab + (((uint64_t)(ka >> logT2) << g.logR) << logT2) + (ka & (T2 - 1));
        st_stream(o, cx2_lo(v[k]));
        st_stream(o + T2, cx2_hi(v[k]));
      }
    };
    wgfft<LOGF, -1, false>(lds, ltw_off, tid, logT, x, store_direct);
    if (!more) break;
    item = next;
    continue;
#endif
    if (FB_DBG(g) & 4) wgfft_passthrough<LOGF>(tid, logTw, x, store);
    else wgfft<LOGF, -1, true>(lw, ltw_w, tid, logTw, x, store);
    __syncthreads();
#if defined(FB_STAMPS) && FB_STAMPS == 1
    STAMP(ts3);
#endif
#if FB_DEFER_CO
    co_tile = tile; co_Aseq = Aseq; co_pending = true;          // copied out at the top of the next tile (or behind the loop)
#else
    copy_out(tile, Aseq);
#endif
#if defined(FB_STAMPS) && FB_STAMPS == 1
    STAMP(ts4);
    acc_s[0] += ts0 - ts5; acc_s[1] += ts1 - ts0; acc_s[2] += ts2 - ts1; acc_s[3] += ts3 - ts2; acc_s[4] += ts4 - ts3; acc_s[5] += 1;
    ts5 = ts4;
#endif
    if (!more) break;
    item = next;
  }
#if FB_DEFER_CO
  if (co_pending) copy_out(co_tile, co_Aseq);
#endif
  if (SPLIT && grp == 0) __builtin_amdgcn_s_barrier();         // the second group's last barrier
#if defined(FB_STAMPS) && FB_STAMPS == 1
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 6; q++) atomicAdd(&g_stamps[blockIdx.x][q], acc_s[q]);
#endif
}

// ------------------------------------------------------------------------------------ P1, paired tiles (round 3)
// Long transforms (L = Fa*Fb >= 2^25: -F 64:D at the optimal response length, dsp::Convolution shapes) leave pass 1 tiles of
// T1 = 2 columns and pass 2 tiles of T2 = 4 rows, so a pass-1 tile meets a pass-2 tile in T1*T2 = 8 elements: 64-byte runs
// of A, half a cache line per store run -- pass 1 then moves its bytes at 2.7 TB/s where the same bytes in 256-byte runs
// (headline geometry) go at 5.0 (tools/run_length_probe.hip: stores in 64-byte runs 3.2-3.4 TB/s at any stride, 128-byte runs
// 4.6-4.8).  Here a work item is a PAIR of adjacent tiles (columns 4j .. 4j+3): the two are transformed one after the other,
// the outputs of the first wait in registers (64) while the second runs through the one exchange buffer, and the four
// columns are then staged and copied out together, half the rows at a time (the buffer holds 2^14 elements: 4 columns x
// Fa/2 rows) -- runs of 2*T1*T2 elements, whole 128-byte lines.  The layout of A and everything behind it are unchanged.
// Full-size tiles of two columns only (Fa = 2^13 at 2^14 points per workgroup): the last stage is the radix-2 one, so the two
// outputs of a butterfly are row ka (lower half) and ka + Fa/2 (upper half).
struct KeepOut {
  cx2* o;
  uint32_t p0;
  int h;
  template <int R> DEV void operator()(const uint32_t, const uint32_t p, const uint32_t, cx2 (&v)[R])
  {
    static_assert(R == 2, "k_fwd_cols_dual: radix-2 last stage");
    if (h == 0) p0 = p;
    o[2 * h] = v[0];
    o[2 * h + 1] = v[1];
  }
};
template <int RAWW>
__global__ __launch_bounds__(512) void k_fwd_cols_dual(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                       const cf* __restrict__ tw, const uint64_t part0,
                                                       const uint32_t nparts, const uint32_t nseq, const uint32_t run)
{
  constexpr int LOGF = 13, LOGT = 1;
  typedef FftPlan<LOGF> P;
  static_assert(FB_TWIDDLE_IN_P2, "k_fwd_cols_dual: the inter-pass twiddle belongs to pass 2");
  static_assert(P::REM == 1 && PTS / 2 / 2 == 8, "k_fwd_cols_dual: 2^13-point columns, radix-2 last stage");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logT2 = g.logT2;
  const uint32_t T = 2, T2 = 1u << logT2;
  const int logL = LOGF + g.logR;
  const uint64_t L = 1ull << logL;
  const int logNp = g.logR - LOGT - 1;                  // pairs of tiles per sequence
  const uint32_t npair = 1u << logNp, ntile = npair << 1;
  const uint32_t total = npair * nseq * nparts;
  auto seq_of = [&](const uint32_t rest) { return nseq == 2 ? (rest & 1u) : 0u; };
  auto part_of = [&](const uint32_t rest) { return nseq == 2 ? (rest >> 1) : rest; };
  auto fetch = [&](const uint32_t item, const uint32_t sub, RawW<RAWW> (&raw)[PTS / 2]) {
    const uint32_t tile = ((item & (npair - 1)) << 1) | sub;
    const uint32_t rest = item >> logNp;
    const uint32_t seq = seq_of(rest);
    const bool pret = in.kind == 3 || in.kind == 5;   // pre-transposed: [part][tile][na][T] pairs, contiguous per tile
    const uint64_t t0 = pret ? ((uint64_t)rest * ntile + tile) * ((uint64_t)T << LOGF) : (part0 + part_of(rest)) * in.part_step + tile * T;
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const uint64_t step = pret ? ((uint64_t)MS << LOGT) : ((uint64_t)MS << g.logR);
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;               // row eb >> 1, column eb & 1 (= 0)
      const uint64_t tb = t0 + (eb & (T - 1)) + (pret ? (uint64_t)((eb >> LOGT) << LOGT) : (((uint64_t)(eb >> LOGT)) << g.logR));
#pragma unroll
      for (int i = 0; i < P::R1; i++) raw[(g2 / 2) * P::R1 + i] = fetch_pair<RAWW>(g, in, seq, tb + i * step);
    }
  };
  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, threadIdx.x, blockDim.x);
  // staged image of one half: A-layout order l = ((ka_local / T2) * 4 + column) * T2 + ka % T2, padded by two elements
  // per 32 (a thread stages 32 consecutive elements: with the exchange buffer's padding of 4 per 64 the lanes of a wave would
  // meet in 8 banks); same size as the exchange buffer
  auto img = [](const uint32_t l) { return l + ((l >> 5) << 1); };
  // copy-out: 16-byte unit u = tid + 512*jj -> image element 2u, A element ((l >> sh) << logR << logT2) + (l & mask)
  const int sh = 2 + logT2;
  const uint32_t co_l0 = 2 * threadIdx.x;
  const uint32_t co_lds = img(co_l0), co_lstep = img(1024);                                   // 1024 is a multiple of 32
  const uint64_t co_goff = (((uint64_t)(co_l0 >> sh) << g.logR) << logT2) + (co_l0 & ((1u << sh) - 1));
  const uint64_t co_gstep = ((uint64_t)(1024u >> sh) << g.logR) << logT2;
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  RawW<RAWW> raw0[PTS / 2], raw1[PTS / 2];
  fetch(item, 0, raw0);
  fetch(item, 1, raw1);
  for (;;) {
    asm volatile("" : "+v"(tid));
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    const uint32_t seq_cur = seq_of(item >> logNp);
    cx2 o0[PTS / 2], o1[PTS / 2];
    uint32_t p0;
    {
      cx2 x[NPAIR];
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        cf a, b;
        decode_pair<RAWW>(g, in, raw0[h], a, b, seq_cur);
        x[h] = make_cx2(a, b);
      }
      if (more) fetch(next, 0, raw0);
      KeepOut keep{o0, 0u, 0};
      wgfft<LOGF, -1, false>(lds, ltw_off, tid, LOGT, x, keep);
      p0 = keep.p0;
    }
    {
      cx2 x[NPAIR];
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        cf a, b;
        decode_pair<RAWW>(g, in, raw1[h], a, b, seq_cur);
        x[h] = make_cx2(a, b);
      }
      if (more) fetch(next, 1, raw1);
      KeepOut keep{o1, 0u, 0};
      wgfft<LOGF, -1, false>(lds, ltw_off, tid, LOGT, x, keep);
    }
    const uint32_t pair = item & (npair - 1);
    cf* __restrict__ Aseq = A + (uint64_t)(item >> logNp) * L + ((uint64_t)(pair * 4) << logT2);
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      __syncthreads();                 // the exchange buffer (second transform's last stage / the other half's copy-out) has been read
#pragma unroll
      for (int h = 0; h < PTS / 4; h++) {
        const uint32_t p = p0 + h;                                             // row of the half
        const uint32_t l = (((p >> logT2) << 2) << logT2) | (p & (T2 - 1));     // column 0 of the four
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          const cx2 v = s2 ? o1[2 * h + hh] : o0[2 * h + hh];
          float* __restrict__ d0 = (float*)&lds[img(l + ((2 * s2) << logT2))];
          float* __restrict__ d1 = (float*)&lds[img(l + ((2 * s2 + 1) << logT2))];
          d0[0] = v.x[0]; d0[1] = v.y[0];
          d1[0] = v.x[1]; d1[1] = v.y[1];
        }
      }
      __syncthreads();
      if (!(FB_DBG(g) & 1)) {
        const char* __restrict__ gb = (const char*)(Aseq + (uint64_t)hh * (L >> 1) + co_goff);
#pragma unroll
        for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
          float4 pr[4];
#pragma unroll
          for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[co_lds + (j4 + q) * co_lstep];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; q++) st_stream((float4*)(gb + (uint64_t)(j4 + q) * co_gstep * sizeof(cf)), pr[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (!more) break;
    item = next;
  }
}
// ------------------------------------------------------------------------------------ nchan_subband = 3 * 2^k, 5 * 2^k
// dsp::Filterbank takes whatever length FFTW / cuFFT plans (Filterbank.C:107-155, FilterbankCUDA.cu:92-116), e.g. -F 96:D.
// Here the transform tiles are powers of two; a forward transform of L = R * L' points (R = 3 or 5) is computed as R
// interleaved sub-sequences w_c[m] = w[R m + c] -- each an ordinary power-of-two forward transform F_c (passes 0-2 unchanged) --
// and one radix-R step:  X[k + q L'] = sum_c W_R^(c q) W_L^(c k) F_c[k],  k < L', q < R.  Bin k + q L' lies in spectrum row
// q * Rr' + k / M: the R combined bands, stored one after the other in the power-of-two X layout, ARE the R * Rr' rows the
// inverse pass walks (k_inv_chan: rows nsub << logR).
//   k_sub_split   : the launch group's samples de-interleaved into R contiguous single-channel blocks (generic byte order /
//                   float rows), so that passes 0-2 see ordinary inputs
//   k_sub_combine : the radix-R step in place on the R sub-spectra of every (part, sequence)
__global__ __launch_bounds__(256) void k_sub_split(const SubSplit p, uint8_t* __restrict__ out)
{
  const uint64_t n = p.nper * p.R;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t j = i / p.R;
    const uint32_t c = (uint32_t)(i - j * p.R);
    const uint64_t t = p.t_first + i;                                       // = t_first + R*j + c
    uint8_t* __restrict__ o = out + (uint64_t)c * p.sub_stride;
    if (p.kind == 0) {                                                      // float rows -> [pol][j][ndim] floats
      const float* __restrict__ x = (const float*)p.base + p.chan_off;
      float* __restrict__ of = (float*)o;
      for (uint32_t q = 0; q < p.npol; q++)
        for (uint32_t d = 0; d < p.ndim; d++) of[(q * p.nper + j) * p.ndim + d] = x[q * p.pol_stride + t * p.ndim + d];
    } else if (p.kind == 2) {                                               // CASPSR 4 B pol0 | 4 B pol1 -> (p0, p1) pairs
      const uint8_t* __restrict__ b = (const uint8_t*)p.base + (t >> 2) * 8 + (t & 3);
      o[2 * j] = b[0];
      o[2 * j + 1] = b[4];
    } else {                                                                // generic: byte ((t*nchan + c)*npol + p)*ndim + d
      const uint32_t es = p.npol * p.ndim;
      const uint8_t* __restrict__ b = (const uint8_t*)p.base + (t * p.nchan + p.ichan) * es;
      for (uint32_t q = 0; q < es; q++) o[j * es + q] = b[q];
    }
  }
}

// MSUB (freq_res = R * 2^k): the combined spectrum goes to a second buffer in PSEUDO-CHANNEL order -- bin R m' + r of channel c is
// bin m' of row c*R + r -- and, for real input, the mirror bins L - k where the inverse pass looks for them: row Rr-1-s, bin
// M' - m' (m' >= 1), row Rr - s, bin 0 (m' = 0).  mo = the caller's freq_res (R * M').
// rm = the factor of freq_res (the radix R of this kernel is nsub = rm times the odd factor of nchan_subband).
template <int R, bool MSUB>
__global__ __launch_bounds__(256) void k_sub_combine(const FbGeom g, cf* __restrict__ X, const uint32_t nseqs /* parts x sequences */,
                                                     cf* __restrict__ Xout, const uint32_t mo, const uint32_t rm)
{
  const int logLs = g.logM + g.logR;                     // sub-sequence length L'
  const uint32_t Ls = 1u << logLs, L = Ls * R;
  const uint32_t X3m = (1u << g.logX3) - 1, Mm = (1u << g.logM) - 1;
  cf wr[R];                                              // W_R^j
#pragma unroll
  for (int j = 0; j < R; j++) {
    const float x = (float)j / (float)R;
    wr[j] = make_float2(__builtin_amdgcn_cosf(x), -__builtin_amdgcn_sinf(x));
  }
  const uint64_t n = (uint64_t)nseqs << logLs;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t o = (uint32_t)(i & (Ls - 1));
    cf* __restrict__ base = X + (i >> logLs) * (uint64_t)L + o;
    // X layout: offset o = ((s' >> logX3) * M + m) << logX3 | s' % X3  ->  bin k = s' * M + m
    const uint32_t t = o >> g.logX3, m = t & Mm, sp = ((t >> g.logM) << g.logX3) | (o & X3m);
    const uint32_t k = (sp << g.logM) + m;
    cf gq[R];
    gq[0] = base[0];
#pragma unroll
    for (int c = 1; c < R; c++) {
      // W_L^(c k): c k mod L = a L' + b -> a / R + (b / L') / R revolutions (b / L' is exact)
      const uint32_t ck = (uint32_t)(((uint64_t)c * k) % L), a = ck >> logLs, b = ck & (Ls - 1);
      const float x = ((float)a + (float)b * __uint_as_float((uint32_t)(127 - logLs) << 23)) / (float)R;
      const cf w = make_float2(__builtin_amdgcn_cosf(x), -__builtin_amdgcn_sinf(x));
      gq[c] = cmul(base[(uint64_t)c << logLs], w);
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
      cf acc = gq[0];
#pragma unroll
      for (int c = 1; c < R; c++) {
        const cf v = cmul(gq[c], wr[(c * q) % R]);
        acc.x += v.x; acc.y += v.y;
      }
      if constexpr (!MSUB) {
        base[(uint64_t)q << logLs] = acc;
      } else {
        const uint32_t kk = k + ((uint32_t)q << logLs);                      // natural bin of the whole transform
        const uint32_t Rr = (uint32_t)R << g.logR, N = g.real_input ? L >> 1 : L;
        const bool up = kk > N;                                               // (real input) a mirror bin
        const uint32_t kq = up ? L - kk : kk;
        const uint32_t cc = kq / mo, mm = kq - cc * mo, mi = mm / rm, r = mm - mi * rm, s = cc * rm + r;
        uint32_t row, bin;
        if (kk == N && g.real_input) { row = Rr >> 1; bin = 0; }              // (never read: the slot nothing else uses)
        else if (!up) { row = s; bin = mi; }
        else if (mi) { row = Rr - 1 - s; bin = (1u << g.logM) - mi; }
        else { row = Rr - s; bin = 0; }
        Xout[(i >> logLs) * (uint64_t)L + (((((uint64_t)(row >> g.logX3) << g.logM) + bin) << g.logX3) | (row & X3m))] = acc;
      }
    }
  }
}
#endif  // FB_HAS(1)

#if FB_HAS(2)
// ------------------------------------------------------------------------------------ P2
// Rr-point forward FFTs along T2 adjacent rows ka of A (one contiguous block) -> spectrum rows
// s' = kb, bin m = ka, stored as X[s'/T3][m][s'%T3].
template <int LOGF, int LOGT>
__global__ __launch_bounds__(512) void k_fwd_rows(const FbGeom g, const cf* __restrict__ A, cf* __restrict__ X,
                                                  const cf* __restrict__ tw, const uint32_t nparts,
                                                  const uint32_t nseq, const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logT = LOGT >= 0 ? LOGT : g.logT2, logT3 = g.logX3;     // X layout block factor
  const uint32_t T2 = 1u << logT, T3 = 1u << logT3;
  const uint64_t L = 1ull << (g.logM + LOGF);
  const uint32_t ntile = 1u << (g.logM - logT);
  const uint32_t total = ntile * nseq * nparts;
  const int logNt = g.logM - logT;          // ntile = 2^logNt
  // The sequences of the launch are walked backwards: the parts written last -- the ones still in the Infinity Cache
  // when the launch ends -- are then the ones the inverse pass, which walks the parts forwards, meets first
  // (+0.8 % Msamples/s in three alternating runs; DSPSR_AMD_DEBUG bit 128 restores the forward order)
  auto seq_of = [&](const uint32_t item) -> uint64_t {
    const uint32_t sq = item >> logNt;
    return (FB_DBG(g) & 128) ? sq : nseq * nparts - 1 - sq;
  };

  // the prefetch keeps the loaded 16-byte pairs untouched (any use would wait for the loads at once);
  // they are rearranged into split form when the tile is started
  auto fetch = [&](const uint32_t item, float4 (&y)[NPAIR]) {
    const uint32_t tile = item & (ntile - 1);
    const cf* __restrict__ Ablk = A + seq_of(item) * L + (((uint64_t)tile << LOGF) << logT);     // g.logR == LOGF
    if (FB_DBG(g) & 2) {     // ablation only; hoisted so that the real path has no per-load branch (and vmcnt(0))
#pragma unroll
      for (int i = 0; i < NPAIR; i++) y[i] = make_float4(tid, i, 1.f, 1.f);
      return;
    }
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++)
        y[(g2 / 2) * P::R1 + i] = ld_stream((const float4*)&Ablk[first_stage_elem<LOGF>(tid, logT, g2, i)]);
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;      // behind the exchange buffer (16-byte aligned)
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  fb_stagger();
  fb_setprio();
  // copy-out of the staged tile (end of the tile loop): thread part of the addresses, once per kernel
  const uint32_t co_swz = (PTS * blockDim.x) >= 256 ? 3u : 0u;
  const uint32_t co_l0 = 2 * threadIdx.x, co_n2 = LOGT >= 0 ? (2u << (LOGF + LOGT - LOG_PTS)) : 2 * blockDim.x;   // full tiles: a constant
  const int co_sh = logT + logT3;
  const bool co_fast = (co_n2 & 63) == 0 && (g.xblocked || ((co_n2 >> co_sh) != 0 && (co_n2 & ((1u << co_sh) - 1)) == 0));   // uniform
  const uint32_t co_lds = lds_pad(co_l0 ^ (((co_l0 >> 4) & co_swz) << 1)), co_lstep = co_n2 + ((co_n2 >> 6) << 2);
  const uint32_t co_goff = (uint32_t)((g.xblocked ? (uint64_t)co_l0
                                                   : ((((uint64_t)(co_l0 >> co_sh) << g.logM) << logT3) + (co_l0 & ((1u << co_sh) - 1)))) * sizeof(cf));
  const uint64_t co_gstep = g.xblocked ? (uint64_t)co_n2 : (((uint64_t)(co_n2 >> co_sh) << g.logM) << logT3);   // elements of X per pair step
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  float4 y[NPAIR];
  fetch(item, y);
#if defined(FB_STAMPS) && FB_STAMPS == 2
  unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, acc_s[6] = {0, 0, 0, 0, 0, 0};
  STAMP(ts5);
#endif
  for (;;) {
    asm volatile("" : "+v"(tid));   // per-tile index math stays inside the loop (see wgfft)
    cx2 x[NPAIR];
#if defined(FB_STAMPS) && FB_STAMPS == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(ts0);
#endif
#pragma unroll
    for (int i = 0; i < NPAIR; i++) x[i] = make_cx2(make_float2(y[i].x, y[i].y), make_float2(y[i].z, y[i].w));
#if FB_TWIDDLE_IN_P2
    // The twiddle W_L^{nb*ka} between the two forward passes is applied HERE, to the elements pass 2 has just loaded, not
    // to pass 1's outputs: pass 1 is bound by the vector instructions it issues (31 packed complex products and 10 sin/cos
    // per thread and tile for this twiddle alone), pass 2 by the fabric with its vector unit two thirds idle.  The product
    // nb*ka is symmetric: the column pair is (ka, ka + 1), the position nb = pos0 + i*S.
    if (!(FB_DBG(g) & 8)) {
      const uint32_t tile_t = item & (ntile - 1);
      constexpr uint32_t S = 1u << (LOGF - P::LOGR1);
#pragma unroll
      for (int g2 = 0; g2 < P::G1; g2 += 2) {
        const uint32_t eb = P::G1 * tid + g2;
        cx2 (&xg)[P::R1] = *reinterpret_cast<cx2 (*)[P::R1]>(&x[(g2 / 2) * P::R1]);
        apply_pass_twiddle<P::R1>(xg, tile_t * T2 + (eb & (T2 - 1)), eb >> logT, S, g.logM + LOGF, tw, g.tw_lo);
      }
    }
#endif
#if defined(FB_STAMPS) && FB_STAMPS == 2
    STAMP(ts1);
#endif
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    if (more) fetch(next, y);
#if defined(FB_STAMPS) && FB_STAMPS == 2
    STAMP(ts2);
#endif

    const uint32_t tile = item & (ntile - 1);
    cf* __restrict__ Xseq = X + seq_of(item) * g.xstride;
    // last-stage outputs go to LDS in X-layout order [s'/T3][klo][s'%T3]; after a barrier the tile is
    // written out as whole runs of T2*T3 elements with 16-byte-per-lane stores.  XOR swizzle of the image
    // (bits 1,2 ^= bits 4,5) against bank conflicts of the 8-byte scatter (42 % of this pass's LDS cycles)
    const uint32_t swz = (PTS * blockDim.x) >= 256 ? 3u : 0u;
    auto store = [&](const uint32_t klo, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      auto img = [&](const uint32_t l) { return lds_pad(l ^ (((l >> 4) & swz) << 1)); };
      const uint32_t l0 = ((((p >> logT3) << logT) + klo) << logT3) | (p & (T3 - 1));
      const uint32_t step = pstride << logT;            // image index step per k (pstride is a multiple of T3)
      const bool aff = (step & 63) == 0 && (pstride & (T3 - 1)) == 0;
      const uint32_t b0 = img(l0), b1 = img(l0 + T3), sp = step + (step >> 4);
      if (aff) {                                       // uniform
#pragma unroll
        for (int k = 0; k < R; k++) {
          float* __restrict__ d0 = (float*)&lds[b0 + k * sp];
          float* __restrict__ d1 = (float*)&lds[b1 + k * sp];
          d0[0] = v[k].x[0]; d0[1] = v[k].y[0];
          d1[0] = v[k].x[1]; d1[1] = v[k].y[1];
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t srow = k * pstride + p;
          const uint32_t l = ((((srow >> logT3) << logT) + klo) << logT3) | (srow & (T3 - 1));
          lds[img(l)] = cx2_lo(v[k]);
          lds[img(l + T3)] = cx2_hi(v[k]);
        }
      }
    };
    if (FB_DBG(g) & 4) wgfft_passthrough<LOGF>(tid, logT, x, store);
    else wgfft<LOGF, -1, true>(lds, ltw_off, tid, logT, x, store);
    __syncthreads();
#if defined(FB_STAMPS) && FB_STAMPS == 2
    STAMP(ts3);
#endif
    if (!(FB_DBG(g) & 1)) {
      const uint32_t nthr = blockDim.x;
      if (co_fast) {
        // thread part of the addresses computed once per kernel, per-pair part an immediate / a uniform step (see pass 1)
        const char* __restrict__ gb = (const char*)(Xseq + (g.xblocked ? (uint64_t)tile * g.xblock : ((uint64_t)(tile * T2) << logT3)));
#pragma unroll
        for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
          float4 pr[4];
#pragma unroll
          for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[co_lds + (j4 + q) * co_lstep];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; q++) st_stream((float4*)(gb + (uint64_t)(j4 + q) * co_gstep * sizeof(cf) + co_goff), pr[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 4
        for (int jj = 0; jj < PTS / 2; jj++) {
          const uint32_t l = 2 * (tid + jj * nthr);
          const uint32_t blkX = l >> (logT + logT3), within = l & ((1u << (logT + logT3)) - 1);
          const float4 pr = *(const float4*)&lds[lds_pad(l ^ (((l >> 4) & swz) << 1))];
          // four-pass mode (xblocked): the tile's image [kb][ka % T2] IS its block of X -- one contiguous 2^14-element
          // store instead of runs of T2 elements scattered over the natural order
          const uint64_t xo = g.xblocked ? (uint64_t)tile * g.xblock + l
                                         : ((((uint64_t)blkX << g.logM) + tile * T2) << logT3) + within;
          st_stream((float4*)&Xseq[xo], pr);
        }
      }
    }
#if defined(FB_STAMPS) && FB_STAMPS == 2
    STAMP(ts4);
    acc_s[0] += ts0 - ts5; acc_s[1] += ts1 - ts0; acc_s[2] += ts2 - ts1; acc_s[3] += ts3 - ts2; acc_s[4] += ts4 - ts3; acc_s[5] += 1;
    ts5 = ts4;
#endif
    if (!more) break;
    item = next;
  }
#if defined(FB_STAMPS) && FB_STAMPS == 2
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 6; q++) atomicAdd(&g_stamps[blockIdx.x][q], acc_s[q]);
#endif
}

#endif  // FB_HAS(2)

// LDS-DMA of one 16-byte plan entry per lane, global -> LDS without passing through registers (lane l of the wave lands at
// `lds_wave_base` + 16*l), issued from inline assembly: the compiler does not see a vector-memory operation, so it does NOT put
// `s_waitcnt vmcnt(0)` in front of the next barrier.  With __builtin_amdgcn_global_load_lds it did -- in the middle of the
// transform, where that wait also drained the whole prefetch of the next tile, issued just before (ISA of round 3's
// k_inv_chan<12,true,2>: global_load_lds_dwordx4 ... s_waitcnt vmcnt(0); s_barrier between the second and the third stage;
// the stamps of profiles/r03_experiments.txt item 4 show the transform phase 1.8k cycles longer for it).  The hardware needs
// no such wait: a barrier does not drain vector memory (MI355X_MICROARCH.md, "Two waves per SIMD" item 7); what orders a reader
// behind the DMA is the issuing wave's covering vmcnt wait plus a barrier, and the callers have both: every tile begins with an
// explicit `s_waitcnt vmcnt(0)` and the entries are read behind the tile's first exchange barrier.  An operation the compiler
// does not count only makes its own counted waits more conservative (the counter is in order).  m0 (the LDS base of the DMA)
// is saved and restored inside the block.
DEV void lds_dma_b128(const void* gsrc, const uint32_t lds_wave_base)
{
  const uint32_t sb = __builtin_amdgcn_readfirstlane(lds_wave_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(sb) : "memory");
}
DEV uint32_t lds_byte_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }

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

#if FB_HAS(3)
// freq_res = R * 2^k, last step: y[n] = sum_r exp(+2 pi i r n / freq_res) y_r[n mod M'] for the kept samples n of every channel and
// part, from the pseudo-channels' whole transforms Y[c*R + r][pol][part][M'] (written by the inverse pass as complex rows), into
// the caller's output: complex rows (kind 1) or detected samples (kind 2; Detection.C:423-474 layouts as in k_inv_chan).
template <int R>
__global__ __launch_bounds__(256) void k_time_combine(const TimeCombine p, const FbOut out)
{
  const uint32_t Mi = 1u << p.logMi;
  const uint64_t n = (uint64_t)p.nparts * p.C * p.nkeep;
  const float inv_mo = 1.0f / (float)p.mo;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(i % p.nkeep);
    const uint64_t pc = i / p.nkeep;
    const uint32_t c = (uint32_t)(pc % p.C), lp = (uint32_t)(pc / p.C);
    const uint32_t nn = p.nfilt_pos + t, ni = nn & (Mi - 1);
    cf a = make_float2(0.f, 0.f), b = a;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const cf* __restrict__ y = p.Y + (uint64_t)(c * R + r) * p.y_chan_stride + ((uint64_t)lp << p.logMi) + ni;
      cf w = make_float2(1.f, 0.f);
      if (r) {
        const float x = (float)((uint32_t)((uint64_t)r * nn % p.mo)) * inv_mo;       // revolutions
        w = make_float2(__builtin_amdgcn_cosf(x), __builtin_amdgcn_sinf(x));
      }
      const cf v0 = cmul(y[0], w);
      a.x += v0.x; a.y += v0.y;
      if (p.npol == 2) { const cf v1 = cmul(y[p.y_pol_stride], w); b.x += v1.x; b.y += v1.y; }
    }
    const uint64_t part = p.part0 + lp;
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
#endif

#if FB_HAS(3) || FB_HAS(5)
// ------------------------------------------------------------------------------------ P3

// T3 output channels (both polarisations) of one part: Hermitian split of spectrum rows s and
// Rr-1-s into the two polarisations (real input), x chirp, inverse M-point FFT, keep window,
// complex output or fused detection.  Columns are (channel, pol) pairs: col = 2*slo + pol, so
// the two butterflies a thread owns are the two polarisations of the same (channel, bin):
// one (a, b, chirp) load serves both and detection needs no cross-lane traffic.
// Items: part fastest, so one XCD re-reads a tile's chirp rows from its L2 for every part.
// FOLD: the detected samples of the tile (T3 channels x nkeep samples, one float4 each) are staged in the
// exchange buffer instead of being written out, and folded at once: thread b owns phase bins b, b + blockDim, ...
// of the tile's channels, loads each touched accumulator from the device profile, adds the samples of the
// bin's intervals one by one in time order and stores it back.  A workgroup processes ALL parts of a tile in
// order and launches are stream ordered, so every (chan, bin) sum has the association order of the CPU loop
// Fold.C:844-852, exactly as the stand-alone fold kernel (fold.hip) -- bit-identical results, without the
// 16 B/sample round trip of the detected time series through HBM.
template <int LOGF, bool FOLD, int LOGT>
__global__ __launch_bounds__(512) void k_inv_chan(const FbGeom g, const cf* __restrict__ X,
                                                  const cf* __restrict__ kernel, const FbOut out,
                                                  const cf* __restrict__ tw, const uint64_t part0,
                                                  const uint32_t nparts, const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logT3 = LOGT >= 1 ? LOGT - 1 : g.logT3;
  const int logT = logT3 + 1;
  const uint32_t T = 1u << logT, T3 = 1u << logT3, M = 1u << LOGF, Rr = g.nsub << g.logR;     // (nsub = 3, 5: not a power of two)
  const uint64_t L = (uint64_t)M * Rr;
  const uint32_t nseq = g.real_input ? 1 : g.npol;
  const int logX3 = g.logX3;                    // X layout: element (row, m) at ((row >> logX3)*M + m) << logX3 | row % X3
  const uint32_t X3 = 1u << logX3;
  const uint32_t ntile = g.C >> logT3;
  auto xi = [&](const uint32_t row, const uint32_t m) -> uint64_t {
    return ((((uint64_t)(row >> logX3) << LOGF) + m) << logX3) | (row & (X3 - 1));
  };
  struct Abk { cf a, b; };   // the chirp is fetched at the start of the item (keeps the prefetch at 64 registers)

  // chunk < 0: all elements; otherwise the elements i with i % NCHUNK == chunk (the prefetch of the next tile is
  // issued in NCHUNK groups spread over the transform, see wgfft_stage)
  constexpr int NCHUNK = P::NS + 1;
  // T = 4 columns = 2 channels x 2 polarisations per tile: the two channels' elements are loaded as aligned 16-byte pairs and
  // the halves exchanged between the lane pair.  -3.7 % where the detected or complex output is written; in the fused
  // kernel it cost 1.6 % while the chirp was still loaded per part (round 1) and gains 5.7 % now that it stays in registers
  // (profiles/r02_experiments.txt, item 23)
  constexpr bool PAIR16 = LOGT == 2 && P::G1 == 2;
  const bool pair16 = PAIR16 && g.real_input && logX3 == 1 && !getenv_pair16_off(g);
  cf special = make_float2(0.f, 0.f);                   // mirror element of bin 0 (pair16 path)
  // a work item = (tile of channels, part of the launch), kept as two 32-bit numbers: a combined 64-bit index costs a
  // software 64-bit division per use (about 300 scalar instructions per tile in the r02c listing)
  struct Item { uint32_t tile, lp; };
  auto fetch = [&](const Item item, Abk (&raw)[PTS / 2], const int chunk) {
    const uint32_t tile = item.tile;
    const cf* __restrict__ X0s = X + (uint64_t)item.lp * nseq * L;
    if (FB_DBG(g) & 2) {     // ablation only; hoisted so that the real path has no per-load branch
      if (chunk <= 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; i++) { raw[i].a = make_float2(tid, i); raw[i].b = raw[i].a; }
      }
      return;
    }
    // element i of a thread's first-stage butterfly is bin m = mb + i*MS of one (channel, pol pair) column, so
    // every address is a base plus a multiple of a wave-uniform step: no per-element index arithmetic, no
    // divergent code between the loads (the m = 0 mirror element, the only irregular one, can only be i = 0)
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const int64_t step = (int64_t)MS << logX3;
    if constexpr (PAIR16) {
      if (pair16) {
        // Two channels per tile and X3 = 2: the elements of lanes 2j (channel 0) and 2j+1 (channel 1) for the same bin
        // are one aligned 16-byte pair, in both streams.  Lane parity q loads the pairs of the elements i = 2u + q --
        // 16 B per lane, half the load instructions (8-byte-per-lane streams run at 5.6 TB/s, 16-byte ones at 7.1 on this
        // chip, tools/load_width_probe.hip); the halves are exchanged between the two lanes when the tile is consumed.
        const uint32_t q = tid & 1, mb = tid >> 1;
        const uint32_t m0 = mb + q * MS;                                  // bin of element i = q
        const cf* __restrict__ pa2 = X0s + ((((uint64_t)tile << LOGF) + m0) << 1);
        const uint64_t rowb = (uint64_t)((Rr >> 1) - 1 - tile) << LOGF;   // row pair of the mirror rows Rr-1-s
        const cf* __restrict__ pb2 = X0s + ((rowb + (M - m0)) << 1);        // mirror bin M - m of element i = q
#pragma unroll
        for (int u = 0; u < P::R1 / 2; u++) {
          const float4 A = ld_stream((const float4*)(pa2 + (int64_t)u * 4 * MS));
          // bin 0 has its own mirror (loaded below): its pair would lie past the row, read the one before instead
          const float4 B = ld_stream((const float4*)((u == 0 && m0 == 0 ? pb2 - 2 : pb2) - (int64_t)u * 4 * MS));
          raw[2 * u].a = make_float2(A.x, A.y); raw[2 * u + 1].a = make_float2(A.z, A.w);
          raw[2 * u].b = make_float2(B.x, B.y); raw[2 * u + 1].b = make_float2(B.z, B.w);
        }
        const uint32_t s = tile * T3 + q;
        special = ld_stream(mb == 0 ? X0s + xi(s ? Rr - s : 0u, 0) : X0s + xi(Rr - 1 - s, M - mb));
        return;
      }
    }
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;
      const uint32_t slo = (eb & (T - 1)) >> 1, mb = eb >> logT;
      const uint32_t s = tile * T3 + slo;
      const cf* __restrict__ pa = X0s + xi(s, mb);
      const cf* __restrict__ pb = g.real_input ? X0s + xi(Rr - 1 - s, M - mb) : pa + (g.npol == 2 ? L : 0);
      const int64_t stepb = g.real_input ? -step : step;
      const cf* __restrict__ pb0 = (g.real_input && mb == 0) ? X0s + xi(s ? Rr - s : 0u, 0) : pb;
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        if (chunk >= 0 && i % NCHUNK != chunk) continue;
        Abk r;
        r.a = ld_stream(pa + i * step);
        r.b = ld_stream(i == 0 ? pb0 : pb + i * stepb);
        raw[(g2 / 2) * P::R1 + i] = r;
      }
    }
  };
  // chirp of a tile (fetched at the start of the item: keeps the prefetch at 64 registers)
  auto load_chirp = [&](const Item item, cf (&kk)[PTS / 2]) {
    const uint32_t ktile = item.tile;
    if (kernel && !(FB_DBG(g) & (2 | 32))) {      // uniform; outside the unrolled loads (no per-load branch / vmcnt(0))
      constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
#pragma unroll
      for (int g2 = 0; g2 < P::G1; g2 += 2) {
        const uint32_t eb = P::G1 * tid + g2;
        const cf* __restrict__ pk = kernel + ((uint64_t)(ktile * T3 + ((eb & (T - 1)) >> 1)) << LOGF) + (eb >> logT);
#pragma unroll
        for (int i = 0; i < P::R1; i++) kk[(g2 / 2) * P::R1 + i] = pk[i * MS];
      }
    } else {
#pragma unroll
      for (int i = 0; i < PTS / 2; i++) kk[i] = make_float2(1.f, 0.f);
    }
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;      // behind the exchange buffer (16-byte aligned)
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  fb_stagger();
  fb_setprio();
  // FOLD: two buffers of out.plan_cap plan entries behind the twiddle tables (cf index, 16-byte aligned), followed by
  // a copy of the launch's nparts + 1 part offsets into the plan (PSL_MAX words): a part's entries are then found
  // without a dependent pair of global loads, and are fetched one item ahead (registers) like the tile itself
  const uint32_t plan_off = (ltw_off + ltw_entries_dev<LOGF>() + 1) & ~1u;
  uint32_t* psl = nullptr;
  const uint4* __restrict__ fent_all = nullptr;
  bool use_psl = false;
  if constexpr (FOLD) {
    psl = (uint32_t*)&lds[plan_off + 4 * out.plan_cap];
    fent_all = (const uint4*)(out.pstart + ((out.nparts_plan + 1 + 3) & ~3u));
  }
  uint32_t jt = 0;                                              // tiles done by this workgroup
  Item item, next;
  uint32_t j = 0;
  // FOLD: workgroup b takes tiles b, b + grid, ... and walks the parts of each in order.
  // Segmented (out.nseg > 1, geometries with fewer channel tiles than compute units): the parts of the launch are cut
  // into nseg runs; workgroup (lane, seg) walks the parts of run `seg` for tiles lane, lane + ntg, ...  Run 0 adds onto
  // the profile (it continues the sums of earlier launches in time order), the others onto zeroed partial profiles that
  // are added to the profile, in run order, after the launch -- the sums of a launch are re-associated per run.
  uint32_t fold_b = blockIdx.x;
  const uint32_t fnseg = FOLD && out.nseg > 1 ? out.nseg : 1u;
  const uint32_t fntg = gridDim.x / fnseg;                         // workgroups per run (host: gridDim.x % nseg == 0)
  const uint32_t fseg = fnseg > 1 ? fold_b / fntg : 0u;
  const uint32_t fpps = (nparts + fnseg - 1) / fnseg;              // parts per run
  const uint32_t fp0 = fseg * fpps;
  const uint32_t fnp = fp0 >= nparts ? 0u : (nparts - fp0 < fpps ? nparts - fp0 : fpps);
  if (fnseg > 1) fold_b -= fseg * fntg;
  if constexpr (FOLD) {
    if (fnp == 0) return;
    // the offsets of the parts THIS workgroup walks (its run of a segmented launch: launches of up to 256 parts are cut into
    // runs of at most FB_PSL_MAX - 1; the whole launch's offsets did not fit, and every entry and accumulator of such launches
    // -- the sub-band and -F 256:D geometries -- then came from global memory in the fold phase), psl[lp - fp0]
    use_psl = out.plan_cap > 0 && fnp + 1 <= FB_PSL_MAX;
    if (use_psl) {
      for (uint32_t q = tid; q <= fnp; q += blockDim.x) psl[q] = out.pstart[part0 + fp0 + q];
      __syncthreads();
    }
    // tiles that share an X layout block (2^(logX3-logT3) of them) go to blocks b, b+8, ... : one XCD under the
    // observed round-robin placement, at the same time, so the block's lines are fetched once (speed only)
    const int lr = logX3 - logT3;
    if (fnseg == 1 && lr > 0 && (gridDim.x & ((8u << lr) - 1)) == 0)
      fold_b = ((((fold_b >> (3 + lr)) << 3) | (fold_b & 7)) << lr) | ((fold_b >> 3) & ((1u << lr) - 1));
  }
  // Without the fold the order of the items is free.  When every workgroup gets the same number of tiles it also walks the
  // parts of a tile one after the other, so that the tile's chirp stays in registers (one chirp read per launch, not per
  // part); otherwise the items are dealt XCD-wise as in the other passes.
  const bool tile_major = FOLD || (ntile >= gridDim.x && ntile % gridDim.x == 0 && !(FB_DBG(g) & 512));
  auto next_item = [&](const uint32_t jj, Item& it) -> bool {
    if (FOLD || tile_major) {
      const uint32_t q = jj / fnp;                     // (32-bit; jj counts this workgroup's items)
      it.tile = fold_b + q * fntg;
      it.lp = fp0 + (jj - q * fnp);
      return it.tile < ntile;
    } else {
      // XCD dealing as persistent_item (wgfft.h) with runs of `run` items; run == nparts (the default) makes the run
      // index the tile and the position in the run the part, without a division by nparts
      const uint32_t grid = gridDim.x, b = blockIdx.x;
      uint32_t hi, lo;
      if (grid & 7) {
        const uint32_t lin = b + jj * grid;
        hi = lin / run; lo = lin - hi * run;
      } else {
        const uint32_t q = jj * (grid >> 3) + (b >> 3);
        const uint32_t qr = q / run;
        hi = qr * 8 + (b & 7); lo = q - qr * run;
      }
      if (run == nparts) { it.tile = hi; it.lp = lo; }
      else { const uint32_t lin = hi * run + lo; it.tile = lin / nparts; it.lp = lin - it.tile * nparts; }
      return it.tile < ntile;                          // (lp < nparts by construction)
    }
  };
  if (!next_item(j, item)) return;
  Abk raw[PTS / 2];
  fetch(item, raw, -1);
  cf kk[PTS / 2];                       // chirp of the current tile
  uint32_t kk_tile = ~0u;
  // FOLD: plan entry of this thread for the item about to be processed (tid < number of active bins of the part)
  uint32_t fe0_cur = 0, fn_cur = 0;
  auto plan_fetch = [&](const Item it) {
    if constexpr (FOLD) {
      const uint32_t lp = it.lp;
      if (use_psl) { fe0_cur = psl[lp - fp0]; fn_cur = psl[lp - fp0 + 1] - fe0_cur; }
      else { fe0_cur = out.pstart[part0 + lp]; fn_cur = out.pstart[part0 + lp + 1] - fe0_cur; }
      if (FB_DBG(g) & 16) fn_cur = 0;
    }
  };
  // The active-bin entries of a part (16 bytes each, at most plan_cap <= blockDim of them) go from global memory STRAIGHT into
  // their LDS buffer (global_load_lds_dwordx4: lane l of a wave lands at the wave's base + 16*l), asynchronously and
  // without passing through registers.  Round 2 fetched them into a register at the top of the tile and stored them to
  // LDS: the kernel sits at 256 registers, the value was spilled to scratch, and the ISA read `s_waitcnt vmcnt(0);
  // global_load; s_waitcnt vmcnt(0); scratch_store; ...; scratch_load; s_waitcnt vmcnt(0); ds_write` -- two exposed memory
  // round trips on the two waves everyone then waits for at the first barrier.  The entries of the NEXT item are now
  // requested in the middle of the current tile's transform (behind a barrier that the previous readers of that buffer
  // have passed) and are waited for, together with the prefetched tile, at the top of the next one.
  // (needs a second stage: the request for the next item is issued from wgfft's `mid` hook behind the first exchange barrier;
  //  single-stage transforms read their entries from global memory)
  const bool plan_dma_ok = FOLD && FftPlan<LOGF>::NS >= 2 && use_psl;
  auto plan_dma = [&](const Item it, const uint32_t buf) {
    if constexpr (FOLD) {
      const uint32_t fe0 = psl[it.lp - fp0], fn = (FB_DBG(g) & 16) ? 0u : psl[it.lp - fp0 + 1] - fe0;
      if (fn <= out.plan_cap && tid < fn)
        lds_dma_b128((const void*)(fent_all + fe0 + tid), lds_byte_addr((const uint4*)&lds[plan_off] + buf * out.plan_cap + (tid & ~63u)));
    }
  };
  if (plan_dma_ok) plan_dma(item, 0);

#if defined(FB_STAMPS) && FB_STAMPS == 3
  unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, acc_s[6] = {0, 0, 0, 0, 0, 0};
  STAMP(ts5);
#endif
  for (;;) {
    asm volatile("" : "+v"(tid));   // per-tile index math stays inside the loop (see wgfft)
    cx2 x[NPAIR];
#if defined(FB_STAMPS) && FB_STAMPS == 3
    // wait for the prefetched tile explicitly so that the wait is timed separately
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(ts0);
#endif
    {
      // FOLD: a workgroup walks the parts of ITS tile, so consecutive items share the chirp rows: they are loaded when
      // the tile changes and stay in registers (the load and its latency were 22 % of the tile, profiles/r02c_*)
      if (item.tile != kk_tile) {
        load_chirp(item, kk);
        kk_tile = item.tile;
      }
      if constexpr (FOLD) {
        // this part's active-bin entries travel with the chirp loads and are parked in LDS (double buffered: slower
        // waves may still be folding the previous tile from the other half); their offsets come from the LDS copy
        plan_fetch(item);
        // the entries requested during the previous tile (or in front of the loop) have landed once every older load has
        // -- they were issued half a tile ago, behind the prefetch of this tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if constexpr (PAIR16) {
        if (pair16) {                       // hand the other channel's halves of the 16-byte pairs to the neighbour lane
          const bool q = tid & 1;
          auto swap1 = [](const cf v) {     // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
            return make_float2(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0xB1, 0xf, 0xf, false)),
                               __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0xB1, 0xf, 0xf, false)));
          };
#pragma unroll
          for (int u = 0; u < P::R1 / 2; u++) {
            const cf alo = raw[2 * u].a, ahi = raw[2 * u + 1].a;            // (channel 0, channel 1) of element 2u + q
            const cf aown = q ? ahi : alo, arecv = swap1(q ? alo : ahi);
            raw[2 * u].a = q ? arecv : aown;
            raw[2 * u + 1].a = q ? aown : arecv;
            const cf blo = raw[2 * u].b, bhi = raw[2 * u + 1].b;            // mirror rows: (channel 1, channel 0)
            const cf bown = q ? blo : bhi, brecv = swap1(q ? bhi : blo);
            raw[2 * u].b = q ? brecv : bown;
            raw[2 * u + 1].b = q ? bown : brecv;
          }
          if ((tid >> 1) == 0) raw[0].b = special;
        }
      }
      // (the uniform real/complex choice is made once, outside the unrolled loops: no branch per element)
      if (g.real_input) {
#pragma unroll
        for (int q = 0; q < PTS / 2; q++) {
          const Abk r = raw[q];
          // W[k] = X0[k] + i X1[k] ; conj(W[L-k]) = X0[k] - i X1[k]
          const cf x0 = make_float2(0.5f * (r.a.x + r.b.x), 0.5f * (r.a.y - r.b.y));
          const cf x1 = make_float2(0.5f * (r.a.y + r.b.y), 0.5f * (r.b.x - r.a.x));
          x[q] = cmuls(make_cx2(x0, x1), kk[q]);          // Response::operate, Response.C:429-441
        }
      } else {
        const bool two = g.npol == 2;
#pragma unroll
        for (int q = 0; q < PTS / 2; q++) {
          const Abk r = raw[q];
          x[q] = cmuls(make_cx2(r.a, two ? r.b : make_float2(0.f, 0.f)), kk[q]);
        }
      }
    }
#if defined(FB_STAMPS) && FB_STAMPS == 3
    STAMP(ts1);
#endif
    const bool more = next_item(++j, next);
    // One burst, and unconditional: the last item of a workgroup is fetched again and dropped.  Under `if (more)` the generic
    // (8-byte) form's loads went to fresh registers and the copies into the loop-carried `raw` sat at the end of the conditional
    // block behind `s_waitcnt vmcnt(16) ... (0)` -- the prefetch was waited for at once (profiles/r04_experiments.txt item 13;
    // what rounds 1-3 read as "the wave time goes to ISSUING the 8-byte loads").  The 16-byte pair form of the headline
    // geometry was not affected.
    fetch(more ? next : item, raw, -1);
#if defined(FB_STAMPS) && FB_STAMPS == 3
    STAMP(ts2);
#endif

    const uint32_t tile = item.tile;
    const uint64_t part = part0 + item.lp;
    // FOLD: the tile's detected samples are staged UNPADDED, channel after channel (16 bytes per sample), so that the
    // samples of a phase bin's run are read at constant offsets from one base (the padded image cost four integer
    // instructions per sample in a phase that only three of eight waves work in).  The channel stride is nkeep rounded up
    // so that the T3 channels a quarter wave writes at once fall on different LDS banks.
    // (many channels of a short transform: the rounded stride would not fit the exchange buffer -- 2*T3*nkeep words always do)
    const uint32_t fcr = (16u >> logT3) & 15u, fcs_r = ((g.nkeep + 15u - fcr) & ~15u) + fcr;
    const uint32_t fcs = ((2u * fcs_r) << logT3) <= PTS * blockDim.x ? fcs_r : g.nkeep;
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if constexpr (FOLD) {
        const uint32_t slo = col >> 1;
        const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t t = t0 + (int32_t)(k * pstride);
          if ((uint32_t)t >= g.nkeep) continue;           // outside the kept window (negative t wraps)
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          *(float4*)&lds[2 * (slo * fcs + (uint32_t)t)] = make_float4(r[0], r[1], r[2], r[3]);
        }
        return;
      }
      if (out.kind == 0) return;
      if (FB_DBG(g) & 1) { if (v[0].x[0] == 1.2345f && v[R - 1].y[1] == 3.3f) out.base[0] = v[0].x[0]; return; }
      const uint32_t chan = out.chan0 + tile * T3 + (col >> 1);
      float* __restrict__ row = out.base + chan * out.chan_stride;
      // output sample of element k: t0 + k*pstride (kept when 0 <= t < nkeep); the addresses are a base plus a
      // multiple of a wave-uniform step
      const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
      float2* __restrict__ o2 = (float2*)(row + part * out.part_step) + t0;
      float4* __restrict__ o4 = (float4*)row + ((int64_t)(part * g.nkeep) + t0);
      // the output kind / layout is uniform: chosen once, outside the unrolled element loop (no branch per element)
      if (out.kind == 1) {
        const bool two = g.npol == 2;
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float2* o = o2 + k * pstride;
          st_stream(o, cx2_lo(v[k]));
          if (two) st_stream((float2*)((float*)o + out.pol_stride), cx2_hi(v[k]));
        }
      } else if (out.ndim == 4) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          st_stream(o4 + k * pstride, make_float4(r[0], r[1], r[2], r[3]));
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t ts = t0 + (int32_t)(k * pstride);
          if ((uint32_t)ts >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          const uint64_t idat = part * g.nkeep + (uint32_t)ts;
          if (out.ndim == 2) {
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
    // FOLD: active phase bins of this part: entries {bin, first interval, count<<16 | hits0, offset0}
    // (fold_internal.h), copied to LDS at the start of the tile when they fit; one (entry, channel) accumulator per
    // work item.  The accumulator of a thread's first work item (nearly always its only one) is requested in the
    // middle of the transform -- behind two barriers, so the previous part's stores of this workgroup are visible --
    // and arrives while the last stage runs, instead of costing a memory round trip in the fold phase.
    uint32_t f_e0 = 0, f_nact = 0;
    const uint4* __restrict__ ent = nullptr;
    const uint4* planl = nullptr;
    bool in_lds = false;
    uint4 en_pre = make_uint4(0, 0, 0, 0);
    float4 acc_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr bool PRE = FOLD && FftPlan<LOGF>::NS >= 2;
    if constexpr (FOLD) {
      f_e0 = fe0_cur;
      f_nact = fn_cur;
      ent = fent_all + f_e0;
      planl = (const uint4*)&lds[plan_off] + (jt & 1) * out.plan_cap;
      in_lds = plan_dma_ok && f_nact <= out.plan_cap;
    }
    // accumulator of (work item w, phase bin b): one float4, or -- profile of npol 2 x ndim 2 -- a float2 in each of the
    // channel's two rows; the partial profiles of a segmented launch are packed in the same shape (rows of nbin bins)
    const bool planes2 = FOLD && out.prof_planes == 2;                       // uniform
    const uint64_t plane = fseg == 0 ? out.plane_stride : 2ull * out.nbin;   // floats between the two rows of a channel
    auto acc_row = [&](const uint32_t w) -> float* {
      const uint32_t cl = tile * T3 + (w & (T3 - 1));              // channel within this input channel's sub-band
      return (float*)(fseg == 0 ? (float4*)out.base + (uint64_t)(out.chan0 + cl) * out.prof_span4
                                : (float4*)out.part + ((uint64_t)(fseg - 1) * g.C + cl) * out.nbin);
    };
    auto acc_load = [&](float* row, const uint32_t b) -> float4 {
      if (planes2) {
        const float2 u = *(const float2*)(row + 2 * b), v = *(const float2*)(row + plane + 2 * b);
        return make_float4(u.x, u.y, v.x, v.y);
      }
      return *(const float4*)(row + 4 * b);
    };
    auto acc_store = [&](float* row, const uint32_t b, const float4 a) {
      if (planes2) {
        *(float2*)(row + 2 * b) = make_float2(a.x, a.y);
        *(float2*)(row + plane + 2 * b) = make_float2(a.z, a.w);
      } else {
        *(float4*)(row + 4 * b) = a;
      }
    };
    auto mid = [&](const int phase) {
      if constexpr (PRE) {
        // only when the part's plan entries are in LDS: the accumulator's address then depends on an LDS read alone.  With
        // the entry possibly coming from global memory (a select, or two branches that the compiler merges again) the
        // load below sat behind s_waitcnt vmcnt(0) -- a wait for the whole prefetch of the next tile, in the middle of the
        // transform, on exactly the three waves that also fold
        if (phase == 2 && in_lds && tid < (f_nact << logT3)) {
          en_pre = planl[tid >> logT3];
          acc_pre = acc_load(acc_row(tid), en_pre.x);
        }
        // every wave is past this tile's first exchange barrier, i.e. has left the previous tile's fold: the other plan
        // buffer is free for the entries of the next item
        if (phase == 2 && plan_dma_ok && more) plan_dma(next, (jt + 1) & 1);
      }
    };
    if (FB_DBG(g) & 4) wgfft_passthrough<LOGF>(tid, logT, x, store);
    else wgfft<LOGF, +1, FOLD>(lds, ltw_off, tid, logT, x, store, mid);
#if defined(FB_STAMPS) && FB_STAMPS == 3
    STAMP(ts3);
#endif
    if constexpr (FOLD) {
      __syncthreads();                       // the tile's detected samples are staged
      // the samples of an interval are fetched from LDS eight at a time (independent loads) and then added one after
      // the other, so the sum keeps the time order
      const bool pre = PRE && in_lds && !(FB_DBG(g) & 4);
      for (uint32_t w = tid; w < (f_nact << logT3); w += blockDim.x) {
        const uint32_t slo = w & (T3 - 1);
        uint4 en;
        float4 acc;
        if (pre && w == tid) {
          en = en_pre;
          acc = acc_pre;
        } else {
          en = in_lds ? planl[w >> logT3] : ent[w >> logT3];
          acc = acc_load(acc_row(w), en.x);
        }
        const uint32_t nint = en.z >> 16;
        float* __restrict__ pp = acc_row(w);
        uint32_t off = en.w, hits = en.z & 0xffffu;
        for (uint32_t i = 0;;) {
          const float4* __restrict__ src = (const float4*)&lds[2 * (slo * fcs + off)];     // consecutive samples: constant offsets
          uint32_t h = 0;
          for (; h + 8 <= hits; h += 8) {
            float4 sm[8];
#pragma unroll
            for (int q = 0; q < 8; q++) sm[q] = src[h + q];
#pragma unroll
            for (int q = 0; q < 8; q++) { acc.x += sm[q].x; acc.y += sm[q].y; acc.z += sm[q].z; acc.w += sm[q].w; }
          }
          for (; h < hits; h++) {
            const float4 sm = src[h];
            acc.x += sm.x; acc.y += sm.y; acc.z += sm.z; acc.w += sm.w;
          }
          if (++i >= nint) break;
          const Interval iv = out.piv[en.y + i];           // further intervals of the bin in this part (rare)
          off = (uint32_t)iv.offset; hits = iv.hits;
        }
        acc_store(pp, en.x, acc);
      }
      // the barrier in front of the next tile's first exchange write also ends this read phase
    }
#if defined(FB_STAMPS) && FB_STAMPS == 3
    STAMP(ts4);
    acc_s[0] += ts0 - ts5; acc_s[1] += ts1 - ts0; acc_s[2] += ts2 - ts1; acc_s[3] += ts3 - ts2; acc_s[4] += ts4 - ts3; acc_s[5] += 1;
    ts5 = ts4;
#endif
    if (!more) break;
    item = next;
    jt++;
  }
#if defined(FB_STAMPS) && FB_STAMPS == 3
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 6; q++) atomicAdd(&g_stamps[blockIdx.x][q], acc_s[q]);
#endif
}

#endif  // FB_HAS(3) || FB_HAS(5)


#if FB_HAS(6)
// ------------------------------------------------------------------------------------ two-pass path (short responses)
// A part needs log2 L forward and log2 M inverse radix-2 levels; a workgroup tile holds 14.  When log2 L + log2 M <= 27
// (complex dual-pol input; the 50 MHz sub-band geometry -F 512:D -x 512 is 18 + 9) TWO tiles cover them, and the spectrum
// never makes its round trip through HBM:  L = Fa * Fb with Fb = 2^13 / M and Fa = L / Fb <= 2^14 (at 2^14 one whole column per
// tile), sample n = nb + Fb*na, bin k = ka + Fa*kb -- and with ka = a*M + j that bin is bin j of channel c = a + (Fa/M)*kb.
// Fa < 2^14 (fewer channels): pass 1 is the ordinary k_raw_transpose + k_fwd_cols on a geometry of its own (M = Fa, Rr = Fb,
// T2 = freq_res: the A layout below is exactly theirs); only Fa = 2^14 needs P0' / P1'.
//   P0' k_raw_cols   the 8-bit block regrouped per column and polarisation: Rt[part][pol][nb][na]
//   P1' k_fwd_col1   ONE Fa = 2^14-point FFT per tile (column nb of one polarisation): the even and the odd samples are the
//                    two interleaved columns of a 2^13-point wgfft, combined by one radix-2 step in registers;
//                    out A[pol][a][nb][j] (each (a, nb) run M contiguous elements)
//   P2' k_rows_inv   tile = (a, part), both polarisations: x W_L^{nb*ka}, Fb-point FFTs over nb in registers (-> kb, i.e. Fb
//                    whole channels), x chirp, through the exchange buffer, inverse M-point FFTs over j, keep window,
//                    detection, fold -- k_fwd_rows and k_inv_chan in one tile.
// Traffic per part at the sub-band geometry: 1 + 1 (regroup) + 1 + 4 (pass 1) + 4 + chirp (pass 2) MB instead of
// 1 + 1 + 1 + 4 + 4 + 4 + 4 + chirp, and one kernel's load / store phases less.

// P0': sample t = nb + Fb*na of a part, 4 bytes (p0 re, p0 im, p1 re, p1 im) -> Rt[part][pol][nb][na] byte pairs (re, im).
// A block regroups NS consecutive samples (NS/Fb rows na of all Fb columns) through LDS: 16-byte loads, 16-byte stores in
// runs of 2*NS/Fb bytes.
__global__ __launch_bounds__(256) void k_raw_cols(const FbGeom g, const FbIn in, uint16_t* __restrict__ Rt, const uint64_t part0)
{
  constexpr uint32_t NS = 8192, MAXFB = 64;
  __shared__ __attribute__((aligned(16))) uint16_t sm[2 * (NS + 8 * MAXFB)];
  const uint32_t tid = threadIdx.x;
  const int logFb = g.logFb2;
  const uint32_t Fb = 1u << logFb, R = NS >> logFb, RP = R + 8;      // rows of the block, row pitch (16-byte aligned, bank skew)
  const uint64_t part = blockIdx.y;
  const uint32_t na0 = blockIdx.x * R;
  const uint64_t t0 = (part0 + part) * in.part_step + ((uint64_t)na0 << logFb);
  if (in.nchan == 1) {
    const uint4* __restrict__ src = (const uint4*)((const uint8_t*)in.base + 4 * t0);
#pragma unroll 4
    for (uint32_t q = 0; q < NS / 4 / 256; q++) {
      const uint32_t v = tid + 256 * q;
      const uint4 w = src[v];
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t sidx = 4 * v + k, nb = sidx & (Fb - 1), r = sidx >> logFb;
        sm[nb * RP + r] = (uint16_t)(ww[k] & 0xffffu);
        sm[(Fb + nb) * RP + r] = (uint16_t)(ww[k] >> 16);
      }
    }
  } else {
    // several input channels in the block (byte ((t*nchan + c)*npol + p)*2 + d): this channel's word of every sample
    const uint32_t* __restrict__ src = (const uint32_t*)in.base;
    for (uint32_t sidx = tid; sidx < NS; sidx += 256) {
      const uint32_t w = src[(t0 + sidx) * in.nchan + in.ichan], nb = sidx & (Fb - 1), r = sidx >> logFb;
      sm[nb * RP + r] = (uint16_t)(w & 0xffffu);
      sm[(Fb + nb) * RP + r] = (uint16_t)(w >> 16);
    }
  }
  __syncthreads();
  const uint32_t r8n = R >> 3;                                         // 16-byte units per (pol, nb) row of the block
  for (uint32_t u = tid; u < 2 * Fb * r8n; u += 256) {
    const uint32_t r8 = u % r8n, row = u / r8n, seq = row >> logFb, nb = row & (Fb - 1);
    const uint4 val = *(const uint4*)&sm[row * RP + 8 * r8];
    uint16_t* __restrict__ dst = Rt + (((part * 2 + seq) << logFb) + nb) * 16384ull + na0 + 8 * r8;
    *(uint4*)dst = val;
  }
}

// P1' last-stage functor: the pair (col 0, col 1) of position k holds E[k], O[k], the 2^13-point transforms of the even and
// the odd samples; Y[k] = E[k] + W^k O[k], Y[k + 2^13] = E[k] - W^k O[k], W = exp(-2 pi i / 2^14).  The radix-2 last stage hands
// over k = p and p + 2^12 (W^(p + 2^12) = -i W^p); a thread's consecutive calls are the adjacent rows p, p + 1: the even
// call parks its four outputs, the odd one writes four 16-byte pairs into the staged image (natural order of ka).
struct Col1Out {
  cf* img;
  int h;
  cf ya[4];
  template <int R> DEV void operator()(const uint32_t, const uint32_t p, const uint32_t pstride, cx2 (&v)[R])
  {
    static_assert(R == 2, "k_fwd_col1: radix-2 last stage");
    const float ang = (float)p * (1.0f / 16384.0f);                    // revolutions, exact
    const cf w = make_float2(__builtin_amdgcn_cosf(ang), -__builtin_amdgcn_sinf(ang));
    const cf w1 = make_float2(w.y, -w.x);                              // -i * w
    const cf t0 = cmul(w, cx2_hi(v[0])), t1 = cmul(w1, cx2_hi(v[1]));
    const cf e0 = cx2_lo(v[0]), e1 = cx2_lo(v[1]);
    const cf y[4] = {make_float2(e0.x + t0.x, e0.y + t0.y), make_float2(e1.x + t1.x, e1.y + t1.y),
                     make_float2(e0.x - t0.x, e0.y - t0.y), make_float2(e1.x - t1.x, e1.y - t1.y)};   // rows p + m * pstride
    if ((h & 1) == 0) {
#pragma unroll
      for (int m = 0; m < 4; m++) ya[m] = y[m];
    } else {
      const uint32_t b = lds_pad(p - 1);                               // p - 1 even; m * pstride is a multiple of 64
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const uint32_t c = m * pstride;
        *(float4*)&img[b + c + ((c >> 6) << 2)] = make_float4(ya[m].x, ya[m].y, y[m].x, y[m].y);
      }
    }
  }
};

template <int RAWW>
__global__ __launch_bounds__(512) void k_fwd_col1(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                  const cf* __restrict__ tw, const uint32_t nparts,
                                                  const uint32_t nseq, const uint32_t run)
{
  constexpr int LOGF = 13, LOGT = 1;
  typedef FftPlan<LOGF> P;
  static_assert(P::REM == 1 && P::R1 == 16 && P::G1 == 2, "k_fwd_col1: 16 x 16 x 16 x 2");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logFb = g.logFb2, logMi = g.logMf;
  const uint32_t Mi = 1u << logMi;
  const uint32_t total = (nseq * nparts) << logFb;                     // item = (part*nseq + seq)*Fb + nb: tiles are contiguous in Rt
  auto fetch = [&](const uint32_t item, RawW<RAWW> (&raw)[PTS / 2]) {
    const uint64_t t0 = (uint64_t)item << 14;
#pragma unroll
    for (int i = 0; i < P::R1; i++) raw[i] = fetch_pair<RAWW>(g, in, 0, t0 + 2 * tid + 1024u * i);   // samples 2*pos, 2*pos + 1
  };
  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  const uint32_t co_lds = lds_pad(2 * threadIdx.x);
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  RawW<RAWW> raw[PTS / 2];
  fetch(item, raw);
#if defined(FB_STAMPS) && FB_STAMPS == 6
  unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, acc_s[6] = {0, 0, 0, 0, 0, 0};
  STAMP(ts5);
#endif
  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
#if defined(FB_STAMPS) && FB_STAMPS == 6
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(ts0);
#endif
#pragma unroll
    for (int h = 0; h < NPAIR; h++) {
      cf a, b;
      decode_pair<RAWW>(g, in, raw[h], a, b, 0);
      x[h] = make_cx2(a, b);                                           // (even sample, odd sample) of position tid + 512*h
    }
#if defined(FB_STAMPS) && FB_STAMPS == 6
    STAMP(ts1);
#endif
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    if (more) fetch(next, raw);
#if defined(FB_STAMPS) && FB_STAMPS == 6
    STAMP(ts2);
#endif
    Col1Out out;
    out.img = lds;
    out.h = 0;
    wgfft<LOGF, -1, true>(lds, ltw_off, tid, LOGT, x, out);
    __syncthreads();
#if defined(FB_STAMPS) && FB_STAMPS == 6
    STAMP(ts3);
#endif
    // copy-out: the staged column (natural order of ka) as 16-byte pairs, runs of M elements: A[seq][ka / M][nb][ka % M]
    const uint32_t nb = item & ((1u << logFb) - 1);
    cf* __restrict__ Aseq = A + ((uint64_t)(item >> logFb) << (14 + logFb));
#pragma unroll
    for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
      float4 pr[4];
#pragma unroll
      for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[co_lds + (j4 + q) * (1024u + 64u)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t ka = 2 * tid + 1024u * (j4 + q);
        st_stream((float4*)&Aseq[((((ka >> logMi) << logFb) + nb) << logMi) + (ka & (Mi - 1))], pr[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#if defined(FB_STAMPS) && FB_STAMPS == 6
    STAMP(ts4);
    acc_s[0] += ts0 - ts5; acc_s[1] += ts1 - ts0; acc_s[2] += ts2 - ts1; acc_s[3] += ts3 - ts2; acc_s[4] += ts4 - ts3; acc_s[5] += 1;
    ts5 = ts4;
#endif
    if (!more) break;
    item = next;
  }
#if defined(FB_STAMPS) && FB_STAMPS == 6
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 6; q++) atomicAdd(&g_stamps[blockIdx.x][q], acc_s[q]);
#endif
}

// P1', three exchanged stages (round 4, second form).  2^14 = 16^3 * 4: the column as FOUR interleaved sub-sequences
// y_c[m] = y[4m + c] -- the four columns of a 2^12-point wgfft (three radix-16 stages, none of them a remainder stage) -- and
// the last radix-4 level in registers:  Y[P + 2^12 q] = sum_c (-i)^(c q) W^(c P) F_c[P],  W = exp(-2 pi i / 2^14).
// The last stage leaves a thread the pair (F_c, F_c+1)[P_k], P_k = p + 256 k, with c = 0 in even lanes and c = 2 in odd lanes of
// the same p: each lane twiddles its own pair (apply_pass_twiddle: W^(c P) for the columns c, c + 1), the two lanes swap pairs
// (DPP) and each computes two of the four outputs -- even lanes rows P and P + 2^13, odd lanes P + 2^12 and P + 3*2^12.
// Against k_fwd_col1 (even / odd halves + radix-2 step) one whole exchanged stage -- 16 b128 reads and 16 b128 writes per
// thread, two barriers -- is replaced by about 350 vector instructions.
struct Col1qOut {
  cf* img;
  const cf* tw;
  const cf* tw_lo;
  int h;
  template <int R> DEV void operator()(const uint32_t col, const uint32_t p, const uint32_t pstride, cx2 (&v)[R])
  {
    static_assert(R == 16, "k_fwd_col1q: radix-16 last stage");
    apply_pass_twiddle<R>(v, col, p, pstride, 14, tw, tw_lo);            // v[k] = (G_c, G_c+1)[p + k*pstride], G_c = W^(c P) F_c
    const bool odd = col != 0;                                            // col = 0 (c = 0, 1) or 2 (c = 2, 3)
    const float sg = odd ? -1.0f : 1.0f;
    auto swp = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, false)); };
    // staged image: natural order of ka, 8-byte elements, rows of odd q moved by 8 elements (XOR of bit 3) so that the even
    // and the odd lanes of a store fall on different banks (their rows differ by a multiple of 2^12 elements)
    const uint32_t q1 = odd ? 1u : 0u;
    const uint32_t b1 = lds_pad((p ^ (q1 << 3)) + (q1 << 12)), b2 = lds_pad((p ^ (q1 << 3)) + ((q1 + 2) << 12));
#pragma unroll
    for (int k = 0; k < R; k++) {
      const cx2 own = v[k];
      cx2 rc;
      rc.x = (v2f){swp(own.x[0]), swp(own.x[1])};
      rc.y = (v2f){swp(own.y[0]), swp(own.y[1])};
      // even: (G0 + G2, G1 + G3) = (A, C) ; odd: (G0 - G2, G1 - G3) = (B, D)
      cx2 sm;
      sm.x = own.x * sg + rc.x;
      sm.y = own.y * sg + rc.y;
      const cf sa = cx2_lo(sm), sb = cx2_hi(sm);
      const cf r = odd ? make_float2(sb.y, -sb.x) : sb;                   // odd: -i D
      const uint32_t c = k * pstride;                                     // multiple of 64: the padding carries over
      const uint32_t o = c + ((c >> 6) << 2);
      img[b1 + o] = make_float2(sa.x + r.x, sa.y + r.y);                  // even: Y[P] = A + C          odd: Y[P + 2^12] = B - i D
      img[b2 + o] = make_float2(sa.x - r.x, sa.y - r.y);                  // even: Y[P + 2^13] = A - C   odd: Y[P + 3*2^12] = B + i D
    }
  }
};

template <int RAWW>
__global__ __launch_bounds__(512) void k_fwd_col1q(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                   const cf* __restrict__ tw, const uint32_t nparts,
                                                   const uint32_t nseq, const uint32_t run)
{
  constexpr int LOGF = 12, LOGT = 2;
  typedef FftPlan<LOGF> P;
  static_assert(P::REM == 0 && P::R1 == 16 && P::G1 == 2, "k_fwd_col1q: 16 x 16 x 16");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logFb = g.logFb2, logMi = g.logMf;
  const uint32_t Mi = 1u << logMi;
  const uint32_t total = (nseq * nparts) << logFb;
  auto fetch = [&](const uint32_t item, RawW<RAWW> (&raw)[PTS / 2]) {
    const uint64_t t0 = (uint64_t)item << 14;
#pragma unroll
    for (int i = 0; i < P::R1; i++) raw[i] = fetch_pair<RAWW>(g, in, 0, t0 + 2 * tid + 1024u * i);   // samples 4*pos + c, c = 2*(tid & 1) + {0, 1}
  };
  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  const uint32_t co_lds0 = lds_pad(2 * threadIdx.x), co_lds1 = lds_pad((2 * threadIdx.x) ^ 8u);
  uint32_t item, next;
  uint32_t j = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, j, run, total, item)) return;
  RawW<RAWW> raw[PTS / 2];
  fetch(item, raw);
  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
#pragma unroll
    for (int h = 0; h < NPAIR; h++) {
      cf a, b;
      decode_pair<RAWW>(g, in, raw[h], a, b, 0);
      x[h] = make_cx2(a, b);
    }
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++j, run, total, next);
    if (more) fetch(next, raw);
    Col1qOut out;
    out.img = lds;
    out.tw = tw;
    out.tw_lo = g.tw_lo;
    out.h = 0;
    wgfft<LOGF, -1, true>(lds, ltw_off, tid, LOGT, x, out);
    __syncthreads();
    const uint32_t nb = item & ((1u << logFb) - 1);
    cf* __restrict__ Aseq = A + ((uint64_t)(item >> logFb) << (14 + logFb));
#pragma unroll
    for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
      float4 pr[4];
#pragma unroll
      for (int q = 0; q < 4; q++)                                        // row block jj holds q = jj / 4: rows of odd q are XOR-8 swizzled
        pr[q] = *(const float4*)&lds[((((j4 + q) >> 2) & 1) ? co_lds1 : co_lds0) + (j4 + q) * (1024u + 64u)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t ka = 2 * tid + 1024u * (j4 + q);
        st_stream((float4*)&Aseq[((((ka >> logMi) << logFb) + nb) << logMi) + (ka & (Mi - 1))], pr[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!more) break;
    item = next;
  }
}

// P2': see the head of this section.  LOGM + LOGFB == 13: a tile is Fb channels x 2 polarisations x M bins = 2^14 points;
// a thread holds, for NJ = 16 / Fb bins j = tid + 512*jq, the Fb rows nb of both polarisations (pair = (pol 0, pol 1)).
// Items, the fused fold (exact time order per tile, or segmented over part runs) and the output forms are k_inv_chan's.
template <int LOGM, int LOGFB, bool FOLD>
__global__ __launch_bounds__(512) void k_rows_inv(const FbGeom g, const cf* __restrict__ A,
                                                  const cf* __restrict__ kernel, const FbOut out,
                                                  const cf* __restrict__ tw, const uint64_t part0,
                                                  const uint32_t nparts, const uint32_t run)
{
  static_assert(LOGM + LOGFB == 13 && LOGFB >= 1 && LOGFB <= 4, "k_rows_inv: Fb channels x 2 pols x M bins = 2^14 points");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  // HALF (M = 512 = 16 * 16 * 2): the inverse transform as the 256-point transforms of the even and of the odd bins -- two
  // more "columns" per (channel, pol), two radix-16 stages, no remainder stage -- and the last radix-2 level in registers:
  // y[m] = E[m] + W^-m O[m], y[m + 256] = E[m] - W^-m O[m].  The last stage leaves E in even lanes and O in odd lanes of the
  // same (channel, m): the odd lane's twiddle, one DPP swap and one packed fma per element replace a whole exchanged stage.
  // Measured (profiles/r04_experiments.txt item 10): fused 430.7 -> 417.3 us per 256 parts (-3 %), but the form that writes
  // its output 497.6 -> 635.8 (+28 %): with 64 columns a wave of the last stage spans all channels and both halves of the
  // transform, two consecutive samples per (channel, half) -- 32-byte store runs.  Off; fused and unfused keep one association.
  constexpr bool HALF = false && (LOGM % 4) == 1;
  constexpr int LOGFI = HALF ? LOGM - 1 : LOGM;                      // length of the exchanged inverse transform
  constexpr int logT3 = LOGFB, logT = LOGFB + 1 + (HALF ? 1 : 0);    // columns: (kb, [bin parity,] pol)
  constexpr uint32_t Fb = 1u << LOGFB, T3 = Fb, NJ = 16 / Fb;
  const int logCa = g.logFa2 - LOGM;                                // Fa / M: channel stride between the rows kb of a tile
  const int logL = g.logFa2 + LOGFB;
  const uint64_t L = 1ull << logL;
  const uint32_t ntile = 1u << logCa;
  struct Abk { cf a, b; };
  struct Item { uint32_t tile, lp; };
  auto chan_of = [&](const uint32_t tile, const uint32_t kb) { return tile + (kb << logCa); };
  // Loads: the lane pair (2q, 2q + 1) needs bins j = 2q, 2q + 1 of both polarisations.  The even lane loads the two bins of
  // polarisation 0, the odd lane those of polarisation 1 -- one aligned 16-byte load each instead of two 8-byte ones (half the
  // load instructions: 8-byte-per-lane streams run at 5.6 TB/s, 16-byte ones at 7.1, tools/load_width_probe.hip; the wave
  // time of the prefetch is the ISSUE of its loads) -- and the halves are swapped between the two lanes when the tile is consumed.
  auto fetch = [&](const Item item, Abk (&raw)[PTS / 2]) {
    const cf* __restrict__ A0 = A + (uint64_t)item.lp * 2 * L + ((tid & 1u) ? L : 0) + (((uint64_t)item.tile << LOGFB) << LOGM) + (tid & ~1u);
#pragma unroll
    for (uint32_t jq = 0; jq < NJ; jq++)
#pragma unroll
      for (uint32_t nb = 0; nb < Fb; nb++) {
        const float4 v = ld_stream((const float4*)(A0 + (nb << LOGM) + 512u * jq));
        Abk r;
        r.a = make_float2(v.x, v.y);             // even lane: pol 0 of bin j     | odd lane: pol 1 of bin j - 1
        r.b = make_float2(v.z, v.w);             //            pol 0 of bin j + 1 |           pol 1 of bin j
        raw[jq * Fb + nb] = r;
      }
  };
  auto swap1 = [](const cf v) {                  // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
    return make_float2(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0xB1, 0xf, 0xf, false)),
                       __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0xB1, 0xf, 0xf, false)));
  };
  auto load_chirp = [&](const Item item, cf (&kk)[PTS / 2]) {
    if (kernel) {
#pragma unroll
      for (uint32_t jq = 0; jq < NJ; jq++)
#pragma unroll
        for (uint32_t kb = 0; kb < Fb; kb++) kk[jq * Fb + kb] = kernel[((uint64_t)chan_of(item.tile, kb) << LOGM) + tid + 512u * jq];
    } else {
#pragma unroll
      for (int i = 0; i < PTS / 2; i++) kk[i] = make_float2(1.f, 0.f);
    }
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGFI>(lds, ltw_off, tw, tid, blockDim.x);
  const uint32_t plan_off = (ltw_off + ltw_entries_dev<LOGM>() + 1) & ~1u;     // (the host sizes the tables for LOGM)
  uint32_t* psl = nullptr;
  const uint4* __restrict__ fent_all = nullptr;
  bool use_psl = false;
  if constexpr (FOLD) {
    psl = (uint32_t*)&lds[plan_off + 4 * out.plan_cap];
    fent_all = (const uint4*)(out.pstart + ((out.nparts_plan + 1 + 3) & ~3u));
  }
  uint32_t jt = 0;
  Item item, next;
  uint32_t j = 0;
  uint32_t fold_b = blockIdx.x;
  const uint32_t fnseg = FOLD && out.nseg > 1 ? out.nseg : 1u;
  const uint32_t fntg = gridDim.x / fnseg;
  const uint32_t fseg = fnseg > 1 ? fold_b / fntg : 0u;
  const uint32_t fpps = (nparts + fnseg - 1) / fnseg;
  const uint32_t fp0 = fseg * fpps;
  const uint32_t fnp = fp0 >= nparts ? 0u : (nparts - fp0 < fpps ? nparts - fp0 : fpps);
  if (fnseg > 1) fold_b -= fseg * fntg;
  if constexpr (FOLD) {
    if (fnp == 0) return;
    use_psl = out.plan_cap > 0 && fnp + 1 <= FB_PSL_MAX;          // offsets of this workgroup's run of parts: psl[lp - fp0]
    if (use_psl) {
      for (uint32_t q = tid; q <= fnp; q += blockDim.x) psl[q] = out.pstart[part0 + fp0 + q];
      __syncthreads();
    }
  }
  const bool tile_major = FOLD || (ntile >= gridDim.x && ntile % gridDim.x == 0);
  auto next_item = [&](const uint32_t jj, Item& it) -> bool {
    if (FOLD || tile_major) {
      const uint32_t q = jj / fnp;
      it.tile = fold_b + q * fntg;
      it.lp = fp0 + (jj - q * fnp);
      return it.tile < ntile;
    } else {
      const uint32_t grid = gridDim.x, b = blockIdx.x;
      uint32_t hi, lo;
      if (grid & 7) {
        const uint32_t lin = b + jj * grid;
        hi = lin / run; lo = lin - hi * run;
      } else {
        const uint32_t q = jj * (grid >> 3) + (b >> 3);
        const uint32_t qr = q / run;
        hi = qr * 8 + (b & 7); lo = q - qr * run;
      }
      if (run == nparts) { it.tile = hi; it.lp = lo; }
      else { const uint32_t lin = hi * run + lo; it.tile = lin / nparts; it.lp = lin - it.tile * nparts; }
      return it.tile < ntile;
    }
  };
  if (!next_item(j, item)) return;
  Abk raw[PTS / 2];
  fetch(item, raw);
  cf kk[PTS / 2];
  uint32_t kk_tile = ~0u;
  uint32_t fe0_cur = 0, fn_cur = 0;
  auto plan_fetch = [&](const Item it) {
    if constexpr (FOLD) {
      const uint32_t lp = it.lp;
      if (use_psl) { fe0_cur = psl[lp - fp0]; fn_cur = psl[lp - fp0 + 1] - fe0_cur; }
      else { fe0_cur = out.pstart[part0 + lp]; fn_cur = out.pstart[part0 + lp + 1] - fe0_cur; }
    }
  };
  const bool plan_dma_ok = FOLD && FftPlan<LOGFI>::NS >= 2 && use_psl;
  auto plan_dma = [&](const Item it, const uint32_t buf) {
    if constexpr (FOLD) {
      const uint32_t fe0 = psl[it.lp - fp0], fn = psl[it.lp - fp0 + 1] - fe0;
      if (fn <= out.plan_cap && tid < fn)
        lds_dma_b128((const void*)(fent_all + fe0 + tid), lds_byte_addr((const uint4*)&lds[plan_off] + buf * out.plan_cap + (tid & ~63u)));
    }
  };
  if (plan_dma_ok) plan_dma(item, 0);
#if defined(FB_STAMPS) && FB_STAMPS == 7
  unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, ts6, acc_s[7] = {0, 0, 0, 0, 0, 0, 0};
  STAMP(ts5);
#endif

  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
#if defined(FB_STAMPS) && FB_STAMPS == 7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(ts0);
#endif
    {
      if (item.tile != kk_tile) {
        load_chirp(item, kk);
        kk_tile = item.tile;
      }
      if constexpr (FOLD) {
        plan_fetch(item);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // forward transform over the rows nb of this tile, per bin j: x W_L^{nb*ka} (the twiddle between the two forward passes,
      // ka = a*M + j), radix-Fb butterfly -> kb in natural order = the tile's Fb channels; x chirp (Response::operate)
#pragma unroll
      for (uint32_t jq = 0; jq < NJ; jq++) {
        cx2 v[Fb];
#pragma unroll
        for (uint32_t nb = 0; nb < Fb; nb++) {
          const Abk r = raw[jq * Fb + nb];
          const bool odd = tid & 1u;
          const cf recv = swap1(odd ? r.a : r.b);       // the even lane hands over bin j + 1 of pol 0, the odd one bin j - 1 of pol 1
          v[nb] = make_cx2(odd ? recv : r.a, odd ? r.b : recv);
        }
        const uint32_t ka = (item.tile << LOGM) + tid + 512u * jq;
        uint32_t jw[4];
        cf t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) jw[q] = (ka << q) & (uint32_t)(L - 1);
        twiddles_big(t, jw, logL, tw, g.tw_lo);
        apply_powers<Fb>(v, t[0], t[1], t[2], t[3]);
        fftR<Fb, -1>(v);
#pragma unroll
        for (uint32_t kb = 0; kb < Fb; kb++) x[jq * Fb + kb] = cmuls(v[kb], kk[jq * Fb + kb]);
      }
    }
#if defined(FB_STAMPS) && FB_STAMPS == 7
    STAMP(ts1);
#endif
    const bool more = next_item(++j, next);
    if (more) fetch(next, raw);
#if defined(FB_STAMPS) && FB_STAMPS == 7
    STAMP(ts2);
#endif
    // rows -> bins: element (bin j, column 2*kb + pol) of the inverse transform's tile, as the stages exchange them
    __syncthreads();                     // every wave has finished with the previous tile's image (last stage / fold phase)
#pragma unroll
    for (uint32_t jq = 0; jq < NJ; jq++) {
      const uint32_t jb = tid + 512u * jq;
      // HALF: position j / 2, column 4*kb + 2*(j & 1) + pol
      const uint32_t e0 = HALF ? (((jb >> 1) << logT) + 2 * (jb & 1u)) : (jb << logT);
#pragma unroll
      for (uint32_t kb = 0; kb < Fb; kb++) {
        const cx2 q = x[jq * Fb + kb];
        *(float4*)&lds[lds_pad(e0 + (HALF ? 4 : 2) * kb)] = make_float4(q.x[0], q.x[1], q.y[0], q.y[1]);
      }
    }
    __syncthreads();
#if defined(FB_STAMPS) && FB_STAMPS == 7
    STAMP(ts6);
#endif

    const uint32_t tile = item.tile;
    const uint64_t part = part0 + item.lp;
    const uint32_t fcr = (16u >> logT3) & 15u, fcs_r = ((g.nkeep + 15u - fcr) & ~15u) + fcr;
    const uint32_t fcs = ((2u * fcs_r) << logT3) <= PTS * blockDim.x ? fcs_r : g.nkeep;
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if constexpr (FOLD) {
        const uint32_t slo = col >> 1;
        const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t t = t0 + (int32_t)(k * pstride);
          if ((uint32_t)t >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          *(float4*)&lds[2 * (slo * fcs + (uint32_t)t)] = make_float4(r[0], r[1], r[2], r[3]);
        }
        return;
      }
      if (out.kind == 0) return;
      const uint32_t chan = out.chan0 + chan_of(tile, col >> 1);
      float* __restrict__ row = out.base + chan * out.chan_stride;
      const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
      float2* __restrict__ o2 = (float2*)(row + part * out.part_step) + t0;
      float4* __restrict__ o4 = (float4*)row + ((int64_t)(part * g.nkeep) + t0);
      if (out.kind == 1) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float2* o = o2 + k * pstride;
          st_stream(o, cx2_lo(v[k]));
          st_stream((float2*)((float*)o + out.pol_stride), cx2_hi(v[k]));
        }
      } else if (out.ndim == 4) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          st_stream(o4 + k * pstride, make_float4(r[0], r[1], r[2], r[3]));
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t ts = t0 + (int32_t)(k * pstride);
          if ((uint32_t)ts >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          const uint64_t idat = part * g.nkeep + (uint32_t)ts;
          if (out.ndim == 2) {
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
    uint32_t f_e0 = 0, f_nact = 0;
    const uint4* __restrict__ ent = nullptr;
    const uint4* planl = nullptr;
    bool in_lds = false;
    uint4 en_pre = make_uint4(0, 0, 0, 0), en_pre2 = make_uint4(0, 0, 0, 0);
    float4 acc_pre = make_float4(0.f, 0.f, 0.f, 0.f), acc_pre2 = acc_pre;
    constexpr bool PRE = FOLD && FftPlan<LOGFI>::NS >= 2;
    constexpr bool PRE2 = PRE && LOGFB >= 3;              // >= 8 channels per tile: a thread may fold a second item
    if constexpr (FOLD) {
      f_e0 = fe0_cur;
      f_nact = fn_cur;
      ent = fent_all + f_e0;
      planl = (const uint4*)&lds[plan_off] + (jt & 1) * out.plan_cap;
      in_lds = plan_dma_ok && f_nact <= out.plan_cap;
    }
    const bool planes2 = FOLD && out.prof_planes == 2;
    const uint64_t plane = fseg == 0 ? out.plane_stride : 2ull * out.nbin;
    auto acc_row = [&](const uint32_t w) -> float* {
      const uint32_t cl = chan_of(tile, w & (T3 - 1));
      return (float*)(fseg == 0 ? (float4*)out.base + (uint64_t)(out.chan0 + cl) * out.prof_span4
                                : (float4*)out.part + ((uint64_t)(fseg - 1) * g.C + cl) * out.nbin);
    };
    auto acc_load = [&](float* row, const uint32_t b) -> float4 {
      if (planes2) {
        const float2 u = *(const float2*)(row + 2 * b), v = *(const float2*)(row + plane + 2 * b);
        return make_float4(u.x, u.y, v.x, v.y);
      }
      return *(const float4*)(row + 4 * b);
    };
    auto acc_store = [&](float* row, const uint32_t b, const float4 a) {
      if (planes2) {
        *(float2*)(row + 2 * b) = make_float2(a.x, a.y);
        *(float2*)(row + plane + 2 * b) = make_float2(a.z, a.w);
      } else {
        *(float4*)(row + 4 * b) = a;
      }
    };
    auto mid = [&](const int phase) {
      if constexpr (PRE) {
        if (phase == 2 && in_lds && tid < (f_nact << logT3)) {
          en_pre = planl[tid >> logT3];
          acc_pre = acc_load(acc_row(tid), en_pre.x);
        }
        // (many channels per tile: a part's active bins x Fb channels exceed the workgroup, so a thread folds a second
        //  item -- its accumulator is requested here as well instead of costing a memory round trip in the fold phase)
        if constexpr (PRE2) {
          if (phase == 2 && in_lds && tid + 512u < (f_nact << logT3)) {
            en_pre2 = planl[(tid + 512u) >> logT3];
            acc_pre2 = acc_load(acc_row(tid + 512u), en_pre2.x);
          }
        }
        if (phase == 2 && plan_dma_ok && more) plan_dma(next, (jt + 1) & 1);
      }
    };
    if constexpr (HALF) {
      // last stage of the half-length transforms -> radix-2 step in registers -> the ordinary store
      auto store_half = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
        constexpr int R = sizeof(v) / sizeof(v[0]);
        const uint32_t odd = (col >> 1) & 1u;                               // bin parity of this lane's pair (even lanes E, odd lanes O)
        apply_pass_twiddle_inv<R>(v, odd, p, pstride, LOGM, tw, g.tw_lo_m);   // O[m] *= exp(+2 pi i m / M), m = p + k*pstride (E: x 1)
        const float sg = odd ? -1.0f : 1.0f;
        auto swp = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, false)); };
#pragma unroll
        for (int k = 0; k < R; k++) {
          const cx2 own = v[k];
          cx2 rc;
          rc.x = (v2f){swp(own.x[0]), swp(own.x[1])};
          rc.y = (v2f){swp(own.y[0]), swp(own.y[1])};
          v[k].x = own.x * sg + rc.x;                                       // even: E + W O = y[m]   odd: E - W O = y[m + M/2]
          v[k].y = own.y * sg + rc.y;
        }
        store((col >> 2) << 1, p + (odd << (LOGM - 1)), pstride, v);
      };
      wgfft<LOGFI, +1, FOLD, true>(lds, ltw_off, tid, logT, x, store_half, mid);
    } else {
      wgfft<LOGM, +1, FOLD, true>(lds, ltw_off, tid, logT, x, store, mid);
    }
#if defined(FB_STAMPS) && FB_STAMPS == 7
    STAMP(ts3);
#endif
    if constexpr (FOLD) {
      __syncthreads();
      const bool pre = PRE && in_lds;
      for (uint32_t w = tid; w < (f_nact << logT3); w += blockDim.x) {
        const uint32_t slo = w & (T3 - 1);
        uint4 en;
        float4 acc;
        if (pre && w == tid) {
          en = en_pre;
          acc = acc_pre;
        } else if (PRE2 && pre && w == tid + 512u) {
          en = en_pre2;
          acc = acc_pre2;
        } else {
          en = in_lds ? planl[w >> logT3] : ent[w >> logT3];
          acc = acc_load(acc_row(w), en.x);
        }
        const uint32_t nint = en.z >> 16;
        float* __restrict__ pp = acc_row(w);
        uint32_t off = en.w, hits = en.z & 0xffffu;
        for (uint32_t i = 0;;) {
          const float4* __restrict__ src = (const float4*)&lds[2 * (slo * fcs + off)];
          uint32_t h = 0;
          for (; h + 8 <= hits; h += 8) {
            float4 sm[8];
#pragma unroll
            for (int q = 0; q < 8; q++) sm[q] = src[h + q];
#pragma unroll
            for (int q = 0; q < 8; q++) { acc.x += sm[q].x; acc.y += sm[q].y; acc.z += sm[q].z; acc.w += sm[q].w; }
          }
          for (; h < hits; h++) {
            const float4 sm = src[h];
            acc.x += sm.x; acc.y += sm.y; acc.z += sm.z; acc.w += sm.w;
          }
          if (++i >= nint) break;
          const Interval iv = out.piv[en.y + i];
          off = (uint32_t)iv.offset; hits = iv.hits;
        }
        acc_store(pp, en.x, acc);
      }
    }
#if defined(FB_STAMPS) && FB_STAMPS == 7
    STAMP(ts4);
    acc_s[0] += ts0 - ts5; acc_s[1] += ts1 - ts0; acc_s[2] += ts2 - ts1; acc_s[3] += ts6 - ts2; acc_s[4] += ts3 - ts6; acc_s[6] += ts4 - ts3; acc_s[5] += 1;
    ts5 = ts4;
#endif
    if (!more) break;
    item = next;
    jt++;
  }
#if defined(FB_STAMPS) && FB_STAMPS == 7
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 7; q++) atomicAdd(&g_stamps[blockIdx.x][q], acc_s[q]);
#endif
}
#endif  // FB_HAS(6)

#if FB_HAS(4)
// ------------------------------------------------------------------------------------ P3a / P3b
// Two-pass inverse transform for freq_res = Ma*Mb beyond one workgroup tile (and for nchan_subband = 1,
// i.e. dsp::Convolution): bin m = m1*Mb + m2, output sample t = t1 + Ma*t2.
//   P3a k_inv_a : spectrum X (bin k = c*freq_res + m, blocked by pass-2 tile: FbGeom::xblocked) -> Hermitian split / pol
//                 select -> x chirp
//                 -> inverse Ma-point FFTs over m1 for Tm adjacent m2 -> x conj(W_M^{m2*t1})
//                 -> U[c][t1/Tt][m2][t1%Tt][pol]
//   P3b k_inv_b : inverse Mb-point FFTs over m2 for Tt adjacent t1 (one contiguous block of U)
//                 -> keep window on t = t1 + Ma*t2 -> complex output or fused detection
// Columns of both tiles are (column, pol) pairs, so the thread's two butterflies are the two polarisations.
// REAL: real input (one packed sequence per part, the polarisations separated by the Hermitian split) or complex input
// (npol sequences), fixed at compile time: the split loop and the mirror addresses are then free of per-element branches
// (the run-time form cost one uniform branch per element and load, 5577 ISA lines per tile at -x 262144).
// FULL: the tile is the whole workgroup tile (Ma * 2*Tm = 2^14 elements, 512 threads): the column count is then a
// compile-time constant and the exchange addresses of the transform fold (wgfft_stage's uniform selects otherwise cost
// two branches per element: 107 per tile in the -x 262144 listing).
#ifndef FB_INVA_NT_LD
#define FB_INVA_NT_LD 1        // experiment (0): k_inv_a reads the spectrum with plain loads
#endif
#ifndef FB_INVA_NT_ST
#define FB_INVA_NT_ST 1        // experiment (0): k_inv_a writes U with plain stores
#endif
#if FB_INVA_NT_LD
#define INVA_LD(p) ld_stream(p)
#else
#define INVA_LD(p) (*(p))
#endif
#if FB_INVA_NT_ST
#define INVA_ST(p, v) st_stream(p, v)
#else
#define INVA_ST(p, v) (*(p) = (v))
#endif
template <int LOGF, bool BLOCKED, bool REAL, bool FULL>
__global__ __launch_bounds__(512) void k_inv_a(const FbGeom g, const cf* __restrict__ X, const cf* __restrict__ kernel,
                                               cf* __restrict__ U, const cf* __restrict__ tw, const uint32_t nparts,
                                               const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logTm = FULL ? 13 - LOGF : g.logTm, logT = logTm + 1, logTt = g.logTt;
  const uint32_t Tm = 1u << logTm, Tt = 1u << logTt;
  const uint64_t L = 1ull << (g.logM + g.logR);
  const uint32_t nseq = REAL ? 1 : g.npol;
  const bool npol2 = g.npol == 2;
  const uint32_t ntile = 1u << (g.logMb - logTm);          // m2 tiles per channel
  const uint32_t per_part = ntile * g.C;
  const uint32_t total = per_part * nparts;
  const int logNt = g.logMb - logTm;        // ntile = 2^logNt
  struct Abk { cf a, b; };

  // Blocked spectrum (FbGeom::xblocked, written by k_fwd_rows): bin k = ka + Fa*kb lies at
  // (ka >> logT2)*xblock + (kb << logT2 | ka % T2).  The tile's Ma x Tm bins are enumerated in MEMORY order -- the low
  // bits of m2 (inside a run of T2), then the bits of m1 that fall into kb (consecutive in memory), then the rest -- so a
  // wave's load covers whole runs (1 KB at -F 64:D -x 262144) although a first-stage butterfly needs bins Ma/16 rows apart;
  // the split and chirp-multiplied elements change to butterfly order through the exchange buffer (one extra LDS round trip).
  // Loading in butterfly order instead touches 32-byte pieces 128 KB apart: +54 % on this pass (r02 experiments, item 18).
  const int nlow = g.logT2 < logTm ? g.logT2 : logTm;
  const int hs0 = g.logM - g.logMb, hs = hs0 < 0 ? 0 : (hs0 > LOGF ? LOGF : hs0);    // m1 bits below `hs` stay in ka
  const int nhh = LOGF - hs, nlh = logTm - nlow;
  const uint32_t maskA = (1u << g.logM) - 1, maskT = (1u << g.logT2) - 1;
  auto tile_elem = [&](const uint32_t e, uint32_t& m1, uint32_t& j) {
    const uint32_t w = e & ((1u << nlow) - 1), e1 = e >> nlow;
    const uint32_t mhh = e1 & ((1u << nhh) - 1), e2 = e1 >> nhh;
    j = ((e2 & ((1u << nlh) - 1)) << nlow) | w;
    m1 = (mhh << hs) | (e2 >> nlh);
  };
  auto xa = [&](const uint32_t k) -> uint32_t {            // k < L
    const uint32_t ka = k & maskA, kb = k >> g.logM;
    return (ka >> g.logT2) * g.xblock + ((kb << g.logT2) | (ka & maskT));
  };
  auto xk = [&](const uint32_t k) -> uint32_t {            // the chirp, permuted likewise by set_kernel (no padding)
    const uint32_t ka = k & maskA, kb = k >> g.logM;
    return (ka >> g.logT2) * g.kblock + ((kb << g.logT2) | (ka & maskT));
  };
  // The enumeration is a permutation of index BITS, and the element index of a thread's i-th element is tid + i*nthr
  // (nthr a power of two), so bin index, spectrum address, chirp address and staging address of that element all split
  // into a part that depends on the thread, a part that depends on the item (uniform) and one uniform increment per
  // bit of i: disjoint bit fields add.  16 elements then cost one vector add each instead of a full decode (the decode
  // per element made this pass issue 2650 vector instructions per thread and tile, 42 % of them integer).
  auto stg = [&](const uint32_t m1, const uint32_t j) { return lds_pad(((m1 << logTm) + j) << 1); };
  uint32_t thr_k, thr_st, Dk[4], Dxa[4], Dxk[4], Dst[4];
  {
    uint32_t m1, j;
    tile_elem(tid, m1, j);
    thr_k = (m1 << g.logMb) + j;
    thr_st = stg(m1, j);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      tile_elem((uint32_t)blockDim.x << b, m1, j);
      Dk[b] = (m1 << g.logMb) + j;
      Dxa[b] = xa(Dk[b]);
      Dxk[b] = xk(Dk[b]);
      Dst[b] = stg(m1, j);
    }
  }
  const uint32_t thr_xa = xa(thr_k), thr_xk = xk(thr_k);
  // Mirror bin L - k = ~k + 1 (real input).  ~k complements every bit field, so its address is xa(L-1) - xa(k); the + 1
  // adds 1 when the low T2 bits of k are not all zero, else carries into the row-group field (+ xblock - (T2-1)), else
  // into kb.  Which case applies is decided by the thread/item part of k unless the increment of element i reaches into
  // ka (uniform test): one select and one subtraction per element instead of a second full address computation.
  const uint32_t XAM = xa((uint32_t)L - 1), dCarryA = g.xblock - maskT,
                 dCarryB = (1u << g.logT2) - maskT - (maskA >> g.logT2) * g.xblock;
  // (the element increments never reach the low T2 bits of k: dspsr_amd_filterbank_create uses the blocked layout only
  //  when Mb >= T2 and the workgroup has at least T2 threads)
  auto inc = [&](const uint32_t (&D)[4], const int i) {
    return ((i & 1) ? D[0] : 0u) + ((i & 2) ? D[1] : 0u) + ((i & 4) ? D[2] : 0u) + ((i & 8) ? D[3] : 0u);
  };

  // The prefetch of the next tile is issued in NCH groups spread over the tile -- behind the split, inside the order
  // exchange, behind the butterflies of the first stages (wgfft's `mid` hook) -- instead of one burst of 32 loads per thread:
  // the burst blocked every wave in its load instructions for a quarter of the tile while the memory pipeline, still
  // draining the previous tile's stores, accepted them (stamps: 11.6k of 45.7k cycles, and as many again at the next barrier).
  // chunk < 0: all elements; otherwise the elements i with i % NCH == chunk.
#ifndef FB_INVA_NMID_MAX
#define FB_INVA_NMID_MAX 2     // experiment: 0 = the prefetch behind the split and inside the order exchange only
#endif
  // (measured at cfg1opt / cfg1, same box: one burst 63.7k / 10.0k Msamples/s, two groups 67.2k / 9.9k, four 77.3k / 10.3-10.6k,
  //  five 75.1k, six 77.1k: four it is -- the natural order has no exchange to hide a group in and takes one in front of the
  //  copy-out stores instead)
  constexpr int CO_CHUNK = BLOCKED ? 0 : 1;
  constexpr int NMID0 = P::NS, NMID = NMID0 < FB_INVA_NMID_MAX ? NMID0 : FB_INVA_NMID_MAX;
#ifdef FB_INVA_ONE_BURST
  constexpr int NCH = 1;
#else
  constexpr int NCH = 1 + (BLOCKED ? 1 : 0) + NMID + CO_CHUNK;
#endif
  auto fetch = [&](const uint32_t item, Abk (&raw)[PTS / 2], const int chunk) {
    const uint32_t r = (FB_DBG(g) & 256) ? item % per_part : item / nparts, part = (FB_DBG(g) & 256) ? item / per_part : item - r * nparts;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    const cf* __restrict__ X0s = X + (uint64_t)part * nseq * g.xstride;
    if (FB_DBG(g) & 2) {
      if (chunk <= 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; i++) { raw[i].a = make_float2(tid, i); raw[i].b = raw[i].a; }
      }
      return;
    }
    if constexpr (BLOCKED) {
      const uint32_t kt = (c << g.logMf) + tile * Tm;
      const uint32_t k0 = kt + thr_k, a0 = xa(kt) + thr_xa;
      if constexpr (REAL) {
        const uint32_t lowT = k0 & maskT, lowA = k0 & maskA;
        const uint32_t E2 = XAM + (lowT ? 1u : (lowA ? dCarryA : dCarryB));            // increment in kb only
        const uint32_t dE = XAM + (lowT ? 1u : dCarryA) - E2;                          // increment reaches into ka: E2 + dE
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) {
          if (chunk >= 0 && i % NCH != chunk) continue;
          const uint32_t ia = a0 + inc(Dxa, i);
          const uint32_t into_ka = (inc(Dk, i) & maskA) ? 1u : 0u;                     // uniform: a scalar, no branch
          uint32_t ib = E2 + into_ka * dE - ia;
          if (i == 0) ib = k0 == 0 ? 0u : ib;                                          // bin 0 is its own mirror
          Abk q;
          q.a = INVA_LD(X0s + ia);
          q.b = INVA_LD(X0s + ib);
          raw[i] = q;
        }
      } else {
        const cf* __restrict__ X1s = npol2 ? X0s + g.xstride : X0s;                    // (one polarisation: loaded twice, zeroed below)
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) {
          if (chunk >= 0 && i % NCH != chunk) continue;
          const uint32_t ia = a0 + inc(Dxa, i);
          Abk q;
          q.a = INVA_LD(X0s + ia);
          q.b = INVA_LD(X1s + ia);
          raw[i] = q;
        }
      }
      return;
    }
    // element i of the first-stage butterfly is bin k0 + i*step (m1 advances by MS): base plus a multiple of a
    // wave-uniform step; the mirror bin L - k runs down with the same step (k = 0, its own mirror, can only be i = 0)
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const int64_t step = (int64_t)MS << g.logMb;
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;
      const uint32_t j = (eb & ((1u << logT) - 1)) >> 1, m1b = eb >> logT;
      const uint64_t k0 = ((uint64_t)c << g.logMf) + ((uint64_t)m1b << g.logMb) + tile * Tm + j;
      const cf* __restrict__ pa = X0s + k0;
      const cf* __restrict__ pb = REAL ? X0s + (L - k0) : pa + (npol2 ? L : 0);
      const int64_t stepb = REAL ? -step : step;
      const cf* __restrict__ pb0 = (REAL && k0 == 0) ? X0s : pb;
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        if (chunk >= 0 && ((g2 / 2) * P::R1 + i) % NCH != chunk) continue;
        Abk q;
        q.a = INVA_LD(pa + i * step);
        q.b = INVA_LD(i == 0 ? pb0 : pb + i * stepb);
        raw[(g2 / 2) * P::R1 + i] = q;
      }
    }
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  // Items are (tile, part) pairs with the part running fastest, and every workgroup takes one contiguous range of them: it
  // walks the parts of a tile one after the other, so the tile's chirp is loaded once and stays in registers
  // (one chirp read per launch instead of one per part: -1/7 of this pass's traffic at 8 parts per launch).
  uint32_t item = (uint32_t)(((uint64_t)total * blockIdx.x) / gridDim.x);
  const uint32_t item_end = (uint32_t)(((uint64_t)total * (blockIdx.x + 1)) / gridDim.x);
  if (item >= item_end) return;
  uint32_t next;
  Abk raw[PTS / 2];
  fetch(item, raw, -1);
  // The chirp of a tile stays in registers while the workgroup walks the tile's parts (loaded when the tile changes: 16
  // loads per thread less on 7 of 8 items; cfg1opt +2.5 %).  32 registers: the full-tile kernels fit them with 0-7 spilled
  // registers, except the blocked ones whose stages are all radix 16 (13-18 spills: those re-read the chirp per part from the
  // L2, as the generic kernels do).
  constexpr bool KEEPK = FULL && !(BLOCKED && LOGF % 4 == 0 && LOGF > 0);
  cf kk[PTS / 2];
  uint32_t kk_r = ~0u;
#if defined(FB_STAMPS) && FB_STAMPS == 4
  unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, ts6, acc_s[7] = {0, 0, 0, 0, 0, 0, 0};
  STAMP(ts5);
#endif
  for (;;) {
    asm volatile("" : "+v"(tid));
    const uint32_t r = (FB_DBG(g) & 256) ? item % per_part : item / nparts, part = (FB_DBG(g) & 256) ? item / per_part : item - r * nparts;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    cx2 x[NPAIR];
#if defined(FB_STAMPS) && FB_STAMPS == 4
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(ts0);
#endif
    {
      if (!KEEPK || r != kk_r) {
      kk_r = r;
      if (BLOCKED && kernel && !(FB_DBG(g) & (2 | 4))) {
        const uint32_t c0 = xk((c << g.logMf) + tile * Tm) + thr_xk;
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) kk[i] = kernel[c0 + inc(Dxk, i)];
      } else if (kernel && !(FB_DBG(g) & 2)) {
        constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
#pragma unroll
        for (int g2 = 0; g2 < P::G1; g2 += 2) {
          const uint32_t eb = P::G1 * tid + g2;
          const cf* __restrict__ pk = kernel + ((uint64_t)c << g.logMf) + ((uint64_t)(eb >> logT) << g.logMb) + tile * Tm +
                                      ((eb & ((1u << logT) - 1)) >> 1);
#pragma unroll
          for (int i = 0; i < P::R1; i++) kk[(g2 / 2) * P::R1 + i] = pk[((uint64_t)i * MS) << g.logMb];
        }
      } else {
#pragma unroll
        for (int i = 0; i < PTS / 2; i++) kk[i] = make_float2(1.f, 0.f);
      }
      }
#pragma unroll
      for (int i = 0; i < PTS / 2; i++) {
        const Abk q = raw[i];
        cf x0, x1;
        if constexpr (REAL) {
          x0 = make_float2(0.5f * (q.a.x + q.b.x), 0.5f * (q.a.y - q.b.y));
          x1 = make_float2(0.5f * (q.a.y + q.b.y), 0.5f * (q.b.x - q.a.x));
        } else {
          x0 = q.a;
          x1 = npol2 ? q.b : make_float2(0.f, 0.f);
        }
        x[i] = cmuls(make_cx2(x0, x1), kk[i]);
      }
    }
    next = item + 1;
    const bool more = next < item_end;
#if defined(FB_STAMPS) && FB_STAMPS == 4
    STAMP(ts1);
#endif
    // unconditional (the last item of the range is fetched again and dropped: 1/64 of the reads at 8 parts per launch).  Under
    // `if (more)` the loads went to fresh registers and the copies into `raw` at the end of the conditional block waited for
    // them (`s_waitcnt vmcnt(0)` straight behind the 32 loads in the ISA): the prefetch overlapped nothing.
    const uint32_t nitem = more ? next : item;
    fetch(nitem, raw, 0);
#if defined(FB_STAMPS) && FB_STAMPS == 4
    STAMP(ts2);
#endif
    if constexpr (BLOCKED) {
      // memory order -> butterfly order: element (m1, j) of the tile (both polarisations, 16 bytes) at word pair m1*Tm + j
      __syncthreads();                         // the previous tile's copy-out has finished with the buffer
#pragma unroll
      for (int i = 0; i < PTS / 2; i++)
        *(float4*)&lds[thr_st + inc(Dst, i)] = make_float4(x[i].x[0], x[i].x[1], x[i].y[0], x[i].y[1]);
      if (NCH > 1) fetch(nitem, raw, 1);
      __syncthreads();
#pragma unroll
      for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
        for (int i = 0; i < P::R1; i++) {
          const float4 pr = *(const float4*)&lds[lds_pad(first_stage_elem<LOGF>(tid, logT, g2, i))];
          x[(g2 / 2) * P::R1 + i].x = (v2f){pr.x, pr.y};
          x[(g2 / 2) * P::R1 + i].y = (v2f){pr.z, pr.w};
        }
      __syncthreads();                         // before the first stage's exchange overwrites the buffer
    }

#if defined(FB_STAMPS) && FB_STAMPS == 4
    STAMP(ts6);
#endif
    cf* __restrict__ Uc = U + ((uint64_t)part * g.C + c) * (2ull << g.logMf);
    // staged image order [t1/Tt][j][t1%Tt][pol]: whole runs of Tm*Tt*2 elements go out with 16-byte stores
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      const uint32_t j = col >> 1;
      apply_pass_twiddle_inv<R>(v, tile * Tm + j, p, pstride, g.logMf, tw, g.tw_lo_m);
#pragma unroll
      for (int k = 0; k < R; k++) {
        const uint32_t t1 = k * pstride + p;
        const uint32_t l = (((((t1 >> logTt) << logTm) + j) << logTt) | (t1 & (Tt - 1))) << 1;
        *(float4*)&lds[lds_pad(l)] = make_float4(v[k].x[0], v[k].y[0], v[k].x[1], v[k].y[1]);
      }
    };
    auto mid = [&](const int phase) {
      if (NCH > 1 && phase >= 1 && phase <= NMID) fetch(nitem, raw, (BLOCKED ? 1 : 0) + phase);
    };
    wgfft<LOGF, +1, true>(lds, ltw_off, tid, logT, x, store, mid);
    __syncthreads();
    if (CO_CHUNK && NCH > 1) fetch(nitem, raw, NCH - 1);
#if defined(FB_STAMPS) && FB_STAMPS == 4
    STAMP(ts3);
#endif
    {
      const uint32_t nthr = blockDim.x;
      const int logRun = logTm + logTt + 1;
      const uint32_t n2 = 2 * nthr;
      if ((n2 & 63) == 0 && (n2 >> logRun) != 0 && (n2 & ((1u << logRun) - 1)) == 0) {      // uniform
        // pair jj = pair 0 + jj*2*nthr elements: constant step in the padded image, uniform step in U (see pass 1)
        const uint32_t l0 = 2 * tid, lstep = n2 + ((n2 >> 6) << 2), lb = lds_pad(l0);
        const uint32_t goff = (uint32_t)(((((uint64_t)(l0 >> logRun) << g.logMb) << (logTt + 1)) + (l0 & ((1u << logRun) - 1))) * sizeof(cf));
        const uint64_t gstep = (((uint64_t)(n2 >> logRun) << g.logMb) << (logTt + 1)) * sizeof(cf);
        const char* __restrict__ gb = (const char*)(Uc + ((uint64_t)(tile * Tm) << (logTt + 1)));
#pragma unroll
        for (int j4 = 0; j4 < PTS / 2; j4 += 4) {
          float4 pr[4];
#pragma unroll
          for (int q = 0; q < 4; q++) pr[q] = *(const float4*)&lds[lb + (j4 + q) * lstep];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (!(FB_DBG(g) & 1)) INVA_ST((float4*)(gb + (uint64_t)(j4 + q) * gstep + goff), pr[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 4
        for (int jj = 0; jj < PTS / 2; jj++) {
          const uint32_t l = 2 * (tid + jj * nthr);
          const uint32_t tb = l >> logRun, within = l & ((1u << logRun) - 1);
          const float4 pr = *(const float4*)&lds[lds_pad(l)];
          INVA_ST((float4*)&Uc[((((uint64_t)tb << g.logMb) + tile * Tm) << (logTt + 1)) + within], pr);
        }
      }
    }
#if defined(FB_STAMPS) && FB_STAMPS == 4
    STAMP(ts4);
    acc_s[0] += ts0 - ts5; acc_s[1] += ts1 - ts0; acc_s[2] += ts2 - ts1; acc_s[3] += ts6 - ts2; acc_s[4] += ts3 - ts6; acc_s[6] += ts4 - ts3; acc_s[5] += 1;
    ts5 = ts4;
#endif
    if (!more) break;
    item = next;
  }
#if defined(FB_STAMPS) && FB_STAMPS == 4
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int q = 0; q < 7; q++) atomicAdd(&g_stamps[blockIdx.x][q], acc_s[q]);
#endif
}

// FOLDB (FbOut kind 4): the tile holds, for one channel, Mb runs of Tt consecutive output samples (run t2 = samples
// t1 + Ma*t2, t1 in the tile's block): exactly the micro-blocks the long-run fold (fold.hip, FOLD_LONG_RUN) sums first.
// The detected samples are staged in the exchange buffer ([t2][t1], XOR-swizzled so that both the stage's writes and the
// per-run reads are conflict free); thread t2 adds its run in time order, cut at the one phase-bin boundary it may hold
// (the host admits this path only for plans whose inner intervals are >= Tt samples), and writes the two piece sums --
// 1/16 of the detected bytes instead of all of them.  fold_segment_combine (fold.hip) then adds, per (channel, bin), the
// pieces of the bin's intervals in time order.  Deterministic; equal to the time-order sum to float rounding like the
// long-run fold itself (other micro-block boundaries, so not bit-equal to it).
// FULL: whole workgroup tile (Mb * 2*Tt = 2^14 elements, 512 threads), column count fixed at compile time (see k_inv_a).
template <int LOGF, bool FOLDB, bool FULL>
__global__ __launch_bounds__(512) void k_inv_b(const FbGeom g, const cf* __restrict__ U, const FbOut out,
                                               const cf* __restrict__ tw, const uint64_t part0, const uint32_t nparts,
                                               const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logTt = FULL ? 13 - LOGF : g.logTt, logT = logTt + 1;
  const uint32_t ntile = 1u << (g.logMa - logTt);          // t1 blocks per channel
  const uint32_t per_part = ntile * g.C;
  const uint32_t total = per_part * nparts;
  const int logNt = g.logMa - logTt;        // ntile = 2^logNt

  auto fetch = [&](const uint32_t item, float4 (&y)[NPAIR]) {
    const uint32_t part = item / per_part, r = item - part * per_part;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    const cf* __restrict__ blk = U + ((uint64_t)part * g.C + c) * (2ull << g.logMf) + (((uint64_t)tile << g.logMb) << logT);
    if (FB_DBG(g) & 2) {
#pragma unroll
      for (int i = 0; i < NPAIR; i++) y[i] = make_float4(tid, i, 1.f, 1.f);
      return;
    }
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++)
        y[(g2 / 2) * P::R1 + i] = ld_stream((const float4*)&blk[first_stage_elem<LOGF>(tid, logT, g2, i)]);
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  uint32_t item, next;
  uint32_t jn = 0;
  if (!persistent_item(blockIdx.x, gridDim.x, jn, run, total, item)) return;
  float4 y[NPAIR];
  fetch(item, y);
  for (;;) {
    asm volatile("" : "+v"(tid));
    cx2 x[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; i++) x[i] = make_cx2(make_float2(y[i].x, y[i].y), make_float2(y[i].z, y[i].w));
    const uint32_t lpart = item / per_part, r = item - lpart * per_part;
    const uint64_t part = part0 + lpart;
    const uint32_t c = r >> logNt, tile = r & (ntile - 1);
    const bool more = persistent_item(blockIdx.x, gridDim.x, ++jn, run, total, next);
    if (more) fetch(next, y);

    // (uniform output kind / ndim decided once per butterfly, not per element: the compiler does not unswitch them out of
    //  the unrolled loop; the keep window is the only per-element test)
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if constexpr (FOLDB) {
        // detected sample (run t2 = k*pstride + p, position j = col/2 in the run) -> float4 slot t2*Tt + (j ^ t2 % Tt)
        const uint32_t j = col >> 1, Ttm = (1u << logTt) - 1;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t t2 = k * pstride + p;
          float q[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, q);
          *(float4*)&lds[2 * ((t2 << logTt) + (j ^ (t2 & Ttm)))] = make_float4(q[0], q[1], q[2], q[3]);
        }
        return;
      }
      if (out.kind == 0) return;
      const uint32_t chan = out.chan0 + c;
      const uint32_t t1 = (tile << logTt) + (col >> 1);
      float* __restrict__ row = out.base + chan * out.chan_stride;
      auto each = [&](auto&& emit) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const uint32_t pos = ((k * pstride + p) << g.logMa) + t1;
          if (pos < g.nfilt_pos || pos >= g.nfilt_pos + g.nkeep) continue;
          emit(pos - g.nfilt_pos, cx2_lo(v[k]), cx2_hi(v[k]));
        }
      };
      if (out.kind == 1) {
        float2* __restrict__ o0 = (float2*)(row + part * out.part_step);
        if (g.npol == 2)
          each([&](const uint32_t t, const cf va, const cf vb) {
            st_stream(o0 + t, va);
            st_stream((float2*)((float*)(o0 + t) + out.pol_stride), vb);
          });
        else
          each([&](const uint32_t t, const cf va, const cf) { st_stream(o0 + t, va); });
        return;
      }
      const uint64_t idat0 = part * g.nkeep;
      if (out.ndim == 4) {
        float4* __restrict__ o = (float4*)row + idat0;
        each([&](const uint32_t t, const cf va, const cf vb) {
          float q[4];
          detect4(va, vb, out.state, q);
          st_stream(o + t, make_float4(q[0], q[1], q[2], q[3]));
        });
      } else if (out.ndim == 2) {
        float2* __restrict__ o = (float2*)row + idat0;
        float2* __restrict__ o1 = (float2*)(row + out.pol_stride) + idat0;
        each([&](const uint32_t t, const cf va, const cf vb) {
          float q[4];
          detect4(va, vb, out.state, q);
          st_stream(o + t, make_float2(q[0], q[1]));
          st_stream(o1 + t, make_float2(q[2], q[3]));
        });
      } else {
        float* __restrict__ o = row + idat0;
        each([&](const uint32_t t, const cf va, const cf vb) {
          float q[4];
          detect4(va, vb, out.state, q);
          o[t] = q[0];
          o[out.pol_stride + t] = q[1];
          o[2 * out.pol_stride + t] = q[2];
          o[3 * out.pol_stride + t] = q[3];
        });
      }
    };
    wgfft<LOGF, +1, FOLDB>(lds, ltw_off, tid, logT, x, store);
    if constexpr (FOLDB) {
      __syncthreads();                                       // the tile's detected samples are staged
      const uint32_t Tt = 1u << logTt, Ttm = Tt - 1;
      for (uint32_t t2 = tid; t2 < (1u << LOGF); t2 += blockDim.x) {
        // run t2: output positions pos0 .. pos0 + Tt - 1 of the backward transform; kept: [nfilt_pos, nfilt_pos + nkeep)
        const uint32_t pos0 = (t2 << g.logMa) + (tile << logTt);
        const uint32_t lo = g.nfilt_pos, hi = g.nfilt_pos + g.nkeep;
        const uint32_t jlo = pos0 >= lo ? 0u : (lo - pos0 < Tt ? lo - pos0 : Tt), jhi = pos0 + Tt <= hi ? Tt : (hi > pos0 ? hi - pos0 : 0u);
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
        if (jlo < jhi) {
          const uint32_t i0 = (uint32_t)part * g.nkeep + (pos0 + jlo - g.nfilt_pos), i1 = i0 + (jhi - jlo);   // sample span in the block
          uint32_t qi = out.blk_first[i0 >> 10];
          while (out.pstart[qi + 1] <= i0) qi++;             // interval that holds sample i0 (inner intervals are >= Tt samples)
          const uint32_t cut = out.pstart[qi + 1] < i1 ? out.pstart[qi + 1] : i1;
          const uint32_t jc = jlo + (cut - i0);
          const float4* __restrict__ src = (const float4*)&lds[2 * (t2 << logTt)];
          for (uint32_t j = jlo; j < jc; j++) { const float4 q = src[j ^ (t2 & Ttm)]; sa.x += q.x; sa.y += q.y; sa.z += q.z; sa.w += q.w; }
          for (uint32_t j = jc; j < jhi; j++) { const float4 q = src[j ^ (t2 & Ttm)]; sb.x += q.x; sb.y += q.y; sb.z += q.z; sb.w += q.w; }
        }
        float4* __restrict__ o = (float4*)out.base + ((((uint64_t)c * out.nparts_plan + part) * ntile + tile) << (LOGF + 1)) + 2 * t2;
        o[0] = sa;
        o[1] = sb;
      }
      // (the next tile's first exchange write sits behind a barrier: wgfft)
    }
    if (!more) break;
    item = next;
  }
}

#endif  // FB_HAS(4)

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
k3a_t fb_pick3a(int logf, bool blocked, bool real, bool full);
k3b_t fb_pick3b(int logf, bool foldb, bool full);
// two-pass path (FB_HAS(6)): pass 1 on whole columns, rows + inverse pass (M = 2^logm, Fb = 2^(13 - logm)), the 8-bit regroup
typedef void (*k1c_t)(FbGeom, FbIn, cf*, const cf*, uint32_t, uint32_t, uint32_t);
k1c_t fb_pick_col1(int variant = 1);     // 1: four sub-sequences + radix-4 in registers (three exchanged stages), 0: even / odd + radix-2
k3_t fb_pick_rinv(int logm, bool fold);
void fb_launch_raw_cols(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0);
void fb_launch_sub_split(hipStream_t stream, const SubSplit& p, uint8_t* out, uint32_t ncu);
void fb_launch_sub_combine(hipStream_t stream, const FbGeom& g, cf* X, uint32_t nseqs, uint32_t ncu, cf* Xout = nullptr, uint32_t mo = 0,
                           uint32_t rm = 1);
void fb_launch_time_combine(hipStream_t stream, const TimeCombine& p, const FbOut& out, uint32_t R, uint32_t ncu);
void fb_launch_raw_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0);
void fb_launch_float_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, cf* Rt, uint64_t part0);

#ifdef FB_ONLY_HEADLINE   // experiment builds: only the kernels of the headline geometry (M = 4096, Rr = 2048, 8-bit)
#if FB_HAS(1)
k1_t fb_pick1_dual(int) { return nullptr; }
k1_t fb_pick1(int logf, int raww, bool full)
{
  if (logf == 12 && raww == 2 && full) return k_fwd_cols<12, 2, 2>;
#ifdef FB_P1_HALF_TILE   // experiment: 2^13-point tiles (two columns), two workgroups per CU, compile-time tile shape
  return logf == 12 && raww == 1 ? (full ? k_fwd_cols<12, 1, 2> : k_fwd_cols<12, 1, 1>) : nullptr;
#else
  return logf == 12 && raww == 1 ? (full ? k_fwd_cols<12, 1, 2> : k_fwd_cols<12, 1, -1>) : nullptr;
#endif
}
#endif
#if FB_HAS(2)
k2_t fb_pick2(int logf, bool full) { return logf == 11 ? (full ? k_fwd_rows<11, 3> : k_fwd_rows<11, -1>) : nullptr; }
#endif
#if FB_HAS(3)
k3_t fb_pick3(int logf, bool full) { return logf != 12 ? nullptr : (full ? k_inv_chan<12, false, 2> : k_inv_chan<12, false, -1>); }
void fb_launch_time_combine(hipStream_t stream, const TimeCombine& p, const FbOut& out, uint32_t R, uint32_t ncu)
{
  switch (R) {
    case 3: hipLaunchKernelGGL(k_time_combine<3>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 5: hipLaunchKernelGGL(k_time_combine<5>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 7: hipLaunchKernelGGL(k_time_combine<7>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 9: hipLaunchKernelGGL(k_time_combine<9>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    case 15: hipLaunchKernelGGL(k_time_combine<15>, dim3(8 * ncu), dim3(256), 0, stream, p, out); break;
    default: break;
  }
}
#endif
#if FB_HAS(5)
k3_t fb_pick3f(int logf, bool full) { return logf != 12 ? nullptr : (full ? k_inv_chan<12, true, 2> : k_inv_chan<12, true, -1>); }
#endif
#if FB_HAS(4)
k3a_t fb_pick3a(int, bool, bool, bool) { return nullptr; }
k3b_t fb_pick3b(int, bool, bool) { return nullptr; }
#endif
#if FB_HAS(6)
k1c_t fb_pick_col1(int) { return nullptr; }
k3_t fb_pick_rinv(int, bool) { return nullptr; }
void fb_launch_raw_cols(dim3, hipStream_t, const FbGeom&, const FbIn&, uint16_t*, uint64_t) {}
#endif
#else
#if FB_HAS(1)
template <int... I> static k1_t pick1(int logf, int raww, bool full, iseq<I...>)
{
  static const k1_t t4[] = {k_fwd_cols<I, 4, -1>...};
  static const k1_t t1[] = {k_fwd_cols<I, 1, -1>...};
  static const k1_t f4[] = {k_fwd_cols<I, 4, full_logt(I)>...};
  static const k1_t f1[] = {k_fwd_cols<I, 1, full_logt(I)>...};
  return full ? (raww == 1 ? f1[logf] : f4[logf]) : (raww == 1 ? t1[logf] : t4[logf]);
}
k1_t fb_pick1(int logf, int raww, bool full) { return pick1(logf, raww, full, seq_t()); }
// (8-bit input only: with float32 input the two tiles' prefetch alone is 128 registers)
k1_t fb_pick1_dual(int raww) { return raww == 1 ? k_fwd_cols_dual<1> : nullptr; }
#endif
#if FB_HAS(2)
template <int... I> static k2_t pick2(int logf, bool full, iseq<I...>)
{
  static const k2_t t[] = {k_fwd_rows<I, -1>...};
  static const k2_t f[] = {k_fwd_rows<I, full_logt(I)>...};
  return full ? f[logf] : t[logf];
}
k2_t fb_pick2(int logf, bool full) { return pick2(logf, full, seq_t()); }
#endif
#if FB_HAS(3)
template <int... I> static k3_t pick3(int logf, bool full, iseq<I...>)
{
  static const k3_t t[] = {k_inv_chan<I, false, -1>...};
  static const k3_t f[] = {k_inv_chan<I, false, full_logt(I)>...};
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
    default: break;
  }
}
#endif
#if FB_HAS(5)
template <int... I> static k3_t pick3f(int logf, bool full, iseq<I...>)
{
  static const k3_t t[] = {k_inv_chan<I, true, -1>...};
  static const k3_t f[] = {k_inv_chan<I, true, full_logt(I)>...};
  return full ? f[logf] : t[logf];
}
k3_t fb_pick3f(int logf, bool full) { return pick3f(logf, full, seq_t()); }
#endif
#if FB_HAS(4)
template <int... I> static k3a_t pick3a(int logf, bool blocked, bool real, bool full, iseq<I...>)
{
  static const k3a_t tn[] = {k_inv_a<I, false, false, false>...};
  static const k3a_t tb[] = {k_inv_a<I, true, false, false>...};
  static const k3a_t rn[] = {k_inv_a<I, false, true, false>...};
  static const k3a_t rb[] = {k_inv_a<I, true, true, false>...};
  static const k3a_t tnf[] = {k_inv_a<I, false, false, true>...};
  static const k3a_t tbf[] = {k_inv_a<I, true, false, true>...};
  static const k3a_t rnf[] = {k_inv_a<I, false, true, true>...};
  static const k3a_t rbf[] = {k_inv_a<I, true, true, true>...};
  if (full) return real ? (blocked ? rbf[logf] : rnf[logf]) : (blocked ? tbf[logf] : tnf[logf]);
  return real ? (blocked ? rb[logf] : rn[logf]) : (blocked ? tb[logf] : tn[logf]);
}
template <int... I> static k3b_t pick3b(int logf, bool foldb, bool full, iseq<I...>)
{
  static const k3b_t t[] = {k_inv_b<I, false, false>...};
  static const k3b_t f[] = {k_inv_b<I, true, false>...};
  static const k3b_t tf[] = {k_inv_b<I, false, true>...};
  static const k3b_t ff[] = {k_inv_b<I, true, true>...};
  // (FOLDB with a radix-2 / radix-4 remainder stage -- LOGF % 4 == 1, 2 -- spills 12-20 registers in the full-tile form and
  //  none in the generic one: those lengths keep the generic kernel)
  if (full && !(foldb && (logf % 4 == 1 || logf % 4 == 2))) return foldb ? ff[logf] : tf[logf];
  return foldb ? f[logf] : t[logf];
}
k3a_t fb_pick3a(int logf, bool blocked, bool real, bool full) { return pick3a(logf, blocked, real, full, seq_t()); }
k3b_t fb_pick3b(int logf, bool foldb, bool full) { return pick3b(logf, foldb, full, seq_t()); }
#endif
#if FB_HAS(6)
k1c_t fb_pick_col1(int variant) { return variant ? k_fwd_col1q<1> : k_fwd_col1<1>; }
k3_t fb_pick_rinv(int logm, bool fold)
{
  switch (logm) {
    case 9: return fold ? k_rows_inv<9, 4, true> : k_rows_inv<9, 4, false>;
    case 10: return fold ? k_rows_inv<10, 3, true> : k_rows_inv<10, 3, false>;
    case 11: return fold ? k_rows_inv<11, 2, true> : k_rows_inv<11, 2, false>;
    case 12: return fold ? k_rows_inv<12, 1, true> : k_rows_inv<12, 1, false>;
    default: return nullptr;
  }
}
void fb_launch_raw_cols(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0)
{
  hipLaunchKernelGGL(k_raw_cols, grid, dim3(256), 0, stream, g, in, Rt, part0);
}
#endif
#endif
#if FB_HAS(1)
void fb_launch_raw_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, uint16_t* Rt, uint64_t part0)
{
  hipLaunchKernelGGL(k_raw_transpose, grid, dim3(256), 0, stream, g, in, Rt, part0);
}
void fb_launch_float_transpose(dim3 grid, hipStream_t stream, const FbGeom& g, const FbIn& in, cf* Rt, uint64_t part0)
{
  hipLaunchKernelGGL(k_float_transpose, grid, dim3(256), 0, stream, g, in, Rt, part0);
}
void fb_launch_sub_split(hipStream_t stream, const SubSplit& p, uint8_t* out, uint32_t ncu)
{
  hipLaunchKernelGGL(k_sub_split, dim3(8 * ncu), dim3(256), 0, stream, p, out);
}
void fb_launch_sub_combine(hipStream_t stream, const FbGeom& g, cf* X, uint32_t nseqs, uint32_t ncu, cf* Xout, uint32_t mo, uint32_t rm)
{
#define FB_SUBC(R)                                                                                                              \
  case R:                                                                                                                       \
    if (!Xout) hipLaunchKernelGGL((k_sub_combine<R, false>), dim3(8 * ncu), dim3(256), 0, stream, g, X, nseqs, X, 0u, 1u);        \
    else hipLaunchKernelGGL((k_sub_combine<R, true>), dim3(8 * ncu), dim3(256), 0, stream, g, X, nseqs, Xout, mo, rm);          \
    break;
  switch (g.nsub) { FB_SUBC(3) FB_SUBC(5) FB_SUBC(7) FB_SUBC(9) FB_SUBC(15) default: break; }
#undef FB_SUBC
}
#endif

#if FB_HAS(0)
constexpr int LOG_POINTS_DEFAULT = 14;  // points per workgroup (32 per thread, 512 threads)

static inline int ilog2(uint64_t v) { int l = 0; while ((1ull << l) < v) l++; return l; }
static inline bool ispow2(uint64_t v) { return v && !(v & (v - 1)); }

struct dspsr_amd_filterbank_impl {
  dspsr_amd_ctx* ctx;
  dspsr_amd_filterbank_config cfg;
  FbGeom g;
  uint64_t N, L;
  uint32_t nseq, max_parts;
  uint32_t nt1, nt2, nt3, nt4 = 0, ncu, wg_per_cu, wg3 = 1, wg1 = 1;
  size_t lds1, lds2, lds3, lds4 = 0;
  uint64_t part_elems = 0;    // scratch elements per part
  cf* A = nullptr;
  cf* X = nullptr;
  cf* kernel = nullptr;
  cf* tw_lo = nullptr;
  cf* tw_lo_m = nullptr;
  uint16_t* Rt = nullptr;   // pre-transposed 8-bit pairs of the parts of one launch group
  PlanSlot* plan_wait = nullptr;             // fold plan on its way to the device: the first kernel that reads it waits (fold_plan_wait)
  k1_t k1d_w1 = nullptr, k1d_w4 = nullptr;   // pass 1 on pairs of tiles (64-byte A runs otherwise), see k_fwd_cols_dual
  float* det = nullptr;     // detected block of perform_fold when the fused kernel would not fill the chip
  size_t det_floats = 0;
  bool kernel_set = false;
  // kernels of this geometry, chosen and given their dynamic-LDS limit once, at create time
  k1_t k1_w1 = nullptr, k1_w4 = nullptr, k1_w2 = nullptr;   // pass 1: one word per sample pair / generic loads / direct 8-bit (experiment)
  k2_t k2 = nullptr;
  k3_t k3 = nullptr, k3f = nullptr;                          // inverse pass: plain, fused fold
  k3a_t k3a = nullptr;
  k3b_t k3b = nullptr;
  float* fpart = nullptr;    // segmented fused fold: partial profiles of the part runs 1 .. nseg-1 of a launch
  size_t fpart_floats = 0;
  uint32_t plan_cap = 0;     // fused fold: plan entries per LDS buffer behind the twiddle tables
  size_t lds3f = 0;          // dynamic LDS of the fused inverse pass
  // two-pass path of short responses (complex dual-pol 8-bit input, nchan_subband * freq_res^2 == 2^27): see FB_HAS(6)
  uint8_t* dsub = nullptr;    // nsub > 1: the launch group's samples de-interleaved into nsub blocks (k_sub_split)
  size_t dsub_bytes = 0;
  bool two_pass = false;
  FbGeom g1t;                 // ... pass 1 of Fa < 2^14 through k_raw_transpose + k_fwd_cols: their geometry (M = Fa, Rr = Fb, T2 = freq_res)
  k1_t k1t = nullptr;
  uint32_t nt1t = 0;
  size_t lds1t = 0;
  k1c_t k1c = nullptr;
  k3_t k2r = nullptr, k2rf = nullptr;
  size_t lds1c = 0, lds2r = 0, lds2rf = 0;
  uint32_t plan_cap2 = 0;
  k3b_t k3bf = nullptr;      // four-pass fused fold: second inverse pass that leaves segment sums (FbOut kind 4)
  float* msum = nullptr;     // ... [chan][part][tile][t2][2] float4 of one input channel's sub-band and one block
  size_t msum_floats = 0;
  // freq_res = 3 * 2^k / 5 * 2^k (msub = 3, 5; see k_time_combine): g and everything above describe the INNER filterbank of
  // nchan_subband * msub pseudo-channels with freq_res / msub bins and the whole transform kept; cfg and these the caller's
  uint32_t msub = 0, out_C = 0, out_M = 0, out_nfilt_pos = 0, out_nkeep = 0;
  cf* Xp = nullptr;          // the combined spectrum in pseudo-channel order (k_sub_combine writes it there)
  cf* Y = nullptr;           // the pseudo-channels' time series of one launch group [pseudo-channel][pol][part][freq_res / msub]
  size_t Xp_elems = 0, Y_elems = 0;
};

}  // namespace dspsr_amd

using namespace dspsr_amd;

struct dspsr_amd_filterbank : dspsr_amd_filterbank_impl {};

static int fb_fail(dspsr_amd_ctx* ctx, int code, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  ctx_set_error_v(ctx, fmt, ap);
  va_end(ap);
  return code;
}

template <typename K> static hipError_t allow_lds(K kern, size_t bytes) { return dspsr_amd_allow_lds((const void*)kern, bytes); }

extern "C" int dspsr_amd_filterbank_create(dspsr_amd_ctx* ctx, const dspsr_amd_filterbank_config* cfg,
                                           dspsr_amd_filterbank** out)
{
  if (!ctx || !cfg || !out) return DSPSR_AMD_EINVAL;
  *out = nullptr;
  if (cfg->npol != 1 && cfg->npol != 2)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: npol=%u not 1 or 2", cfg->npol);
  // freq_res: a power of two, or 3 or 5 times one with a power-of-two nchan_subband (dspsr -x 12288).  The transforms inside a
  // tile stay powers of two: bins m = R m' + r of a channel are R pseudo-channels of freq_res / R bins (the inner filterbank of
  // nchan_subband * R channels below, whole transforms kept), whose time series k_time_combine adds with the twiddles
  // exp(+2 pi i r n / freq_res) -- the decimation-in-frequency form of the freq_res-point backward transform.
  // (odd factors 3, 5, 7, 9, 15 of either length; both lengths at once as long as the product of the two factors is one of those)
  auto odd_part = [](uint32_t v) { while (v && !(v & 1)) v >>= 1; return v; };
  auto radix_ok = [](uint32_t r) { return r == 3 || r == 5 || r == 7 || r == 9 || r == 15; };
  uint32_t msub = 0;
  if (!ispow2(cfg->freq_res)) {
    msub = odd_part(cfg->freq_res);
    if (!radix_ok(msub) || cfg->nchan_subband == 0 || !(ispow2(cfg->nchan_subband) || radix_ok(odd_part(cfg->nchan_subband) * msub)) ||
        cfg->freq_res / msub < 2 || cfg->force_four_pass == 1)
      return fb_fail(ctx, DSPSR_AMD_EINVAL,
                     "dspsr_amd_filterbank_create: freq_res=%u must be 2^k >= 2, or 2^k (k >= 1) times 3, 5, 7, 9 or 15 (times the odd "
                     "factor of nchan_subband=%u: again one of those; freq_res=1 is the non-convolving filterbank, not built yet)",
                     cfg->freq_res, cfg->nchan_subband);
  } else if (cfg->freq_res < 2)
    return fb_fail(ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_create: freq_res=%u must be >= 2 "
                   "(freq_res=1 is the non-convolving filterbank, not built yet)", cfg->freq_res);
  if (cfg->nfilt_pos + cfg->nfilt_neg >= cfg->freq_res)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nfilt_pos+nfilt_neg=%u >= freq_res=%u",
                   cfg->nfilt_pos + cfg->nfilt_neg, cfg->freq_res);
  // (the geometry below is built from these: the caller's values, or the inner filterbank's)
  const uint32_t nchan_sb = msub ? cfg->nchan_subband * msub : cfg->nchan_subband, fres = msub ? cfg->freq_res / msub : cfg->freq_res,
                 nfpos = msub ? 0u : cfg->nfilt_pos, nfneg = msub ? 0u : cfg->nfilt_neg;
  // nchan_subband: a power of two, or 3 or 5 times one (the forward transform then runs as 3 / 5 interleaved sub-sequences,
  // k_sub_split / k_sub_combine).
  uint32_t nsub = 1;
  if (!ispow2(nchan_sb)) {
    nsub = nchan_sb ? odd_part(nchan_sb) : 0;
    if (!radix_ok(nsub))
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nchan_subband=%u must be 2^k or 2^k times 3, 5, 7, 9 or 15",
                     cfg->nchan_subband);
    if (cfg->force_four_pass == 1)
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nchan_subband=%u (not a power of two) has no four-pass form",
                     cfg->nchan_subband);
  }
  if (cfg->input_nchan == 0) return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: input_nchan=0");

  dspsr_amd_filterbank* fb = new dspsr_amd_filterbank;
  fb->ctx = ctx;
  fb->cfg = *cfg;
  FbGeom& g = fb->g;
  // (C, Rr, logL, logC describe the power-of-two geometry passes 0-2 run on: one of nsub sub-sequences; fb->N, fb->L and g.C
  //  are the whole transform's)
  const uint64_t M = fres, C = nchan_sb / nsub;
  fb->msub = msub;
  fb->out_C = cfg->nchan_subband; fb->out_M = cfg->freq_res; fb->out_nfilt_pos = cfg->nfilt_pos;
  fb->out_nkeep = cfg->freq_res - cfg->nfilt_pos - cfg->nfilt_neg;
  fb->N = (uint64_t)nchan_sb * M;
  fb->L = cfg->real_input ? 2 * fb->N : fb->N;
  const uint64_t Rr = fb->L / nsub / M;
  const int logMf = ilog2(M), logL = ilog2(fb->L / nsub), logC = ilog2(C);
  g.logM = logMf;
  g.logR = ilog2(Rr);
  g.logMf = logMf;
  g.four_pass = 0;
  g.xblocked = 0;
  g.xblock = g.kblock = 0;
  g.xstride = fb->L;
  g.logMa = g.logMb = g.logTm = g.logTt = 0;
  g.logFb2 = g.logFa2 = 0;
  g.nsub = 1;
  g.tw_lo = g.tw_lo_m = nullptr;
  g.real_input = cfg->real_input ? 1 : 0;
  g.npol = cfg->npol;
  g.C = nchan_sb;
  g.nsub = nsub;
  g.nfilt_pos = nfpos;
  g.nkeep = fres - nfpos - nfneg;
  g.dbg = FB_ENV_INT("DSPSR_AMD_DEBUG", 0);
  fb->nseq = cfg->real_input ? 1 : cfg->npol;
  // tiles: every workgroup holds min(2^14, available) points = 32 per thread
  const int LOG_POINTS = FB_ENV_INT("DSPSR_AMD_LOG_POINTS", LOG_POINTS_DEFAULT);
  fb->wg_per_cu = FB_ENV_INT("DSPSR_AMD_WG_PER_CU", 1);
  if (fb->wg_per_cu < 1) fb->wg_per_cu = 1;
  auto imin = [](int a, int b) { return a < b ? a : b; };
  const int logPol = 1;   // the inverse passes always carry (pol0, pol1) column pairs
  // three passes (freq_res and the spectrum rows each fit one workgroup tile) when possible ...
  uint64_t p1 = 0, p2 = 0, p3 = 0, p4 = 0;
  bool three_ok = g.logM <= MAX_LOGF && g.logR <= MAX_LOGF;
  const int LOG_POINTS1 = FB_ENV_INT("DSPSR_AMD_P1_LOG_POINTS", LOG_POINTS);
  if (three_ok) {
    g.logT1 = imin(g.logR, LOG_POINTS1 - g.logM);
    g.logT2 = imin(g.logM, LOG_POINTS - g.logR);
    int t3 = LOG_POINTS - g.logM - logPol;
    if (t3 < 0) t3 = 0;
    g.logX3 = imin(logC, t3);                    // X layout: keeps the pass-2 store runs at T2*X3 elements
    if (FB_ENV_SET("DSPSR_AMD_LOG_X3")) g.logX3 = imin(logC, FB_ENV_INT("DSPSR_AMD_LOG_X3", 0) > t3 ? FB_ENV_INT("DSPSR_AMD_LOG_X3", 0) : t3);   // experiment: longer pass-2 runs
    // pass-3 tile: may be smaller than a layout block (DSPSR_AMD_P3_LOG_POINTS), two workgroups then share a CU
    const int LOG_POINTS3 = FB_ENV_INT("DSPSR_AMD_P3_LOG_POINTS", LOG_POINTS);
    int t3t = LOG_POINTS3 - g.logM - logPol;
    if (t3t < 0) t3t = 0;
    g.logT3 = imin(g.logX3, t3t);
    p1 = M << g.logT1; p2 = Rr << g.logT2; p3 = (M << g.logT3) << logPol;
    three_ok = !(p1 < 32 || p2 < 32 || p3 < 32 || p3 > (1u << LOG_POINTS) || g.logT1 < 1 || g.logT2 < 1);
  }
  // ... otherwise four: L = Fa*Fb forward (whole spectrum, blocked by pass-2 tile), freq_res = Ma*Mb inverse in two passes.
  // This also covers nchan_subband = 1 (dsp::Convolution) and freq_res up to 2^26.
  if (nsub > 1 && !three_ok) {
    delete fb;
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nchan_subband=%u (not a power of two) needs freq_res <= 8192 "
                   "and a sub-geometry of at least 32 points per pass", cfg->nchan_subband);
  }
  if (cfg->force_four_pass == 1 || !three_ok) {
    int la = (logL + 1) / 2;
    if (la > MAX_LOGF) la = MAX_LOGF;
    const int lb = logL - la;
    // spectrum layout between pass 2 and the inverse: blocked (every pass-2 tile one contiguous block) when the natural
    // order would leave pass 2 with runs of fewer than 16 elements (128 bytes); measured per geometry, blocked is then
    // 15-40 % faster over the whole launch group, natural 3 % faster otherwise (profiles/r02x_inverse_split.txt)
    const int logT2f = imin(la, LOG_POINTS - (logL - la));
    const bool blocked = FB_ENV_SET("DSPSR_AMD_X_NATURAL") ? false : FB_ENV_SET("DSPSR_AMD_X_BLOCKED") ? true : logT2f < 4;
    // freq_res = Ma*Mb: the split that measured fastest (same file).  The second inverse pass likes Mb = 256 (two
    // radix-16 stages, 32 adjacent output samples per run), the first one Ma <= 2^11 (>= 4 columns per tile).
    static const signed char lma_best[27] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, -1, 10, 11, 11, 11, 12, 12, 13};
    int lma = (logMf + 1) / 2;
    if (logMf >= 14 && logMf <= 26) lma = lma_best[logMf] > 0 ? lma_best[logMf] : (blocked ? 11 : 9);
    lma = FB_ENV_INT("DSPSR_AMD_LMA", lma);
    if (lma > MAX_LOGF) lma = MAX_LOGF;
    const int lmb = logMf - lma;
    g.logM = la; g.logR = lb; g.logT3 = g.logX3 = 0;
    g.logT1 = imin(lb, LOG_POINTS - la);
    g.logT2 = imin(la, LOG_POINTS - lb);
    g.logMa = lma; g.logMb = lmb;
    g.logTm = imin(lmb, LOG_POINTS - logPol - lma);
    g.logTt = imin(lma, LOG_POINTS - logPol - lmb);
    p1 = (1ull << la) << g.logT1; p2 = (1ull << lb) << g.logT2;
    p3 = ((1ull << lma) << g.logTm) << logPol; p4 = ((1ull << lmb) << g.logTt) << logPol;
    const bool ok = lb >= 1 && lb <= MAX_LOGF && lmb >= 1 && lmb <= MAX_LOGF && g.logT1 >= 1 && g.logT2 >= 1 &&
                    g.logTm >= 0 && g.logTt >= 0 && p1 >= 32 && p2 >= 32 && p3 >= 32 && p4 >= 32;
    if (!ok) {
      delete fb;
      return fb_fail(ctx, DSPSR_AMD_EINVAL,
                     "dspsr_amd_filterbank_create: nchan_subband=%llu freq_res=%llu cannot be tiled "
                     "(forward 2^%d x 2^%d, inverse 2^%d x 2^%d; every pass needs 32..16384 points per workgroup "
                     "and factors <= 2^%d)", (unsigned long long)C, (unsigned long long)M, la, lb, lma, lmb, MAX_LOGF);
    }
    g.four_pass = 1;
    // (k_inv_a's address arithmetic assumes that a thread's 16 elements differ in bits of k above the low T2 ones)
    g.xblocked = (blocked && lmb >= g.logT2 && (p3 / PTS) >= (1u << g.logT2)) ? 1 : 0;
    if (g.xblocked) {
      g.xblock = (1u << (lb + g.logT2)) + (uint32_t)FB_ENV_INT("DSPSR_AMD_XPAD", 0);     // padding: even (16-byte stores)
      g.xstride = (uint64_t)g.xblock << (la - g.logT2);
      g.kblock = (uint32_t)((fb->N >> la) << g.logT2);
    }
  }
  fb->nt1 = (uint32_t)(p1 / PTS);
  fb->nt2 = (uint32_t)(p2 / PTS);
  fb->nt3 = (uint32_t)(p3 / PTS);
  fb->nt4 = (uint32_t)(p4 / PTS);
  fb->ncu = ctx->ncu;
  fb->lds1 = lds_total_words_host((uint32_t)p1, g.logM) * sizeof(cf);
  fb->lds2 = lds_total_words_host((uint32_t)p2, g.logR) * sizeof(cf);
  fb->lds3 = lds_total_words_host((uint32_t)p3, g.four_pass ? g.logMa : g.logM) * sizeof(cf);
  fb->lds4 = g.four_pass ? lds_total_words_host((uint32_t)p4, g.logMb) * sizeof(cf) : 0;
  fb->wg3 = (!g.four_pass && 2 * fb->lds3 + 1024 <= 160 * 1024) ? 2 * fb->wg_per_cu : fb->wg_per_cu;
  fb->wg1 = (2 * fb->lds1 + 1024 <= 160 * 1024 && fb->nt1 <= 256) ? 2 * fb->wg_per_cu : fb->wg_per_cu;
  {
    // kernels of this geometry and their dynamic-LDS limits (once; perform only launches)
    const bool notfixed = FB_ENV_SET("DSPSR_AMD_RUNTIME_LOGT");    // experiments: force the generic kernels
    const bool full1 = !notfixed && g.logT1 == full_logt(g.logM), full2 = !notfixed && g.logT2 == full_logt(g.logR),
               full3 = !notfixed && !g.four_pass && g.logT3 + 1 == full_logt(g.logM);
    fb->k1_w1 = fb_pick1(g.logM, 1, full1);
    fb->k1_w4 = fb_pick1(g.logM, 4, full1);
#if defined(FB_ONLY_HEADLINE) && defined(DSPSR_AMD_EXPERIMENT)
    fb->k1_w2 = fb_pick1(g.logM, 2, full1);
#endif
    // two-column tiles of 2^13 rows whose A runs would be half cache lines: transformed in pairs (k_fwd_cols_dual)
    if (full1 && g.logM == 13 && g.logT1 == 1 && g.logT1 + g.logT2 < 4 && g.logR >= 2 && !FB_ENV_SET("DSPSR_AMD_NO_DUAL")) {
      fb->k1d_w1 = fb_pick1_dual(1);
      fb->k1d_w4 = fb_pick1_dual(4);
    }
    fb->k2 = fb_pick2(g.logR, full2);
    if (g.four_pass) {
      fb->k3a = fb_pick3a(g.logMa, g.xblocked != 0, g.real_input != 0, !notfixed && g.logTm == 13 - g.logMa && g.logMa <= 12 && fb->nt3 == 512);
      const bool full4 = !notfixed && g.logTt == 13 - g.logMb && g.logMb <= 12 && fb->nt4 == 512;
      fb->k3b = fb_pick3b(g.logMb, false, full4);
      fb->k3bf = fb_pick3b(g.logMb, true, full4);
    } else {
      fb->k3 = fb_pick3(g.logM, full3);
      fb->k3f = fb_pick3f(g.logM, full3);
      // fused fold: the LDS left over behind the twiddle tables holds the part's fold plan (two buffers)
      const size_t psl_bytes = FB_PSL_MAX * sizeof(uint32_t);
      const size_t spare = 160 * 1024 - 64 - fb->lds3 - 16 - psl_bytes;
      uint32_t cap = fb->lds3 + 64 + 16 + psl_bytes < 160 * 1024 ? (uint32_t)(spare / 32) : 0;
      if (cap > fb->nt3) cap = fb->nt3;           // one plan entry per thread of the workgroup (nt3 <= 512)
      if (cap < 16) cap = 0;
      fb->plan_cap = cap;
      fb->lds3f = fb->lds3 + 16 + (size_t)cap * 32 + psl_bytes;
    }
    hipError_t e = hipSuccess;
    bool have = (fb->k1_w1 || fb->k1_w4) && fb->k2 && (g.four_pass ? (fb->k3a && fb->k3b) : (fb->k3 != nullptr));
    if (have) {
      if (fb->k1_w1) e = allow_lds(fb->k1_w1, fb->lds1);
      if (e == hipSuccess && fb->k1_w4) e = allow_lds(fb->k1_w4, fb->lds1);
      if (e == hipSuccess && fb->k1_w2) e = allow_lds(fb->k1_w2, fb->lds1);
      if (e == hipSuccess && fb->k1d_w1) e = allow_lds(fb->k1d_w1, fb->lds1);
      if (e == hipSuccess && fb->k1d_w4) e = allow_lds(fb->k1d_w4, fb->lds1);
      if (e == hipSuccess) e = allow_lds(fb->k2, fb->lds2);
      if (e == hipSuccess && fb->k3) e = allow_lds(fb->k3, fb->lds3);
      if (e == hipSuccess && fb->k3f) e = allow_lds(fb->k3f, fb->lds3f);
      if (e == hipSuccess && fb->k3a) e = allow_lds(fb->k3a, fb->lds3);
      if (e == hipSuccess && fb->k3b) e = allow_lds(fb->k3b, fb->lds4);
      if (e == hipSuccess && fb->k3bf) e = allow_lds(fb->k3bf, fb->lds4);
    }
    if (!have || e != hipSuccess) {
      delete fb;
      return have ? fb_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_create: hipFuncSetAttribute: %s", hipGetErrorString(e))
                  : fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: geometry not in this (experiment) build");
    }
  }
  // Two-pass path: forward and inverse levels together fit two workgroup tiles (FB_HAS(6)).  Complex dual-pol input,
  // 512 <= freq_res <= 4096, Fb = 2^13 / freq_res channels per inverse tile, Fa = L / Fb with freq_res <= Fa <= 2^14, i.e.
  // Fb <= nchan_subband <= 2^27 / freq_res^2 (the 50 MHz sub-band geometry -F 512:D -x 512 is the upper end).  Taken per call
  // when the input is the generic 8-bit block (fb_run); the three-pass kernels above serve every other input form.
  // force_four_pass == 2 switches it off (comparison runs and tests).
  {
    const int lfb = 13 - logMf, lfa = logL - lfb;
    if (nsub == 1 && !g.four_pass && !cfg->real_input && cfg->npol == 2 && cfg->force_four_pass != 2 && logMf >= 9 && logMf <= 12 &&
        lfa >= logMf && lfa <= 14 && ctx->ncu > 0 && FB_ENV_INT("DSPSR_AMD_NO_TWO_PASS", 0) == 0) {
      hipError_t e2 = hipSuccess;
      bool have1 = false;
      if (lfa == 14) {
        const int c1v = FB_ENV_INT("DSPSR_AMD_COL1_V", 1);
        fb->k1c = fb_pick_col1(c1v);
        fb->lds1c = lds_total_words_host(1u << 14, c1v ? 12 : 13) * sizeof(cf);
        have1 = fb->k1c != nullptr;
        if (have1) e2 = allow_lds(fb->k1c, fb->lds1c);
      } else {
        // pass 1 = the ordinary column pass on a geometry of its own: T1 adjacent columns nb per tile, A blocked by T2 = freq_res
        FbGeom& q = fb->g1t;
        q = g;
        q.logM = lfa; q.logR = lfb;
        q.logT1 = imin(lfb, LOG_POINTS_DEFAULT - lfa);
        q.logT2 = logMf;
        const uint64_t p1t = (1ull << lfa) << q.logT1;
        fb->nt1t = (uint32_t)(p1t / PTS);
        fb->lds1t = lds_total_words_host((uint32_t)p1t, lfa) * sizeof(cf);
        fb->k1t = fb_pick1(lfa, 1, q.logT1 == full_logt(lfa));
        have1 = fb->k1t != nullptr && q.logT1 >= 1 && p1t >= 32;
        if (have1) e2 = allow_lds(fb->k1t, fb->lds1t);
      }
      fb->k2r = fb_pick_rinv(logMf, false);
      fb->k2rf = fb_pick_rinv(logMf, true);
      if (have1 && fb->k2r && fb->k2rf) {
        g.logFb2 = lfb;
        g.logFa2 = lfa;
        fb->g1t.logFb2 = lfb; fb->g1t.logFa2 = lfa;
        fb->lds2r = lds_total_words_host(1u << 14, logMf) * sizeof(cf);
        const size_t psl_bytes = FB_PSL_MAX * sizeof(uint32_t);
        const size_t spare = 160 * 1024 - 64 - fb->lds2r - 16 - psl_bytes;
        uint32_t cap = fb->lds2r + 64 + 16 + psl_bytes < 160 * 1024 ? (uint32_t)(spare / 32) : 0;
        if (cap > 512) cap = 512;
        if (cap < 16) cap = 0;
        fb->plan_cap2 = cap;
        fb->lds2rf = fb->lds2r + 16 + (size_t)cap * 32 + psl_bytes;
        if (e2 == hipSuccess) e2 = allow_lds(fb->k2r, fb->lds2r);
        if (e2 == hipSuccess) e2 = allow_lds(fb->k2rf, fb->lds2rf);
        fb->two_pass = e2 == hipSuccess;
      }
    }
  }
  fb->max_parts = cfg->max_parts ? cfg->max_parts : 1;
  // per part: nseq sequences of L points; the two-pass inverse re-uses A for 2 polarisations x N bins
  fb->part_elems = fb->nseq * g.xstride;       // (A needs nseq*L; X the same or, blocked and padded, a little more)
  if (g.four_pass && fb->part_elems < 2 * fb->N) fb->part_elems = 2 * fb->N;
  const size_t scratch = (size_t)fb->max_parts * fb->part_elems * sizeof(cf);
  if (hipMalloc((void**)&fb->A, scratch) != hipSuccess || hipMalloc((void**)&fb->X, scratch) != hipSuccess) {
    if (fb->A) (void)hipFree(fb->A);
    delete fb;
    return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_create: hipMalloc of 2 x %zu scratch bytes failed",
                   scratch);
  }
  if (g.four_pass && g.logMf > LOG_TWN) {     // fine twiddle table of the two-pass inverse, built in double
    const int sh = g.logMf - LOG_TWN;
    std::vector<cf> lo(1u << sh);
    for (uint32_t j = 0; j < (1u << sh); j++) {
      const double a = -2.0 * M_PI * (double)j / (double)M;
      lo[j] = make_float2((float)cos(a), (float)sin(a));
    }
    if (hipMalloc((void**)&fb->tw_lo_m, lo.size() * sizeof(cf)) != hipSuccess ||
        hipMemcpy(fb->tw_lo_m, lo.data(), lo.size() * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess) {
      dspsr_amd_filterbank_destroy(fb);
      return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_create: twiddle table allocation failed");
    }
    g.tw_lo_m = fb->tw_lo_m;
  }
  if (g.logM + g.logR > LOG_TWN) {            // fine twiddle table of pass 1, built in double
    const int sh = g.logM + g.logR - LOG_TWN;
    std::vector<cf> lo(1u << sh);
    for (uint32_t j = 0; j < (1u << sh); j++) {
      const double a = -2.0 * M_PI * (double)j / (double)fb->L;
      lo[j] = make_float2((float)cos(a), (float)sin(a));
    }
    if (hipMalloc((void**)&fb->tw_lo, lo.size() * sizeof(cf)) != hipSuccess ||
        hipMemcpy(fb->tw_lo, lo.data(), lo.size() * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess) {
      dspsr_amd_filterbank_destroy(fb);
      return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_create: twiddle table allocation failed");
    }
    g.tw_lo = fb->tw_lo;
  }
  *out = fb;
  return DSPSR_AMD_OK;
}

extern "C" void dspsr_amd_filterbank_destroy(dspsr_amd_filterbank* fb)
{
  if (!fb) return;
  (void)hipStreamSynchronize(fb->ctx->stream);
  if (fb->A) (void)hipFree(fb->A);
  if (fb->X) (void)hipFree(fb->X);
  if (fb->kernel) (void)hipFree(fb->kernel);
  if (fb->Rt) (void)hipFree(fb->Rt);
  if (fb->det) (void)hipFree(fb->det);
  if (fb->fpart) (void)hipFree(fb->fpart);
  if (fb->msum) (void)hipFree(fb->msum);
  if (fb->dsub) (void)hipFree(fb->dsub);
  if (fb->Xp) (void)hipFree(fb->Xp);
  if (fb->Y) (void)hipFree(fb->Y);
  if (fb->tw_lo) (void)hipFree(fb->tw_lo);
  if (fb->tw_lo_m) (void)hipFree(fb->tw_lo_m);
  delete fb;
}

extern "C" int dspsr_amd_filterbank_set_kernel(dspsr_amd_filterbank* fb, const float* kernel_host, uint64_t ncomplex)
{
  if (!fb) return DSPSR_AMD_EINVAL;
  if (!kernel_host) {  // no response: plain filterbank
    if (fb->kernel) (void)hipFree(fb->kernel);
    fb->kernel = nullptr;
    fb->kernel_set = true;
    return DSPSR_AMD_OK;
  }
  const uint64_t expect = (uint64_t)fb->cfg.input_nchan * fb->N;
  if (ncomplex != expect)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_set_kernel: kernel has %llu bins, expected %llu",
                   (unsigned long long)ncomplex, (unsigned long long)expect);
  if (!fb->kernel && hipMalloc((void**)&fb->kernel, expect * sizeof(cf)) != hipSuccess)
    return fb_fail(fb->ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_set_kernel: hipMalloc failed");
  const cf* src = (const cf*)kernel_host;
  std::vector<cf> perm;
  if (fb->g.xblocked) {
    // four-pass geometries: the chirp lies on the device in the order of the blocked spectrum (k_inv_a loads both alike)
    const FbGeom& g = fb->g;
    const uint64_t N = fb->N, maskA = (1ull << g.logM) - 1, maskT = (1ull << g.logT2) - 1;
    perm.resize(expect);
    for (uint64_t ic = 0; ic < fb->cfg.input_nchan; ic++)
      for (uint64_t k = 0; k < N; k++) {
        const uint64_t ka = k & maskA, kb = k >> g.logM;
        perm[ic * N + (ka >> g.logT2) * g.kblock + ((kb << g.logT2) | (ka & maskT))] = src[ic * N + k];
      }
    src = perm.data();
  }
  if (fb->msub) {
    // the inner filterbank's channels are the pseudo-channels (c, r): bin m' of pseudo-channel c*R + r is bin R*m' + r of channel c
    const uint64_t N = fb->N, R = fb->msub, Mo = fb->out_M, Mi = Mo / R;
    perm.resize(expect);
    for (uint64_t ic = 0; ic < fb->cfg.input_nchan; ic++)
      for (uint64_t k = 0; k < N; k++) {
        const uint64_t c = k / Mo, m = k - c * Mo, mi = m / R, r = m - mi * R;
        perm[ic * N + (c * R + r) * Mi + mi] = src[ic * N + k];
      }
    src = perm.data();
  }
  hipError_t e = hipMemcpyAsync(fb->kernel, src, expect * sizeof(cf), hipMemcpyHostToDevice, fb->ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(fb->ctx->stream);
  if (e != hipSuccess)
    return fb_fail(fb->ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_set_kernel: %s", hipGetErrorString(e));
  fb->kernel_set = true;
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_filterbank_sizes(const dspsr_amd_filterbank* fb, uint64_t* nsamp_fft,
                                          uint64_t* nsamp_overlap, uint64_t* nsamp_step, uint32_t* nkeep)
{
  if (!fb) return DSPSR_AMD_EINVAL;
  const uint64_t nfilt_tot = fb->cfg.nfilt_pos + fb->cfg.nfilt_neg;
  const uint64_t fft = fb->cfg.real_input ? 2 * fb->N : fb->N;                          // Filterbank.C:139-148
  const uint64_t ovl = (fb->cfg.real_input ? 2 : 1) * nfilt_tot * fb->cfg.nchan_subband;
  if (nsamp_fft) *nsamp_fft = fft;
  if (nsamp_overlap) *nsamp_overlap = ovl;
  if (nsamp_step) *nsamp_step = fft - ovl;
  if (nkeep) *nkeep = fb->msub ? fb->out_nkeep : fb->g.nkeep;
  return DSPSR_AMD_OK;
}

static int raw_kind(int raw_layout) { return raw_layout == DSPSR_AMD_RAW_CASPSR ? 2 : raw_layout == DSPSR_AMD_RAW_UWB16 ? 4 : 1; }

static uint32_t grid_for(uint64_t items, uint32_t ncu)
{
  uint64_t gsz = items < ncu ? items : ncu;
  if (gsz >= 8) gsz &= ~7ull;
  return (uint32_t)gsz;
}


// Fused inverse pass of one launch (ns parts starting at part0).  With fewer channel tiles than compute units the parts
// are cut into runs folded by different workgroups (k_inv_chan, "Segmented"): partial profiles zeroed before, added to the
// profile in run order after the launch.  segmented == false: one workgroup owns a tile for all parts (exact time order).
static int fb_launch_fused(dspsr_amd_filterbank* fb, k3_t k3, const cf* X, const cf* kern, FbOut co, uint64_t part0, uint32_t ns,
                           bool segmented, bool two_pass = false)
{
  dspsr_amd_ctx* ctx = fb->ctx;
  const FbGeom& g = fb->g;
  // (two-pass path: the tile is Fb = 2^logFb2 channels, one 512-thread workgroup per compute unit)
  const uint32_t tiles = two_pass ? g.C >> g.logFb2 : g.C >> g.logT3, wgs = two_pass ? fb->ncu : fb->ncu * fb->wg3;
  uint32_t nseg = 1;
  if (segmented && tiles < wgs) {
    nseg = wgs / tiles;
    if (nseg > ns) nseg = ns;
    if (nseg > 16) nseg = 16;
    if (nseg < 1) nseg = 1;
  }
  if (fb->plan_wait) {
    const int rc = fold_plan_wait(co.fold, fb->plan_wait);
    fb->plan_wait = nullptr;
    if (rc != DSPSR_AMD_OK) return rc;
  }
  co.plan_cap = two_pass ? fb->plan_cap2 : fb->plan_cap;
  co.nseg = nseg;
  co.part = nullptr;
  if (nseg > 1) {
    const size_t need = (size_t)(nseg - 1) * g.C * co.nbin * 4;
    if (need > fb->fpart_floats) {
      (void)hipStreamSynchronize(ctx->stream);
      if (fb->fpart) (void)hipFree(fb->fpart);
      fb->fpart = nullptr; fb->fpart_floats = 0;
      if (hipMalloc((void**)&fb->fpart, need * sizeof(float)) != hipSuccess)
        return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform_fold: hipMalloc of %zu partial-profile bytes failed",
                       need * sizeof(float));
      fb->fpart_floats = need;
    }
    if (hipMemsetAsync(fb->fpart, 0, need * sizeof(float), ctx->stream) != hipSuccess)
      return fb_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_perform_fold: hipMemsetAsync failed");
    co.part = fb->fpart;
  }
  const uint32_t grid = nseg > 1 ? tiles * nseg : grid_for(tiles, wgs);
  hipLaunchKernelGGL(k3, dim3(grid), dim3(two_pass ? 512u : fb->nt3), two_pass ? fb->lds2rf : fb->lds3f, ctx->stream, g, X, kern, co,
                     ctx->tw, part0, ns, ns);
  if (nseg > 1) return fold_combine_partials(co.fold, fb->fpart, nseg - 1, co.chan0, g.C);
  return DSPSR_AMD_OK;
}

// output channels per input channel / kept samples per part as the caller sees them (freq_res = 3 * 2^k / 5 * 2^k: g describes the
// inner filterbank of pseudo-channels)
static inline uint32_t fb_out_C(const dspsr_amd_filterbank* fb) { return fb->msub ? fb->out_C : fb->g.C; }
static inline uint32_t fb_out_nkeep(const dspsr_amd_filterbank* fb) { return fb->msub ? fb->out_nkeep : fb->g.nkeep; }

static int fb_run(dspsr_amd_filterbank* fb, FbIn in, FbOut out, uint64_t npart, uint64_t in_chan_stride_bytes_or_floats)
{
  dspsr_amd_ctx* ctx = fb->ctx;
  if (!fb->kernel_set)
    return fb_fail(ctx, DSPSR_AMD_ESTATE, "dspsr_amd_filterbank_perform: set_kernel (Engine::setup) not called");
  if (npart == 0) return DSPSR_AMD_OK;
  const FbGeom& g = fb->g;
  // 8-bit real dual-pol single-channel input: one 32-bit word per sample pair; regroup it per tile first
  // (k_raw_transpose) unless the rows are already long enough or the layout preconditions fail
  const bool fast8 = (in.kind == 1 || in.kind == 2) && g.real_input && g.npol == 2 && fb->cfg.input_nchan == 1 &&
                     ((uintptr_t)in.base % 4) == 0;
  // complex dual-pol generic order: the same regroup per polarisation; pass 1 then reads (re, im) byte pairs exactly
  // like the (pol0, pol1) pairs of real input, one aligned word per two columns
  const bool fastc = in.kind == 1 && !g.real_input && g.npol == 2 && fb->cfg.input_nchan == 1 &&
                     ((uintptr_t)in.base % 16) == 0 && (in.part_step % 4) == 0 && g.logR >= 3;
  // the two-pass path of short responses takes exactly this input form (and out.kind 0..3; the four-pass segment sums never
  // apply: freq_res <= 4096)
  // (whole columns, Fa = 2^14: k_raw_cols also takes blocks of several input channels; Fa < 2^14 goes through k_raw_transpose)
  const bool two = fb->two_pass && in.kind == 1 && !g.real_input && g.npol == 2 && out.kind != 4 && (in.part_step % 4) == 0 &&
                   (fb->k1c ? ((uintptr_t)in.base % (fb->cfg.input_nchan == 1 ? 16 : 4)) == 0
                            : (fb->cfg.input_nchan == 1 && ((uintptr_t)in.base % 16) == 0));
  bool pret = two || ((fast8 || fastc) && g.logR >= 2 && g.logT1 <= 5 && !FB_ENV_SET("DSPSR_AMD_NO_PRETRANSPOSE"));   // rows of >= 128 B need no regrouping
  if (pret && in.kind == 2 && (in.part_step % 4) != 0) pret = false;
  if (pret && !fb->Rt) {
    if (hipMalloc((void**)&fb->Rt, (size_t)fb->max_parts * fb->nseq * fb->L * sizeof(uint16_t)) != hipSuccess)
      return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform: hipMalloc of the 8-bit regroup buffer failed");
  }
  // float32 rows (what Filterbank::Engine::perform is given): regrouped likewise, 8-byte elements, into the idle X scratch
  const bool pretf = in.kind == 0 && g.npol == 2 && g.logR >= 6 && g.logM >= 1 && g.logT1 >= 1 && g.logT1 <= 4 &&
                     ((uintptr_t)in.base % 16) == 0 && (in.part_step % 4) == 0 && (in.pol_stride % 4) == 0 &&
                     (in_chan_stride_bytes_or_floats % 4) == 0 && !FB_ENV_SET("DSPSR_AMD_NO_PRETRANSPOSE");
  int raww = (pret || (fast8 && in.kind == 1)) ? 1 : 4;
#ifdef FB_ONLY_HEADLINE
  if (fast8 && FB_ENV_SET("DSPSR_AMD_DIRECT8") && g.logT1 == 2 && (in.part_step % 4) == 0 && ((uintptr_t)in.base % 8) == 0) {
    pret = false;     // experiment: 4-column tiles read straight from the stream (no regroup pass)
    raww = 2;
  }
#endif
  k1_t k1 = raww == 1 ? fb->k1_w1 : raww == 2 ? fb->k1_w2 : fb->k1_w4;
  const k1_t k1d = raww == 1 ? fb->k1d_w1 : raww == 4 ? fb->k1d_w4 : nullptr;
  k2_t k2 = fb->k2;
  k3_t k3 = out.kind == 3 ? fb->k3f : fb->k3;
  k3a_t k3a = fb->k3a;
  k3b_t k3b = out.kind == 4 ? fb->k3bf : fb->k3b;
  if (!k1 || !k2 || (g.four_pass ? (!k3a || !k3b) : !k3))
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: geometry not in this (experiment) build");
  hipError_t e;
  const uint32_t Rr = 1u << g.logR, M = 1u << g.logM;
  const float* in_f32 = (const float*)in.base;
  const bool fused_segmented = out.kind == 3 && dspsr_amd_filterbank_fold_is_fused(fb) == 2;
  for (uint32_t ichan = 0; ichan < fb->cfg.input_nchan; ichan++) {
    FbIn ci = in;
    if (in.kind == 0) ci.base = in_f32 + ichan * in_chan_stride_bytes_or_floats;
    ci.ichan = ichan;
    ci.nchan = fb->cfg.input_nchan;
    FbOut co = out;
    co.chan0 = ichan * g.C;
    const cf* kern = fb->kernel ? fb->kernel + (uint64_t)ichan * fb->N : nullptr;
    uint32_t nb_step = 0;
    for (uint64_t part0 = 0; part0 < npart; part0 += nb_step) {
      uint32_t nb = (uint32_t)((npart - part0) < fb->max_parts ? (npart - part0) : fb->max_parts);
      {  // the kernels count their work items in 32 bits: keep every pass of a launch group below 2^31 items
        uint64_t per_part_items = (uint64_t)(Rr >> g.logT1) * fb->nseq;
        const uint64_t i2 = (uint64_t)(M >> g.logT2) * fb->nseq, i3 = g.four_pass ? 0 : (uint64_t)(g.C >> g.logT3),
                       i3a = g.four_pass ? ((uint64_t)g.C << (g.logMb - g.logTm)) : 0, i3b = g.four_pass ? ((uint64_t)g.C << (g.logMa - g.logTt)) : 0;
        if (i2 > per_part_items) per_part_items = i2;
        if (i3 > per_part_items) per_part_items = i3;
        if (i3a > per_part_items) per_part_items = i3a;
        if (i3b > per_part_items) per_part_items = i3b;
        while (nb > 1 && per_part_items * nb >= (1ull << 31)) nb /= 2;
        // (the sub-sequences of a launch group share one de-interleaved block: the parts must start a multiple of nsub samples
        //  apart -- always so for nchan_subband = nsub * 2^k, for freq_res = nsub * 2^k only when the kept length allows it)
        if (g.nsub > 1 && in.part_step % g.nsub) nb = 1;
        nb_step = nb;
      }
      if (g.nsub > 1) {
        // nchan_subband = 3 * 2^k / 5 * 2^k: the group's samples as nsub interleaved sub-sequences, passes 0-2 on each (the
        // power-of-two geometry), one radix-nsub step on the sub-spectra, then the inverse pass on nsub << logR rows
        const uint32_t R = g.nsub;
        if (in.kind == 4)
          return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: 16-bit UWB blocks need power-of-two nchan_subband and "
                         "freq_res (k_sub_split de-interleaves 8-bit and float32 input)");
        const uint32_t ndim = g.real_input ? 1u : 2u;
        const uint64_t step = in.part_step, nper = ((uint64_t)(nb - 1) * step + fb->L) / R;       // (step and L are multiples of nsub)
        const size_t es = in.kind == 0 ? (size_t)g.npol * ndim * sizeof(float) : (size_t)g.npol * ndim;   // bytes per sample, all pols
        const size_t sub_stride = (nper * es + 15) & ~(size_t)15;
        if (sub_stride * R > fb->dsub_bytes) {
          (void)hipStreamSynchronize(ctx->stream);
          if (fb->dsub) (void)hipFree(fb->dsub);
          fb->dsub = nullptr; fb->dsub_bytes = 0;
          if (hipMalloc((void**)&fb->dsub, sub_stride * R) != hipSuccess)
            return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform: hipMalloc of %zu sub-sequence bytes failed", sub_stride * R);
          fb->dsub_bytes = sub_stride * R;
        }
        SubSplit sp = {in.kind, in.base, in.kind == 0 ? (uint64_t)ichan * in_chan_stride_bytes_or_floats : 0, in.pol_stride,
                       fb->cfg.input_nchan, ichan, (uint32_t)g.npol, ndim, part0 * step, nper, R, sub_stride};
        fb_launch_sub_split(ctx->stream, sp, fb->dsub, fb->ncu);
        const uint64_t Ls = fb->L / R;
        for (uint32_t c = 0; c < R; c++) {
          FbIn cs = in;
          cs.kind = in.kind == 0 ? 0 : 1;                           // (the split writes the generic byte order)
          cs.base = fb->dsub + (size_t)c * sub_stride;
          cs.pol_stride = in.kind == 0 ? nper * ndim : 0;
          cs.part_step = step / R;
          cs.nchan = 1; cs.ichan = 0;
          const bool f8 = cs.kind == 1 && g.real_input && g.npol == 2;
          const bool fc = cs.kind == 1 && !g.real_input && g.npol == 2 && (cs.part_step % 4) == 0 && g.logR >= 3;
          const bool prt = (f8 || fc) && g.logR >= 2 && g.logT1 <= 5 && (cs.part_step % 4) == 0;
          const bool prf = cs.kind == 0 && g.npol == 2 && g.logR >= 6 && g.logT1 >= 1 && g.logT1 <= 4 && (cs.part_step % 4) == 0 &&
                           (cs.pol_stride % 4) == 0;
          if (prt && !fb->Rt && hipMalloc((void**)&fb->Rt, (size_t)fb->max_parts * fb->nseq * fb->L * sizeof(uint16_t)) != hipSuccess)
            return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform: hipMalloc of the 8-bit regroup buffer failed");
          const int rw = (prt || f8) ? 1 : 4;
          k1_t k1s = rw == 1 ? fb->k1_w1 : fb->k1_w4;
          if (!k1s) return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: geometry not in this (experiment) build");
          FbIn cr = cs;
          if (prt) {
            fb_launch_raw_transpose(dim3((Rr + 255) / 256, (M + 63) / 64, nb * fb->nseq), ctx->stream, g, cs, fb->Rt, 0);
            cr.kind = 3; cr.base = fb->Rt;
          } else if (prf) {
            // (the float regroup buffer is the X scratch in the power-of-two path; X holds finished sub-spectra here: use Rt's
            //  place in A's idle upper half -- A needs nseq * L' of its nseq * L elements per part)
            cf* ft = fb->A + (size_t)nb * fb->nseq * Ls;
            fb_launch_float_transpose(dim3((Rr + FB_FT_COLS - 1) / FB_FT_COLS, (M + FB_FT_ROWS - 1) / FB_FT_ROWS, nb * fb->nseq), ctx->stream, g, cs, ft, 0);
            cr.kind = 5; cr.base = ft;
          }
          const uint64_t n1s = (uint64_t)(Rr >> g.logT1) * fb->nseq * nb, n2s = (uint64_t)(M >> g.logT2) * fb->nseq * nb;
          hipLaunchKernelGGL(k1s, dim3(grid_for(n1s, fb->ncu * fb->wg1)), dim3(fb->nt1), fb->lds1, ctx->stream, g, cr, fb->A, ctx->tw, 0ull,
                             nb, fb->nseq, 32u);
          hipLaunchKernelGGL(k2, dim3(grid_for(n2s, fb->ncu * fb->wg_per_cu)), dim3(fb->nt2), fb->lds2, ctx->stream, g, fb->A, fb->X + c * Ls,
                             ctx->tw, nb, fb->nseq, 4u);
        }
        const uint64_t n3s = (uint64_t)(g.C >> g.logT3) * nb;
        if (fb->msub) {
          // freq_res = R * 2^k: the spectrum in pseudo-channel order (second buffer), the inverse pass on the R * nchan_subband
          // pseudo-channels keeping whole transforms (complex rows into Y), then the radix-R step in time into the caller's output
          const uint64_t Mi = 1ull << g.logM, xe = (uint64_t)fb->max_parts * fb->nseq * fb->L,
                         ye = (uint64_t)g.C * g.npol * fb->max_parts * Mi;
          if (!fb->Xp && hipMalloc((void**)&fb->Xp, xe * sizeof(cf)) != hipSuccess)
            return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform: hipMalloc of the pseudo-channel spectrum failed");
          if (!fb->Y && hipMalloc((void**)&fb->Y, ye * sizeof(cf)) != hipSuccess)
            return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform: hipMalloc of the pseudo-channel time series failed");
          fb_launch_sub_combine(ctx->stream, g, fb->X, nb * fb->nseq, fb->ncu, fb->Xp, fb->out_M, fb->msub);
          // Y[pseudo-channel][pol][part of the group][Mi] complex: rows (pseudo-channel, pol), parts 2*Mi floats apart
          FbOut yo = {1, (float*)fb->Y, (uint64_t)g.npol * fb->max_parts * Mi * 2, (uint64_t)fb->max_parts * Mi * 2, Mi * 2, 0, 2, 0};
          hipLaunchKernelGGL(fb->k3, dim3(grid_for(n3s, fb->ncu * fb->wg3)), dim3(fb->nt3), fb->lds3, ctx->stream, g, fb->Xp, kern, yo, ctx->tw,
                             0ull, nb, nb);
          TimeCombine tc = {fb->Y, (uint64_t)g.npol * fb->max_parts * Mi, (uint64_t)fb->max_parts * Mi, (uint32_t)g.logM, fb->out_M,
                            fb->out_nfilt_pos, fb->out_nkeep, fb->out_C, (uint32_t)g.npol, part0, nb};
          FbOut cu = co;
          cu.chan0 = ichan * fb->out_C;
          if (cu.kind == 1 || cu.kind == 2) fb_launch_time_combine(ctx->stream, tc, cu, fb->msub, fb->ncu);
          continue;
        }
        fb_launch_sub_combine(ctx->stream, g, fb->X, nb * fb->nseq, fb->ncu);
        if (co.kind == 3) {
          const int rc = fb_launch_fused(fb, k3, fb->X, kern, co, part0, nb, fused_segmented);
          if (rc != DSPSR_AMD_OK) return rc;
        } else {
          hipLaunchKernelGGL(k3, dim3(grid_for(n3s, fb->ncu * fb->wg3)), dim3(fb->nt3), fb->lds3, ctx->stream, g, fb->X, kern, co, ctx->tw,
                             part0, nb, nb);
        }
        continue;
      }
      if (two) {
        // Two passes (FB_HAS(6)): regroup per column, whole-column forward pass, rows + inverse pass -- the spectrum never
        // leaves the chip.  Launches are whole groups (the segmented fused fold pays a memset and a combine pass per launch).
        const uint32_t Fb = 1u << g.logFb2;
        FbIn cr = ci;
        cr.kind = 3;
        cr.base = fb->Rt;
        if (fb->k1c) {
          fb_launch_raw_cols(dim3((uint32_t)(fb->L / 8192), nb), ctx->stream, g, ci, fb->Rt, part0);
          const uint64_t n1c = (uint64_t)Fb * 2 * nb;
          hipLaunchKernelGGL(fb->k1c, dim3(grid_for(n1c, fb->ncu)), dim3(512), fb->lds1c, ctx->stream, g, cr, fb->A, ctx->tw, nb, 2u, 32u);
        } else {
          const FbGeom& q = fb->g1t;
          const uint32_t Fa = 1u << q.logM;
          fb_launch_raw_transpose(dim3((Fb + 255) / 256, (Fa + 63) / 64, nb * 2), ctx->stream, q, ci, fb->Rt, part0);
          const uint64_t n1t = (uint64_t)(Fb >> q.logT1) * 2 * nb;
          hipLaunchKernelGGL(fb->k1t, dim3(grid_for(n1t, fb->ncu)), dim3(fb->nt1t), fb->lds1t, ctx->stream, q, cr, fb->A, ctx->tw, part0,
                             nb, 2u, 32u);
        }
        const uint32_t tiles = g.C >> g.logFb2;
        if (co.kind == 3) {
          const int rc = fb_launch_fused(fb, fb->k2rf, fb->A, kern, co, part0, nb, fused_segmented, true);
          if (rc != DSPSR_AMD_OK) return rc;
        } else {
          hipLaunchKernelGGL(fb->k2r, dim3(grid_for((uint64_t)tiles * nb, fb->ncu)), dim3(512), fb->lds2r, ctx->stream, g, fb->A, kern, co,
                             ctx->tw, part0, nb, nb);
        }
        continue;
      }
      // persistent grids: one workgroup per CU (LDS-limited), a multiple of 8 so the XCD-aware item order applies
      const uint64_t n1 = (uint64_t)(Rr >> g.logT1) * fb->nseq * nb, n2 = (uint64_t)(M >> g.logT2) * fb->nseq * nb,
                     n3 = g.four_pass ? 0 : (uint64_t)(g.C >> g.logT3) * nb;
      // XCD dealing of the persistent items (wgfft.h persistent_item); the environment overrides are for experiments
      const int env_run1 = FB_ENV_INT("DSPSR_AMD_RUN1", 0), env_run2 = FB_ENV_INT("DSPSR_AMD_RUN2", 0),
                env_run3 = FB_ENV_INT("DSPSR_AMD_RUN3", 0);
      const uint32_t run1 = env_run1 > 0 ? env_run1 : 32, run2 = env_run2 > 0 ? env_run2 : 4, run3 = env_run3 > 0 ? env_run3 : nb;
      if (pret) {
        fb_launch_raw_transpose(dim3((Rr + 255) / 256, (M + 63) / 64, nb * fb->nseq), ctx->stream, g, ci, fb->Rt, part0);
        ci.kind = 3;
        ci.base = fb->Rt;
      } else if (pretf) {
        fb_launch_float_transpose(dim3((Rr + FB_FT_COLS - 1) / FB_FT_COLS, (M + FB_FT_ROWS - 1) / FB_FT_ROWS, nb * fb->nseq), ctx->stream, g, ci, fb->X, part0);
        ci.kind = 5;
        ci.base = fb->X;
      }
      if (k1d)
        hipLaunchKernelGGL(k1d, dim3(grid_for(n1 / 2, fb->ncu * fb->wg1)), dim3(fb->nt1), fb->lds1, ctx->stream, g, ci, fb->A, ctx->tw,
                           part0, nb, fb->nseq, run1 / 2 ? run1 / 2 : 1u);     // (run is a divisor in persistent_item: never 0)
      else
        hipLaunchKernelGGL(k1, dim3(grid_for(n1, fb->ncu * fb->wg1)), dim3(fb->nt1), fb->lds1, ctx->stream, g, ci, fb->A, ctx->tw,
                           part0, nb, fb->nseq, run1);
      ci = in; ci.ichan = ichan; ci.nchan = fb->cfg.input_nchan;
      if (in.kind == 0) ci.base = in_f32 + ichan * in_chan_stride_bytes_or_floats;
      // Pass 2 and the inverse pass run in sub-groups of a few parts, so that
      // part of the spectrum pass 2 has just written is still in the 256 MB Infinity Cache when the inverse pass reads
      // it (measured with whole groups of 8 / 16 / 32 parts: 31.5 / 33.4 / 35.2 µs per part in the inverse pass, pass 2
      // unchanged) while passes 0 and 1 keep the long launch their persistent workgroups want.
      // Sub-group = about 512 MB of spectrum (8 parts of the headline geometry; small geometries keep whole launches:
      // cut into 8 parts, -F 256:D loses 9 % and the 50 MHz sub-band geometry 24 %).
      const int p23sub_env = FB_ENV_INT("DSPSR_AMD_P23_SUB", -1);
      uint64_t p23auto = (512ull << 20) / (fb->part_elems * sizeof(cf));
      if (p23auto < 1) p23auto = 1;
      // (the fused kernel gains less, +1.4 % Msamples/s measured in three alternating runs, but consistently)
      const bool sub_fused = FB_ENV_INT("DSPSR_AMD_P23_SUB_FUSED", 1) != 0;
      // (segmented fused fold -- geometries with fewer channel tiles than compute units: every launch of the fused kernel
      //  brings a memset and a combine pass over the partial profiles, so whole launches win: 50 MHz sub-band geometry
      //  43.5k -> 46.0k Msamples/s, -F 256:D 60.3k -> 60.6-61.3k, tools/exp_p23.sh)
      const uint32_t p23sub = (g.four_pass || (co.kind == 3 && (!sub_fused || fused_segmented)) || p23sub_env == 0) ? nb
                              : (p23sub_env > 0 ? (uint32_t)p23sub_env : (uint32_t)(p23auto < nb ? p23auto : nb));
      if (p23sub < nb) {
        const size_t lds3s = co.kind == 3 ? fb->lds3f : fb->lds3;
        if (co.kind == 3) co.plan_cap = fb->plan_cap;
        for (uint32_t s0 = 0; s0 < nb; s0 += p23sub) {
          const uint32_t ns = nb - s0 < p23sub ? nb - s0 : p23sub;
          const uint64_t off = (uint64_t)s0 * fb->part_elems;
          const uint64_t n2s = (uint64_t)(M >> g.logT2) * fb->nseq * ns;
          const uint64_t n3s = co.kind == 3 ? (uint64_t)(g.C >> g.logT3) : (uint64_t)(g.C >> g.logT3) * ns;
          hipLaunchKernelGGL(k2, dim3(grid_for(n2s, fb->ncu * fb->wg_per_cu)), dim3(fb->nt2), fb->lds2, ctx->stream, g, fb->A + off,
                             fb->X + off, ctx->tw, ns, fb->nseq, run2);
          if (co.kind == 3) {
            const int rc = fb_launch_fused(fb, k3, fb->X + off, kern, co, part0 + s0, ns, fused_segmented);
            if (rc != DSPSR_AMD_OK) return rc;
          } else {
            hipLaunchKernelGGL(k3, dim3(grid_for(n3s, fb->ncu * fb->wg3)), dim3(fb->nt3), lds3s, ctx->stream, g, fb->X + off, kern, co,
                               ctx->tw, part0 + s0, ns, env_run3 > 0 ? (uint32_t)env_run3 : ns);
          }
        }
        continue;
      }
      hipLaunchKernelGGL(k2, dim3(grid_for(n2, fb->ncu * fb->wg_per_cu)), dim3(fb->nt2), fb->lds2, ctx->stream, g, fb->A, fb->X,
                         ctx->tw, nb, fb->nseq, run2);
      if (!g.four_pass) {
        // fused fold: one workgroup owns a tile (T3 channels) for all parts of the launch
        const uint64_t items3 = co.kind == 3 ? (uint64_t)(g.C >> g.logT3) : n3;
        const size_t lds3 = co.kind == 3 ? fb->lds3f : fb->lds3;
        if (co.kind == 3) co.plan_cap = fb->plan_cap;       // LDS left over behind the twiddle tables holds the part's fold plan
        if (co.kind == 3) {
          const int rc = fb_launch_fused(fb, k3, fb->X, kern, co, part0, nb, fused_segmented);
          if (rc != DSPSR_AMD_OK) return rc;
        } else {
          hipLaunchKernelGGL(k3, dim3(grid_for(items3, fb->ncu * fb->wg3)), dim3(fb->nt3), lds3, ctx->stream, g, fb->X, kern, co,
                             ctx->tw, part0, nb, run3);
        }
      } else {
        // two-pass inverse: X (whole spectrum) -> U (in the A buffer, dead after pass 2) -> output
        const uint64_t n3a = ((uint64_t)g.C << (g.logMb - g.logTm)) * nb, n3b = ((uint64_t)g.C << (g.logMa - g.logTt)) * nb;
        hipLaunchKernelGGL(k3a, dim3(grid_for(n3a, fb->ncu * fb->wg_per_cu)), dim3(fb->nt3), fb->lds3, ctx->stream, g, fb->X, kern,
                           fb->A, ctx->tw, nb, 8u);
        if (out.kind == 4 && fb->plan_wait) {      // the fused second pass reads the segment plan
          const int rc = fold_plan_wait(out.fold, fb->plan_wait);
          fb->plan_wait = nullptr;
          if (rc != DSPSR_AMD_OK) return rc;
        }
        hipLaunchKernelGGL(k3b, dim3(grid_for(n3b, fb->ncu * fb->wg_per_cu)), dim3(fb->nt4), fb->lds4, ctx->stream, g, fb->A, co,
                           ctx->tw, part0, nb, 8u);
      }
    }
    if (out.kind == 4) {      // every part of this sub-band has left its segment sums: add them to the profile in time order
      const int rc = fold_segment_combine(out.fold, fb->msum, co.chan0, g.C, (uint32_t)npart, g.nkeep, g.nfilt_pos, g.logTt, g.logMa,
                                          g.logMb, out.bin_start, out.piv);
      if (rc != DSPSR_AMD_OK) return rc;
    }
  }
  e = hipGetLastError();
  if (e != hipSuccess)
    return fb_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_perform: launch failed: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_filterbank_perform(dspsr_amd_filterbank* fb, const float* in_dev, uint64_t in_chan_stride,
                                            uint64_t in_pol_stride, float* out_dev, uint64_t out_chan_stride,
                                            uint64_t out_pol_stride, uint64_t npart, uint64_t in_step,
                                            uint64_t out_step)
{
  if (!fb || !in_dev) return DSPSR_AMD_EINVAL;
  const uint32_t ndim = fb->cfg.real_input ? 1 : 2;
  if (in_step % ndim)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: in_step=%llu not a multiple of ndim",
                   (unsigned long long)in_step);
  if (out_dev && out_step < 2ull * fb_out_nkeep(fb))
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: out_step=%llu < 2*nkeep=%u",
                   (unsigned long long)out_step, 2 * fb_out_nkeep(fb));
  const uint64_t nchan_out = (uint64_t)fb->cfg.input_nchan * fb_out_C(fb);
  const uint64_t row = npart ? (npart - 1) * out_step + 2ull * fb_out_nkeep(fb) : 0;      // floats one output row spans
  if (out_dev && npart && ((fb->cfg.npol > 1 && out_pol_stride < row) || (nchan_out > 1 && out_chan_stride < row)))
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: output rows of %llu floats overlap "
                   "(chan stride %llu, pol stride %llu)", (unsigned long long)row, (unsigned long long)out_chan_stride,
                   (unsigned long long)out_pol_stride);
  FbIn in = {0, in_dev, in_pol_stride, in_step / ndim, fb->cfg.input_nchan, 0, 1.0f};
  FbOut out = {out_dev ? 1 : 0, out_dev, out_chan_stride, out_pol_stride, out_step, 0, 2, 0};
  return fb_run(fb, in, out, npart, in_chan_stride);
}

extern "C" int dspsr_amd_filterbank_perform_raw(dspsr_amd_filterbank* fb, const int8_t* raw_dev, int raw_layout,
                                                float scale, float* out_dev, uint64_t out_chan_stride,
                                                uint64_t out_pol_stride, uint64_t npart, uint64_t out_step)
{
  if (!fb || !raw_dev) return DSPSR_AMD_EINVAL;
  if (raw_layout == DSPSR_AMD_RAW_CASPSR &&
      !(fb->cfg.real_input && fb->cfg.npol == 2 && fb->cfg.input_nchan == 1))
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_perform_raw: CASPSR layout needs real dual-pol single-channel input");
  if (raw_layout == DSPSR_AMD_RAW_UWB16 && (fb->cfg.real_input || fb->cfg.input_nchan != 1))
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_perform_raw: UWB 16-bit layout needs complex single-channel input");
  if (raw_layout != DSPSR_AMD_RAW_CASPSR && raw_layout != DSPSR_AMD_RAW_GENERIC && raw_layout != DSPSR_AMD_RAW_UWB16)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_raw: unknown raw layout %d", raw_layout);
  uint64_t step;
  dspsr_amd_filterbank_sizes(fb, nullptr, nullptr, &step, nullptr);
  FbIn in = {raw_kind(raw_layout), raw_dev, 0, step, fb->cfg.input_nchan, 0, scale};
  FbOut out = {out_dev ? 1 : 0, out_dev, out_chan_stride, out_pol_stride, out_step, 0, 2, 0};
  return fb_run(fb, in, out, npart, 0);
}

extern "C" int dspsr_amd_filterbank_perform_detect(dspsr_amd_filterbank* fb, const float* in_f32_dev,
                                                   uint64_t in_chan_stride, uint64_t in_pol_stride, uint64_t in_step,
                                                   const int8_t* raw_dev, int raw_layout, float scale, int state,
                                                   uint32_t ndim, float* det_dev, uint64_t det_chan_stride,
                                                   uint64_t det_pol_stride, uint64_t npart)
{
  if (!fb || !det_dev || (!in_f32_dev == !raw_dev)) return DSPSR_AMD_EINVAL;
  if (fb->cfg.npol != 2)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_perform_detect: Cannot detect polarization when npol != 2");
  if (ndim != 1 && ndim != 2 && ndim != 4)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_detect: invalid ndim=%u", ndim);
  if (state != DSPSR_AMD_COHERENCE && state != DSPSR_AMD_STOKES)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_detect: invalid state=%d", state);
  uint64_t step;
  dspsr_amd_filterbank_sizes(fb, nullptr, nullptr, &step, nullptr);
  FbIn in;
  if (in_f32_dev) {
    const uint32_t idim = fb->cfg.real_input ? 1 : 2;
    if (in_step % idim)
      return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_detect: in_step=%llu not a multiple of ndim", (unsigned long long)in_step);
    in = {0, in_f32_dev, in_pol_stride, in_step / idim, fb->cfg.input_nchan, 0, 1.0f};
  } else {
    if (raw_layout == DSPSR_AMD_RAW_CASPSR &&
        !(fb->cfg.real_input && fb->cfg.npol == 2 && fb->cfg.input_nchan == 1))
      return fb_fail(fb->ctx, DSPSR_AMD_EINVAL,
                     "dspsr_amd_filterbank_perform_detect: CASPSR layout needs real dual-pol single-channel input");
    if (raw_layout == DSPSR_AMD_RAW_UWB16 && (fb->cfg.real_input || fb->cfg.input_nchan != 1))
      return fb_fail(fb->ctx, DSPSR_AMD_EINVAL,
                     "dspsr_amd_filterbank_perform_detect: UWB 16-bit layout needs complex single-channel input");
    if (raw_layout != DSPSR_AMD_RAW_CASPSR && raw_layout != DSPSR_AMD_RAW_GENERIC && raw_layout != DSPSR_AMD_RAW_UWB16)
      return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_detect: unknown raw layout %d", raw_layout);
    in = {raw_kind(raw_layout), raw_dev, 0, step, fb->cfg.input_nchan, 0, scale};
  }
  {
    // rows must not overlap: channel-major (TimeSeries FPT order) or plane-major layouts are accepted
    const uint64_t nchan_out = (uint64_t)fb->cfg.input_nchan * fb_out_C(fb), row = npart * fb_out_nkeep(fb) * ndim, planes = 4 / ndim;
    const bool chan_major = (planes == 1 || det_pol_stride >= row) &&
                            (nchan_out == 1 || det_chan_stride >= (planes - 1) * det_pol_stride + row);
    const bool plane_major = planes > 1 && (nchan_out == 1 || det_chan_stride >= row) &&
                             det_pol_stride >= (nchan_out - 1) * det_chan_stride + row;
    if (npart && !chan_major && !plane_major)
      return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_detect: detected rows of %llu floats overlap "
                     "(chan stride %llu, pol stride %llu)", (unsigned long long)row, (unsigned long long)det_chan_stride,
                     (unsigned long long)det_pol_stride);
  }
  FbOut out = {2, det_dev, det_chan_stride, det_pol_stride, 0, state, ndim, 0};
  return fb_run(fb, in, out, npart, in_chan_stride);
}

extern "C" int dspsr_amd_filterbank_npass(const dspsr_amd_filterbank* fb, int raw_input)
{
  if (!fb) return 0;
  if (fb->g.four_pass) return 4;
  return fb->two_pass && raw_input ? 2 : 3;
}

extern "C" int dspsr_amd_filterbank_fold_is_fused(const dspsr_amd_filterbank* fb)
{
  if (!fb || fb->msub) return 0;            // (freq_res = 3 * 2^k / 5 * 2^k: the last step is a pass of its own, k_time_combine)
  // (segment sums pay when most of the transform is kept: at -F 64:D -x 16384 only 1817 of 16384 samples are, the unfused pass
  //  writes just those, and the fused one measured 541 against 458 us per 8 parts)
  if (fb->g.four_pass)
    return fb->cfg.fused_fold != DSPSR_AMD_FUSED_NEVER && fb->k3bf && fb->g.logTt >= 3 &&
           (2ull * fb->g.nkeep >= (1ull << fb->g.logMf) || fb->cfg.fused_fold == DSPSR_AMD_FUSED_ALWAYS) ? 3 : 0;
  if (fb->g.nkeep >= 65536) return 0;
  if (fb->cfg.fused_fold == DSPSR_AMD_FUSED_ALWAYS) return 1;
  if (fb->cfg.fused_fold == DSPSR_AMD_FUSED_NEVER) return 0;
  const uint64_t tiles = (uint64_t)(fb->g.C >> fb->g.logT3);
  if (tiles >= fb->ncu) return 1;           // one workgroup per tile fills the chip: exact time-order sums
  return tiles >= 8 ? 2 : 0;                // fewer tiles: the parts of a launch are folded in runs (re-associated sums)
}

extern "C" int dspsr_amd_filterbank_perform_fold(dspsr_amd_filterbank* fb, const float* in_f32_dev,
                                                 uint64_t in_chan_stride, uint64_t in_pol_stride, uint64_t in_step,
                                                 const int8_t* raw_dev, int raw_layout, float scale, int state,
                                                 dspsr_amd_fold* fold, uint64_t npart)
{
  if (!fb || !fold || (!in_f32_dev == !raw_dev)) return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = fb->ctx;
  if (fb->cfg.npol != 2)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_fold: Cannot detect polarization when npol != 2");
  if (state != DSPSR_AMD_COHERENCE && state != DSPSR_AMD_STOKES)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_fold: invalid state=%d", state);
  const uint32_t nchan = fb->cfg.input_nchan * fb_out_C(fb);
  // profile shapes: npol 1 x ndim 4 (one float4 per bin, the CPU default, LoadToFoldConfig.C:104) or npol 2 x ndim 2 (rows
  // (PP, QQ) and (Re, Im): what the reference's GPU pipeline detects and folds, LoadToFold1.C:1105-1109)
  const bool planes2 = fold->npol == 2 && fold->ndim == 2;
  if (!fold->profile || fold->nchan != nchan || !((fold->npol == 1 && fold->ndim == 4) || planes2))
    return fb_fail(ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_perform_fold: fold shape must be nchan=%u with npol=1 ndim=4 or npol=2 ndim=2 (is %u/%u/%u)",
                   nchan, fold->nchan, fold->npol, fold->ndim);
  if (fold->folding_nbin != fold->nbin)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dsp::Fold::fold folding_nbin != output->nbin (%u != %u)",
                   fold->folding_nbin, fold->nbin);
  if (npart > 0xffffffffull) return DSPSR_AMD_EINVAL;
  uint64_t step;
  dspsr_amd_filterbank_sizes(fb, nullptr, nullptr, &step, nullptr);
  FbIn in;
  if (in_f32_dev) {
    const uint32_t idim = fb->cfg.real_input ? 1 : 2;
    if (in_step % idim)
      return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_fold: in_step=%llu not a multiple of ndim", (unsigned long long)in_step);
    in = {0, in_f32_dev, in_pol_stride, in_step / idim, fb->cfg.input_nchan, 0, 1.0f};
  } else {
    if (raw_layout == DSPSR_AMD_RAW_CASPSR && !(fb->cfg.real_input && fb->cfg.npol == 2 && fb->cfg.input_nchan == 1))
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_fold: CASPSR layout needs real dual-pol single-channel input");
    if (raw_layout == DSPSR_AMD_RAW_UWB16 && (fb->cfg.real_input || fb->cfg.input_nchan != 1))
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_fold: UWB 16-bit layout needs complex single-channel input");
    if (raw_layout != DSPSR_AMD_RAW_CASPSR && raw_layout != DSPSR_AMD_RAW_GENERIC && raw_layout != DSPSR_AMD_RAW_UWB16)
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_fold: unknown raw layout %d", raw_layout);
    in = {raw_kind(raw_layout), raw_dev, 0, step, fb->cfg.input_nchan, 0, scale};
  }
  // In the fused kernel one workgroup owns a tile of channels for all parts of a launch (that keeps the sums in
  // time order), so it only pays when the channel tiles alone fill the chip: measured on MI355X, 512 tiles
  // (-F 1024:D -x 4096) +14 %, 128 tiles (-F 256:D) -13 %, 32 tiles (-F 512:D on a 50 MHz sub-band) 4x slower.
  // Below the threshold, and for the four-pass geometry, Detection and Fold run as separate launches on an
  // internal block -- the sums are bit-identical either way.
  // (a bound profile whose rows are not float4 aligned also takes the separate launches: the fold kernel adds scalars)
  const bool prof_vec4 = planes2 ? (fold->span % 2 == 0 && ((uintptr_t)fold->profile % 16) == 0)
                                 : (fold->span % 4 == 0 && ((uintptr_t)fold->profile % 16) == 0);
  if (dspsr_amd_filterbank_fold_is_fused(fb) == 3 && !planes2 && prof_vec4 && npart &&
      fold_plan_max_run(fold) >= FOLD_LONG_RUN_HOST && npart * (uint64_t)fb->g.nkeep < (1ull << 32)) {
    // Four-pass geometry (dsp::Convolution shapes, -F N:D with a long response) and wide phase bins: the second inverse pass
    // reduces its tile to the sums of the Tt-sample segments it holds and a second kernel adds those in time order -- the
    // detected time series (16 bytes per sample, written and read once) never reaches HBM.  Plans that do not qualify
    // (gaps from zero weights, short inner intervals, a fold that does not cover the call) take Detection + Fold below.
    bool ok = false;
    const uint32_t* d_off = nullptr; const uint32_t* d_blk = nullptr; const uint32_t* d_bs = nullptr;
    const Interval* d_siv = nullptr;
    PlanSlot* sslot = nullptr;
    int rc = fold_build_segment_plan(fold, npart * (uint64_t)fb->g.nkeep, 1u << fb->g.logTt, &ok, &d_off, &d_blk, &d_bs, &d_siv, &sslot);
    if (rc != DSPSR_AMD_OK) return rc;
    if (ok) {
      const size_t need = ((size_t)fb->g.C * npart << (fb->g.logMf - fb->g.logTt)) * 8;       // segments x 2 pieces x float4
      if (need > fb->msum_floats) {
        (void)hipStreamSynchronize(ctx->stream);
        if (fb->msum) (void)hipFree(fb->msum);
        fb->msum = nullptr; fb->msum_floats = 0;
        if (hipMalloc((void**)&fb->msum, need * sizeof(float)) != hipSuccess)
          return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform_fold: hipMalloc of %zu segment-sum bytes failed", need * sizeof(float));
        fb->msum_floats = need;
      }
      FbOut sout = {4, fb->msum, 0, 0, 0, state, 4, 0, fold->nbin, fold->span / 4, 1u, fold->span, nullptr, fold->nchan, 0, fold, d_off,
                    (uint32_t)npart, 0, d_siv, d_blk, d_bs};
      fb->plan_wait = sslot;
      rc = fb_run(fb, in, sout, npart, in_chan_stride);
      if (fb->plan_wait) { (void)fold_plan_wait(fold, sslot); fb->plan_wait = nullptr; }       // (no consumer was launched)
      const int rc2 = fold_part_plan_submitted(fold, sslot);
      return rc != DSPSR_AMD_OK ? rc : rc2;
    }
  }
  const int fmode = dspsr_amd_filterbank_fold_is_fused(fb);
  if ((fmode != 1 && fmode != 2) || !prof_vec4 || fold_plan_max_run(fold) >= (uint32_t)FB_ENV_INT("DSPSR_AMD_FUSED_MAX_RUN", (int)FOLD_FUSED_MAX_RUN)) {
    const uint64_t row = npart * fb_out_nkeep(fb) * 4;                       // floats per channel
    const size_t need = (size_t)row * nchan;
    if (!need) return DSPSR_AMD_OK;
    if (need > fb->det_floats) {
      (void)hipStreamSynchronize(ctx->stream);
      if (fb->det) (void)hipFree(fb->det);
      fb->det = nullptr;
      fb->det_floats = 0;
      if (hipMalloc((void**)&fb->det, need * sizeof(float)) != hipSuccess)
        return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform_fold: hipMalloc of %zu bytes failed", need * sizeof(float));
      fb->det_floats = need;
    }
    // (ndim 2: the channel's two rows of npart*nkeep float2 one after the other)
    FbOut dout = {2, fb->det, row, planes2 ? row / 2 : 0, 0, state, planes2 ? 2u : 4u, 0};
    const int rc = fb_run(fb, in, dout, npart, in_chan_stride);
    if (rc != DSPSR_AMD_OK) return rc;
    return dspsr_amd_fold_fold(fold, fb->det, row, planes2 ? row / 2 : 0);
  }
  const uint32_t* d_start = nullptr;
  const Interval* d_iv = nullptr;
  PlanSlot* slot = nullptr;
  int rc = fold_build_part_plan(fold, fb->g.nkeep, (uint32_t)npart, &d_start, &d_iv, &slot);
  if (rc != DSPSR_AMD_OK) return rc;
  // (float4 from one channel to the next: span/4, or both rows of the channel, 2*span/4)
  FbOut out = {3, fold->profile, 0, 0, 0, state, 4, 0, fold->nbin, planes2 ? fold->span / 2 : fold->span / 4, planes2 ? 2u : 1u,
               fold->span, nullptr, fold->nchan, 0, fold, d_start, (uint32_t)npart, 0, d_iv};
  fb->plan_wait = slot;
  rc = fb_run(fb, in, out, npart, in_chan_stride);
  if (fb->plan_wait) { (void)fold_plan_wait(fold, slot); fb->plan_wait = nullptr; }            // (no consumer was launched)
  const int rc2 = fold_part_plan_submitted(fold, slot);
  return rc != DSPSR_AMD_OK ? rc : rc2;
}

#else
}  // namespace dspsr_amd
#endif  // FB_HAS(0)
