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
#include "stamps.h"

namespace dspsr_amd {

struct TfpParams {
  const uint8_t* raw;      // generic 8-bit order, real, 2 pols, 1 channel: byte 2*t + p   (or the CASPSR 4+4 interleave)
  float* out;              // [nout][nchan][npol_out]
  uint64_t npart;          // parts available
  uint32_t sfactor;        // time scrunch factor (>= 1)
  uint32_t pscrunch;       // 1: sum the two polarisations (Intensity), 0: PPQQ
  float scale;
  int logT;                // columns of a workgroup tile = 2 polarisations x 2^(logT-1) consecutive parts
  int caspsr;
};

// Round 3: the 2*nchan REAL samples of one polarisation and part are transformed as nchan COMPLEX points
//   z[n] = x[2n] + i x[2n+1],  Z = FFT_nchan(z),  X[k] = A + w^k B,  A = (Z[k] + conj Z[nchan-k]) / 2,
//   B = (Z[k] - conj Z[nchan-k]) / 2i,  w = exp(-i pi / nchan)          (k < nchan; the Nyquist bin is not kept)
// instead of packing the two polarisations into one complex sequence of 2*nchan points and splitting the spectrum
// (rounds 1-2).  Both polarisations of a part are now the two columns of a thread's butterfly pair: one 32-bit word of
// the byte stream holds (p0[2n], p1[2n], p0[2n+1], p1[2n+1]) = (Re z0, Re z1, Im z0, Im z1), the transform is one radix-2
// level shorter -- at nchan = 4096 exactly three radix-16 stages instead of three plus a radix-2 stage with its own LDS
// exchange and barriers -- and the pair (Z[k], Z[nchan-k]) yields bins k AND nchan-k (X[nchan-k] = conj(A - w^k B)), so a
// thread owns the bin pairs k <= nchan/2.  The pass is bound by its exchanged stages (0.8 TB/s of input), so the stage
// it no longer runs is what it gains.

// the (p0, p1) byte pairs of samples 2n and 2n+1 of one part: one aligned word in the generic order, two half words
// (4 bytes apart) in the CASPSR order
template <bool CASPSR> DEV uint32_t tfp_word(const uint8_t* img, const uint32_t t0 /* even sample index */)
{
  if constexpr (CASPSR) {
    const uint8_t* a = img + (t0 >> 2) * 8 + (t0 & 3);
    const uint32_t p0 = *(const uint16_t*)a, p1 = *(const uint16_t*)(a + 4);          // (p0[t0], p0[t0+1]), (p1[t0], p1[t0+1])
    return (p0 & 0xffu) | ((p1 & 0xffu) << 8) | ((p0 >> 8) << 16) | ((p1 >> 8) << 24);
  } else {
    return (uint32_t)*(const uint16_t*)(img + 2 * t0) | ((uint32_t)*(const uint16_t*)(img + 2 * t0 + 2) << 16);
  }
}
// the same from a 4-byte aligned image (the LDS copy of a 16-byte aligned block): generic order = one aligned word
template <bool CASPSR> DEV uint32_t tfp_word_aligned(const uint8_t* img, const uint32_t t0)
{
  if constexpr (CASPSR) return tfp_word<true>(img, t0);
  else return *(const uint32_t*)(img + 2 * t0);
}

// COAL: the parts of a tile are ONE contiguous range of 2^15 bytes.  Every thread loads 16-byte pieces of the range
// (prefetched one tile ahead, 4 registers per piece), the bytes go through the -- at that moment idle -- exchange buffer, and
// the thread picks its 16 words from LDS.  Needs a 16-byte aligned block (the host checks; otherwise the words are loaded
// from global memory as half words).
template <int LOGC, bool CASPSR, bool COAL>
__global__ __launch_bounds__(512) void k_tfp(const TfpParams p, const cf* __restrict__ tw)
{
  typedef FftPlan<LOGC> P;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const uint32_t nt = blockDim.x;
  // columns of the tile: always 14 - LOGC (dspsr_amd_tfp_filterbank sets p.logT so), a compile-time constant here: with the
  // run-time value every exchange address of the transform carried two uniform branches (150-171 per tile at 4096 channels)
  constexpr int logT = 14 - LOGC, logTp = logT - 1;            // columns; parts per tile
  const uint32_t Tp = 1u << logTp, C = 1u << LOGC;             // C = nchan complex points per polarisation and part
  const uint32_t npol_out = p.pscrunch ? 1 : 2;
  const uint64_t nout = p.npart / p.sfactor;
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LOGC>(lds, ltw_off, tw, tid, nt);
  // bin pairs owned by a thread: k = tid + j*nt < C/2, paired with C-k; k = 0 is paired with C/2 (both are their own mirrors)
  constexpr int NB = (1 << LOGC) / 2 > 512 ? (1 << LOGC) / 2 / 512 : 1;
  float wc[NB], ws[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) {
    // w^k = exp(-i pi k / C): argument of v_cos/v_sin in revolutions, k / 2C exact in float
    const float x = (float)(threadIdx.x + j * nt) * __uint_as_float((uint32_t)(127 - (LOGC + 1)) << 23);
    wc[j] = __builtin_amdgcn_cosf(x);
    ws[j] = __builtin_amdgcn_sinf(x);
  }
  const float hs = 0.5f * p.scale;

  // work items: groups of Tp consecutive parts, dealt in order so that a workgroup owns whole output samples
  // when sfactor >= Tp, or several whole output samples when sfactor < Tp (host guarantees sfactor % Tp == 0
  // or Tp % sfactor == 0)
  const uint32_t groups_per_out = p.sfactor > Tp ? p.sfactor >> logTp : 1;
  const uint64_t nitem = p.sfactor > Tp ? nout : (nout * p.sfactor + Tp - 1) >> logTp;
  const uint64_t raw_bytes = p.npart * (uint64_t)C * 4;         // 2C samples x 2 polarisations per part

  // element i of a thread's first-stage butterfly pair: column pair (2*part_local, +1) = both polarisations, position n
  auto elem = [&](const int g2, const int i, uint32_t& pl, uint32_t& n) {
    const uint32_t e = first_stage_elem<LOGC>(tid, logT, g2, i);
    pl = (e & ((1u << logT) - 1)) >> 1;
    n = e >> logT;
  };
  uint32_t rw[NPAIR];                                           // non-COAL: the words of the next tile
  auto fetch = [&](const uint64_t group) {
    const uint64_t part0 = group << logTp;
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        uint32_t pl, n;
        elem(g2, i, pl, n);
        const uint64_t part = part0 + pl;
        rw[(g2 / 2) * P::R1 + i] = part < p.npart ? tfp_word<CASPSR>(p.raw + part * (uint64_t)C * 4, 2 * n) : 0x80808080u;
      }
  };
  constexpr uint32_t NCH = (PTS * 2) / 16;                      // 16-byte pieces per thread: 2^15 bytes per tile
  uint4 piece[NCH];
  auto fetch_pieces = [&](const uint64_t group) {
    const uint64_t base = (group << logTp) * (uint64_t)C * 4;
#pragma unroll
    for (uint32_t r = 0; r < NCH; r++) {
      const uint64_t off = base + 16ull * (tid + r * nt);
      piece[r] = off + 16 <= raw_bytes ? *(const uint4*)(p.raw + off) : make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    }
  };
  if (blockIdx.x < nitem) {
    if constexpr (COAL) fetch_pieces((uint64_t)blockIdx.x * groups_per_out);
    else fetch((uint64_t)blockIdx.x * groups_per_out);
  }
  FB_ST_BEGIN(8);
  for (uint64_t item = blockIdx.x; item < nitem; item += gridDim.x) {
    float acc[NB][2][2];                                        // [bin pair][k / C-k][pol]
#pragma unroll
    for (int j = 0; j < NB; j++) acc[j][0][0] = acc[j][0][1] = acc[j][1][0] = acc[j][1][1] = 0.f;
    for (uint32_t gi = 0; gi < groups_per_out; gi++) {
      const uint64_t group = item * groups_per_out + gi;
      asm volatile("" : "+v"(tid));
      cx2 x[NPAIR];
      FB_ST(8, 0);
      if constexpr (COAL) {
        // the tile's bytes, in file order, into the exchange buffer (the previous tile's read-back ended with a barrier)
        uint8_t* img = (uint8_t*)lds;
#pragma unroll
        for (uint32_t r = 0; r < NCH; r++) *(uint4*)(img + 16u * (tid + r * nt)) = piece[r];
        __syncthreads();
#pragma unroll
        for (int g2 = 0; g2 < P::G1; g2 += 2)
#pragma unroll
          for (int i = 0; i < P::R1; i++) {
            uint32_t pl, n;
            elem(g2, i, pl, n);
            rw[(g2 / 2) * P::R1 + i] = tfp_word_aligned<CASPSR>(img + (pl << (LOGC + 2)), 2 * n);
          }
      }
#pragma unroll
      for (int h = 0; h < NPAIR; h++) {
        const uint32_t w = rw[h];                               // (p0[2n], p1[2n], p0[2n+1], p1[2n+1])
        x[h].x = (v2f){__builtin_fmaf((float)(int8_t)(w & 0xff), p.scale, hs), __builtin_fmaf((float)(int8_t)((w >> 8) & 0xff), p.scale, hs)};
        x[h].y = (v2f){__builtin_fmaf((float)(int8_t)((w >> 16) & 0xff), p.scale, hs), __builtin_fmaf((float)(int8_t)(w >> 24), p.scale, hs)};
      }
      FB_ST(8, 1);                               // image through LDS + decode
      {
        const uint64_t next = gi + 1 < groups_per_out ? group + 1 : (item + gridDim.x) * groups_per_out;
        if (gi + 1 < groups_per_out || item + gridDim.x < nitem) {
          if constexpr (COAL) fetch_pieces(next);
          else fetch(next);
        }
      }
      if constexpr (COAL) __syncthreads();      // every thread has taken its words: the exchanges may overwrite the image
      FB_ST(8, 2);                               // next tile's loads issued
      // staged transform: one plane of C float4 per part (column pair), (Re p0, Re p1, Im p0, Im p1) -- the register order of
      // the pair and what the read-back wants for packed arithmetic; the planes are 8 float4 apart modulo the bank period, so the
      // lanes of a store (two parts, consecutive positions) and of a read-back (consecutive bins of one part) fall on
      // different banks (position-major staging read back with a 2-way conflict)
      float4* const stg = (float4*)lds;
      const uint32_t plane = C + (Tp <= 64 ? 8u : 0u);       // (the buffer's slack holds 512 float4 of padding)
      auto store = [&](const uint32_t col, const uint32_t pp, const uint32_t pstride, auto& v) {
        constexpr int R = sizeof(v) / sizeof(v[0]);
        float4* const d = stg + (col >> 1) * plane + pp;
#pragma unroll
        for (int k = 0; k < R; k++) d[k * pstride] = make_float4(v[k].x[0], v[k].x[1], v[k].y[0], v[k].y[1]);
      };
      wgfft<LOGC, -1, true>(lds, ltw_off, tid, logT, x, store);
      __syncthreads();
      FB_ST(8, 3);                               // transform + staging
      // real-transform post-processing, power, time scrunch (parts added in time order)
      const uint64_t part_first = group << logTp, part_end = nout * p.sfactor;
      const uint32_t phase_first = (uint32_t)(part_first % p.sfactor);       // wave-uniform: no division per bin
      const uint64_t out_first = part_first / p.sfactor;
      // powers of bins k and C-k of both polarisations (packed: .x = p0, .y = p1) from (Z[k], Z[C-k]) of one part;
      // staged float4 = (Re p0, Re p1, Im p0, Im p1)
      auto powers1 = [&](const float4 zk, const float4 zm, const float c, const float sn, v2f& pk, v2f& pm) {
        const v2f zr = {zk.x, zk.y}, zi = {zk.z, zk.w}, mr = {zm.x, zm.y}, mi = {zm.z, zm.w};
        const v2f ar = 0.5f * (zr + mr), ai = 0.5f * (zi - mi);            // A = (Z[k] + conj Z[C-k]) / 2
        const v2f br = 0.5f * (zi + mi), bi = 0.5f * (mr - zr);            // B = (Z[k] - conj Z[C-k]) / 2i
        const v2f wr = c * br + sn * bi, wi = c * bi - sn * br;            // w^k B,  w^k = (c, -sn)
        const v2f xr = ar + wr, xi = ai + wi, yr = ar - wr, yi = ai - wi;  // X[k] = A + w^k B,  X[C-k] = conj(A - w^k B)
        pk = xr * xr; pk += xi * xi;                                        // TFPFilterbank.C:56-59: Re^2 then += Im^2
        pm = yr * yr; pm += yi * yi;
      };
      auto powers = [&](const uint32_t k, const uint32_t c2, const float c, const float sn, float (&pw)[2][2]) {
        v2f pk, pm;
        if (k == 0) {               // bins 0 and C/2 are their own mirrors: X[0] from Z[0] (w = 1), X[C/2] from Z[C/2] (w = -i)
          const float4 z0 = stg[c2 * plane], zh = stg[c2 * plane + C / 2];
          v2f unused;
          powers1(z0, z0, 1.0f, 0.0f, pk, unused);
          powers1(zh, zh, 0.0f, 1.0f, pm, unused);
        } else {
          powers1(stg[c2 * plane + k], stg[c2 * plane + (C - k)], c, sn, pk, pm);
        }
        if (p.pscrunch) { pk[0] += pk[1]; pm[0] += pm[1]; pk[1] = pm[1] = 0.f; }   // :79-80 pol sum BEFORE the time sum
        pw[0][0] = pk[0]; pw[0][1] = pk[1]; pw[1][0] = pm[0]; pw[1][1] = pm[1];
      };
      // Whole group inside one output sample (tscrunch >= Tp, all parts present): no per-part bookkeeping.
      // 0 + p0 == p0, so starting an output sample from zero gives the sums of TScrunch.C:193-200 bit for bit.
      const bool whole = p.sfactor >= Tp && part_first + Tp <= part_end;
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const uint32_t k = tid + j * nt;
        if (k >= C / 2 && C > 1) continue;
        const uint32_t km = k ? C - k : C / 2;                               // the second bin of the pair
        const bool two = km != k;                                            // (C = 1: bin 0 alone)
        if (whole) {
          const bool emit = phase_first + Tp == p.sfactor;
          float s[2][2];
#pragma unroll
          for (int b = 0; b < 2; b++) { s[b][0] = phase_first == 0 ? 0.f : acc[j][b][0]; s[b][1] = phase_first == 0 ? 0.f : acc[j][b][1]; }
          for (uint32_t c2 = 0; c2 < Tp; c2++) {
            float pw[2][2];
            powers(k, c2, wc[j], ws[j], pw);
#pragma unroll
            for (int b = 0; b < 2; b++) { s[b][0] += pw[b][0]; s[b][1] += pw[b][1]; }
          }
#pragma unroll
          for (int b = 0; b < 2; b++) { acc[j][b][0] = s[b][0]; acc[j][b][1] = s[b][1]; }
          if (emit) {
            float* o = p.out + (out_first * C + k) * npol_out;
            o[0] = s[0][0];
            if (!p.pscrunch) o[1] = s[0][1];
            if (two) {
              float* om = p.out + (out_first * C + km) * npol_out;
              om[0] = s[1][0];
              if (!p.pscrunch) om[1] = s[1][1];
            }
          }
        } else {
          uint32_t phase = phase_first;
          uint64_t oidx = out_first;
          for (uint32_t c2 = 0; c2 < Tp; c2++) {
            if (part_first + c2 >= part_end) break;
            float pw[2][2];
            powers(k, c2, wc[j], ws[j], pw);
#pragma unroll
            for (int b = 0; b < 2; b++) {
              if (phase == 0) { acc[j][b][0] = pw[b][0]; acc[j][b][1] = pw[b][1]; }       // TScrunch.C:193-194
              else { acc[j][b][0] += pw[b][0]; acc[j][b][1] += pw[b][1]; }                 // TScrunch.C:199-200
            }
            if (++phase == p.sfactor) {
              float* o = p.out + (oidx * C + k) * npol_out;
              o[0] = acc[j][0][0];
              if (!p.pscrunch) o[1] = acc[j][0][1];
              if (two) {
                float* om = p.out + (oidx * C + km) * npol_out;
                om[0] = acc[j][1][0];
                if (!p.pscrunch) om[1] = acc[j][1][1];
              }
              phase = 0;
              oidx++;
            }
          }
        }
      }
      __syncthreads();    // LDS is overwritten by the next tile's exchanges
      FB_ST(8, 4);                               // split, powers, time scrunch, stores
      FB_ST_TILE(8, 5);
    }
  }
  FB_ST_END(8);
}

// nchan = 4096 (digifil -F 4096, BASELINE config 5) with an even time-scrunch factor: the real-transform post-processing on the LAST
// STAGE'S REGISTERS.  k_tfp stages the transform in LDS (16 ds_write_b128 per thread), passes a barrier and reads every bin with
// its Hermitian mirror back (16 ds_read_b128): a fourth 128 KB round trip through LDS per tile, 23 % of its cycles together with
// the arithmetic (stamps, profiles/r05_experiments.txt item 1).  Here the last stage runs in wgfft's MIRROR form: thread
// (j = tid / 4, column c = tid % 4) holds, for k < 16, Z[256 k + pa] and Z[256 k + pb] of ITS column (pa = j, pb = 256 - j; j = 0:
// pa = 0, pb = 128) -- and the mirror C - (256 k + pa) = 256 (15 - k) + pb is the other half of its register 15 - k.  The pair
// (v[k], halves of v[15-k] swapped) gives X at four bins per packed evaluation; nothing is staged.  The four lanes of a quad are
// the tile's (part, polarisation) columns of one position pair, so the polarisation sum and the time sum over the tile's two
// parts are DPP adds inside the quad, in the reference's order: (p0 + p1) per part (TFPFilterbank.C:79-80), parts in time order
// (TScrunch.C:193-200).  Lanes c = 0 (and 1 without pscrunch) carry the running sums of their 33 bins across the sfactor / 2
// tiles of an output sample.  Lanes j = 0 hold the self-mirrored columns p = 0 (mirror of 256 k: 256 (16 - k)) and p = 128: their
// mirror operands are picked per lane (v_cndmask), bin C/2 = 2048 is one extra scalar evaluation.
template <bool CASPSR, bool PSC>
__global__ __launch_bounds__(512) void k_tfp4k(const TfpParams p, const cf* __restrict__ tw)
{
  constexpr int LOGC = 12, logT = 2, logTp = 1;
  typedef FftPlan<LOGC> P;
  static_assert(P::NS == 3 && P::R1 == 16 && P::G1 == 2 && P::REM == 0, "k_tfp4k: 16 x 16 x 16");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  constexpr uint32_t nt = 512, C = 1u << LOGC;
  constexpr uint32_t npol_out = PSC ? 1 : 2;
  const uint64_t nout = p.npart / p.sfactor;
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LOGC>(lds, ltw_off, tw, tid, nt);
  const uint32_t col = threadIdx.x & 3u, jp = threadIdx.x >> 2, pa = jp, pb = jp ? 256u - jp : 128u;
  const bool j0 = jp == 0;
  // w^(256 k + pa), w^(256 k + pb), w = exp(-i pi / C) = (c, -s): argument bin / 2C revolutions, exact in float -- the values the
  // generic kernel uses for the same bins, so the two kernels agree bit for bit (32 registers held across the tile loop)
  v2f wc[8], ws[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float xa = (float)(256u * k + pa) * (1.0f / 8192.0f), xb = (float)(256u * k + pb) * (1.0f / 8192.0f);
    wc[k] = (v2f){__builtin_amdgcn_cosf(xa), __builtin_amdgcn_cosf(xb)};
    ws[k] = (v2f){__builtin_amdgcn_sinf(xa), __builtin_amdgcn_sinf(xb)};
  }
  // samples decoded at HALF scale: A = (Z[k] + conj Z[C-k]) / 2, B = (Z[k] - conj Z[C-k]) / 2i then need no factor 1/2, and a
  // power-of-two scale commutes with every rounding -- the same bits as k_tfp's 0.5f * (...)
  const float sc2 = 0.5f * p.scale, hs2 = 0.5f * sc2;
  const uint32_t groups_per_out = p.sfactor >> logTp;            // sfactor is even (host)
  const uint64_t nitem = nout;
  auto elem = [&](const int g2, const int i, uint32_t& pl, uint32_t& n) {
    const uint32_t e = first_stage_elem<LOGC>(tid, logT, g2, i);
    pl = (e & ((1u << logT) - 1)) >> 1;
    n = e >> logT;
  };
  // The tile's 2^15 bytes travel global -> LDS by LDS-DMA (16 bytes per lane, 1 KB per wave instruction, four per wave), straight
  // into the head of the exchange buffer in file order: no staging registers, no ds_write, and the request for tile i + 1 is
  // issued behind tile i's last exchange read, so it lands while the post-processing runs.  Every part of a launch is whole
  // (nout * sfactor <= npart): no bounds to check.
  constexpr uint32_t NCH = (PTS * 2) / 16, IMG_SKEW = 64;       // 16-byte pieces per thread (pieces 0, 1: part 0; 2, 3: part 1)
  auto fetch_image = [&](const uint64_t group) {
    const uint8_t* src = p.raw + (group << logTp) * (uint64_t)C * 4 + 16u * threadIdx.x;
    // (the second part's 16 KB lie IMG_SKEW bytes further on: the lanes of a first-stage read alternate between the two parts at
    //  the same offset -- the same LDS bank 16 KB apart)
#pragma unroll
    for (uint32_t r = 0; r < NCH; r++)
      lds_dma_b128(src + 16u * nt * r, lds_byte_addr((const uint8_t*)lds + 16u * ((threadIdx.x & ~63u) + r * nt) + (r >> 1) * IMG_SKEW));
  };
  auto dpp_pol = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, false)); };    // lane ^ 1
  auto dpp_part1 = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xEE, 0xf, 0xf, false)); };  // quad_perm [2,3,2,3]
  if (blockIdx.x < nitem) fetch_image((uint64_t)blockIdx.x * groups_per_out);      // (ltw_fill ended with a barrier)
  FB_ST_BEGIN(8);
  for (uint64_t item = blockIdx.x; item < nitem; item += gridDim.x) {
    v2f accK[8], accM[8];                                       // running sums: bins 256 k + (pa | pb); 256 (15 - k) + (pb | pa)
    float accH = 0.f;                                           // bin C/2 (lanes j = 0)
#pragma unroll
    for (int k = 0; k < 8; k++) accK[k] = accM[k] = (v2f){0.f, 0.f};
    for (uint32_t gi = 0; gi < groups_per_out; gi++) {
      const uint64_t group = item * groups_per_out + gi;
      asm volatile("" : "+v"(tid));
      cx2 x[NPAIR];
      FB_ST(8, 0);
      {
        // this tile's image has been requested one tile ago (or in front of the loop): each wave waits for its own pieces, the
        // barrier for everybody's; each thread then picks the 16 words of its first-stage butterflies.  The first exchange
        // write of the transform sits behind a barrier of its own (wgfft), which also ends these reads.
        const uint8_t* img = (const uint8_t*)lds;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < P::R1; i++) {
          uint32_t pl, n;
          elem(0, i, pl, n);
          const uint32_t w = tfp_word_aligned<CASPSR>(img + (pl << (LOGC + 2)) + pl * IMG_SKEW, 2 * n);     // (p0[2n], p1[2n], p0[2n+1], p1[2n+1])
          x[i].x = (v2f){__builtin_fmaf((float)(int8_t)(w & 0xff), sc2, hs2), __builtin_fmaf((float)(int8_t)((w >> 8) & 0xff), sc2, hs2)};
          x[i].y = (v2f){__builtin_fmaf((float)(int8_t)((w >> 16) & 0xff), sc2, hs2), __builtin_fmaf((float)(int8_t)(w >> 24), sc2, hs2)};
        }
      }
      FB_ST(8, 1);                               // image wait + decode
      const bool more = gi + 1 < groups_per_out || item + gridDim.x < nitem;
      const uint64_t next = gi + 1 < groups_per_out ? group + 1 : (item + gridDim.x) * groups_per_out;
      FB_ST(8, 2);
      auto post = [&](cx2 (&vv)[1][16]) {
        cx2 (&v)[16] = vv[0];
        __syncthreads();                         // every wave has read the last exchange: the next tile's image may land
        if (more) fetch_image(next);
        FB_ST(8, 3);                             // transform
        // one packed evaluation: (Z[k], mirror operands) of bins 256 k + (pa | pb) -> their powers and those of the mirror bins, added
        // to the running sums.  The accumulators start an output sample at zero (0 + x == x: the sums of TScrunch.C:193-200 bit for bit)
        auto eval = [&](const int k, const v2f mr, const v2f mi) {
          const v2f zr = v[k].x, zi = v[k].y;
          const v2f ar = zr + mr, ai = zi - mi;                       // A = Z[k] + conj Z[C-k]           (halved by the input scale)
          const v2f br = zi + mi, bi = mr - zr;                       // B = (Z[k] - conj Z[C-k]) / i
          const v2f wck = wc[k], wsk = ws[k];
          const v2f wr = wck * br + wsk * bi, wi = wck * bi - wsk * br;           // w^bin B,  w^bin = (c, -s)
          const v2f xr = ar + wr, xi = ai + wi, yr = ar - wr, yi = ai - wi;       // X[k] = A + w^k B,  X[C-k] = conj(A - w^k B)
          v2f pk = xr * xr; pk += xi * xi;                            // TFPFilterbank.C:56-59: Re^2 then += Im^2
          v2f pm = yr * yr; pm += yi * yi;
          if constexpr (PSC) {                                        // :79-80 pol sum BEFORE the time sum
            pk += (v2f){dpp_pol(pk[0]), dpp_pol(pk[1])};
            pm += (v2f){dpp_pol(pm[0]), dpp_pol(pm[1])};
          }
          // time order: part 0 of the tile (this quad's lanes 0, 1), then part 1 (lanes 2, 3)
          accK[k] = (accK[k] + pk) + (v2f){dpp_part1(pk[0]), dpp_part1(pk[1])};
          accM[k] = (accM[k] + pm) + (v2f){dpp_part1(pm[0]), dpp_part1(pm[1])};
        };
        if (threadIdx.x >= 64) {                 // (wave uniform)
          // mirror operands: the other half of v[15 - k]
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const cx2 g = v[15 - k];
            eval(k, __builtin_shufflevector(g.x, g.x, 1, 0), __builtin_shufflevector(g.y, g.y, 1, 0));
          }
        } else {
          // wave 0 holds the lanes j = 0 (columns p = 0 and p = 128, their own mirrors): low half Z[256 (16 - k)] (k = 0: Z[0]
          // itself), high half Z[256 (15 - k) + 128] (the same half), picked per lane
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const cx2 g = v[15 - k], sp = v[(16 - k) & 15];
            eval(k, (v2f){j0 ? sp.x[0] : g.x[1], j0 ? g.x[1] : g.x[0]}, (v2f){j0 ? sp.y[0] : g.y[1], j0 ? g.y[1] : g.y[0]});
          }
          // bin C/2 = 2048 = 256 * 8 + 0: its own mirror, w = -i: A = 2 Re Z, w B = -i 2 Im Z  (meaningful in lanes j = 0 only)
          const float xr = v[8].x[0] + v[8].x[0], xi = -(v[8].y[0] + v[8].y[0]);
          float ph = xr * xr; ph += xi * xi;
          if constexpr (PSC) ph += dpp_pol(ph);
          accH = (accH + ph) + dpp_part1(ph);
        }
      };
      wgfft<LOGC, -1, false, false, true>(lds, ltw_off, tid, logT, x, post);
      if (gi + 1 == groups_per_out && col < npol_out) {
        // lanes c = 0 (pol 0 / the pol sum) and, without pscrunch, c = 1 (pol 1) hold the output sample's sums
        // (offsets from an opaque copy of the position pair: they are loop invariant, and hoisted out of the tile loop the 33
        //  addresses spilled the kernel)
        uint32_t qa = pa, qb = pb;
        asm volatile("" : "+v"(qa), "+v"(qb));
        float* const o = p.out + item * (uint64_t)C * npol_out + col;
        const uint32_t oa = qa * npol_out, ob = qb * npol_out, ks = 256u * npol_out;     // bin 256 k + p at o[p * npol + k * ks]
#pragma unroll
        for (int k = 0; k < 8; k++) {
          o[oa + k * ks] = accK[k][0];
          o[ob + k * ks] = accK[k][1];
          if (!j0) {
            o[ob + (15 - k) * ks] = accM[k][0];
            o[oa + (15 - k) * ks] = accM[k][1];
          } else {
            if (k) o[(16 - k) * ks] = accM[k][0];
            o[ob + (15 - k) * ks] = accM[k][1];                // (pb = 128)
          }
        }
        if (j0) o[8 * ks] = accH;
      }
      FB_ST(8, 4);                               // split, powers, time scrunch, stores
      FB_ST_TILE(8, 5);
    }
  }
  FB_ST_END(8);
}

// The same post-processing on the last stage's registers for nchan = 512, 1024, 2048 and 8192 (digifil -F 512 ... 8192).  The last
// stage has radix RL = 2 / 4 / 8 / 2 and a thread holds H = 16 / RL butterfly pairs v[h][k]: pair q = H tid + h is column q % T,
// position pair (j, P - j) with j = q / T.  The tile's columns are (part, pol) = (c / 2, c % 2):
//   nchan 2048 (T = 8):  h = pol; the four lanes of a quad are the tile's four parts, all at position j = tid / 4
//   nchan 1024 (T = 16): h = 2 ph + pol; lane l of a quad holds parts 2 l + ph, position j = tid / 4
//   nchan 512  (T = 32): h = 2 ph + pol; lane l of a quad holds parts 4 l + ph
//   nchan 8192 (T = 2):  h = 2 jh + pol; one part per tile, four position pairs j = 4 tid + jh per thread
// so the polarisation sum is an add inside the thread and the time sum over the tile's parts a chain of quad broadcasts in time
// order (every lane of the quad ends up with the same sums; lane 0 of it stores).  Everything else -- pairing of the bins, twiddle
// arguments, half-scale decode, order of every rounding -- is k_tfp4k's, i.e. the generic kernel's: bit-identical outputs.
template <int LOGC, bool CASPSR, bool PSC>
__global__ __launch_bounds__(512) void k_tfpm(const TfpParams p, const cf* __restrict__ tw)
{
  typedef FftPlan<LOGC> P;
  static_assert(LOGC == 9 || LOGC == 10 || LOGC == 11 || LOGC == 13, "k_tfpm: 16 x 16 x {2, 4, 8} and 16 x 16 x 16 x 2");
  constexpr int logT = 14 - LOGC, logTp = logT - 1;
  constexpr int LOGRL = P::REM, RL = 1 << LOGRL, H = (PTS / RL) / 2, NK = RL / 2;
  constexpr int logP = LOGC - LOGRL;
  constexpr uint32_t nt = 512, C = 1u << LOGC, Pn = 1u << logP, T = 1u << logT, NPT = 1u << logTp;
  constexpr int JH = H > (int)T ? H / (int)T : 1;              // position pairs per thread
  constexpr int PH = (H / 2) / JH;                             // parts per thread
  constexpr uint32_t LP = NPT / PH;                            // lanes of a quad that hold different parts (1 or 4)
  static_assert(LP == 1 || LP == 4, "k_tfpm: the parts of a tile are one lane or the four lanes of a quad");
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  constexpr uint32_t npol_out = PSC ? 1 : 2;
  const uint64_t nout = p.npart / p.sfactor;
  const uint32_t ltw_off = lds_pad(PTS * nt) + 8;
  ltw_fill<LOGC>(lds, ltw_off, tw, tid, nt);
  // position pair jh of this thread: (pa, pb) = (j, P - j); j = 0 goes with P / 2
  auto jof = [&](const int jh) -> uint32_t { return JH == 1 ? threadIdx.x >> 2 : (uint32_t)JH * threadIdx.x + jh; };
  // w^(P k + pa), w^(P k + pb), w = exp(-i pi / C): argument bin / 2C revolutions, exact in float (the generic kernel's values)
  v2f wc[JH][NK], ws[JH][NK];
#pragma unroll
  for (int jh = 0; jh < JH; jh++) {
    const uint32_t j = jof(jh), pa = j, pb = j ? Pn - j : Pn / 2;
#pragma unroll
    for (int k = 0; k < NK; k++) {
      const float xa = (float)(Pn * k + pa) * (0.5f / (float)C), xb = (float)(Pn * k + pb) * (0.5f / (float)C);
      wc[jh][k] = (v2f){__builtin_amdgcn_cosf(xa), __builtin_amdgcn_cosf(xb)};
      ws[jh][k] = (v2f){__builtin_amdgcn_sinf(xa), __builtin_amdgcn_sinf(xb)};
    }
  }
  const float sc2 = 0.5f * p.scale, hs2 = 0.5f * sc2;           // half scale: see k_tfp4k
  const uint32_t groups_per_out = p.sfactor >> logTp;            // sfactor is a multiple of the tile's parts (host)
  const uint64_t nitem = nout;
  auto elem = [&](const int g2, const int i, uint32_t& pl, uint32_t& n) {
    const uint32_t e = first_stage_elem<LOGC>(tid, logT, g2, i);
    pl = (e & ((1u << logT) - 1)) >> 1;
    n = e >> logT;
  };
  // image by LDS-DMA as in k_tfp4k; part q of the tile lies q * IMG_SKEW bytes further on, so that the lanes of a first-stage
  // read -- consecutive lanes = consecutive parts at the same offset -- fall on different banks
  constexpr uint32_t NCH = (PTS * 2) / 16, IMG_SKEW = NPT > 1 ? 256u / NPT : 0u;
  auto fetch_image = [&](const uint64_t group) {
    const uint8_t* src = p.raw + (group << logTp) * (uint64_t)C * 4 + 16u * threadIdx.x;
#pragma unroll
    for (uint32_t r = 0; r < NCH; r++) {
      const uint32_t off = 16u * ((threadIdx.x & ~63u) + r * nt);                    // this wave's 1 KB of the tile
      lds_dma_b128(src + 16u * nt * r, lds_byte_addr((const uint8_t*)lds + off + (off >> (LOGC + 2)) * IMG_SKEW));
    }
  };
  auto bc0 = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x00, 0xf, 0xf, false)); };
  auto bc1 = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x55, 0xf, 0xf, false)); };
  auto bc2 = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xAA, 0xf, 0xf, false)); };
  auto bc3 = [](const float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xFF, 0xf, 0xf, false)); };
  if (blockIdx.x < nitem) fetch_image((uint64_t)blockIdx.x * groups_per_out);
  FB_ST_BEGIN(8);
  for (uint64_t item = blockIdx.x; item < nitem; item += gridDim.x) {
    // running sums of an output sample: [jh][pol][k] bins P k + (pa | pb) and their mirrors P (RL - 1 - k) + (pb | pa); bin C / 2
    v2f accK[JH][npol_out][NK], accM[JH][npol_out][NK];
    float accH[npol_out];
#pragma unroll
    for (int jh = 0; jh < JH; jh++)
#pragma unroll
      for (uint32_t q = 0; q < npol_out; q++)
#pragma unroll
        for (int k = 0; k < NK; k++) accK[jh][q][k] = accM[jh][q][k] = (v2f){0.f, 0.f};
#pragma unroll
    for (uint32_t q = 0; q < npol_out; q++) accH[q] = 0.f;
    for (uint32_t gi = 0; gi < groups_per_out; gi++) {
      const uint64_t group = item * groups_per_out + gi;
      asm volatile("" : "+v"(tid));
      cx2 x[NPAIR];
      FB_ST(8, 0);
      {
        const uint8_t* img = (const uint8_t*)lds;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < P::R1; i++) {
          uint32_t pl, n;
          elem(0, i, pl, n);
          const uint32_t w = tfp_word_aligned<CASPSR>(img + (pl << (LOGC + 2)) + pl * IMG_SKEW, 2 * n);
          x[i].x = (v2f){__builtin_fmaf((float)(int8_t)(w & 0xff), sc2, hs2), __builtin_fmaf((float)(int8_t)((w >> 8) & 0xff), sc2, hs2)};
          x[i].y = (v2f){__builtin_fmaf((float)(int8_t)((w >> 16) & 0xff), sc2, hs2), __builtin_fmaf((float)(int8_t)(w >> 24), sc2, hs2)};
        }
      }
      FB_ST(8, 1);
      const bool more = gi + 1 < groups_per_out || item + gridDim.x < nitem;
      const uint64_t next = gi + 1 < groups_per_out ? group + 1 : (item + gridDim.x) * groups_per_out;
      FB_ST(8, 2);
      auto post = [&](cx2 (&v)[H][RL]) {
        __syncthreads();                         // every wave has read the last exchange: the next tile's image may land
        if (more) fetch_image(next);
        FB_ST(8, 3);
        const bool wave0 = threadIdx.x < 64;     // (wave uniform) only wave 0 holds position pairs with j = 0
#pragma unroll
        for (int jh = 0; jh < JH; jh++) {
          const bool j0 = wave0 && jof(jh) == 0;
          // powers of this thread's parts (ph) and polarisations for the bins of position pair jh
          v2f pk[PH][2][NK], pm[PH][2][NK];
          float ph2[PH][2];
#pragma unroll
          for (int ph = 0; ph < PH; ph++)
#pragma unroll
            for (int pol = 0; pol < 2; pol++) {
              cx2 (&u)[RL] = v[(jh * PH + ph) * 2 + pol];
#pragma unroll
              for (int k = 0; k < NK; k++) {
                const cx2 g = u[RL - 1 - k], sp = u[(RL - k) & (RL - 1)];
                v2f mr, mi;
                if (wave0) {                     // lanes j = 0: low half Z[P (RL - k)] (k = 0: Z[0] itself), high half Z[P (RL-1-k) + P/2]
                  mr = (v2f){j0 ? sp.x[0] : g.x[1], j0 ? g.x[1] : g.x[0]};
                  mi = (v2f){j0 ? sp.y[0] : g.y[1], j0 ? g.y[1] : g.y[0]};
                } else {
                  mr = __builtin_shufflevector(g.x, g.x, 1, 0);
                  mi = __builtin_shufflevector(g.y, g.y, 1, 0);
                }
                const v2f zr = u[k].x, zi = u[k].y;
                const v2f ar = zr + mr, ai = zi - mi;
                const v2f br = zi + mi, bi = mr - zr;
                const v2f wck = wc[jh][k], wsk = ws[jh][k];
                const v2f wr = wck * br + wsk * bi, wi = wck * bi - wsk * br;
                const v2f xr = ar + wr, xi = ai + wi, yr = ar - wr, yi = ai - wi;
                v2f a = xr * xr; a += xi * xi;                    // TFPFilterbank.C:56-59: Re^2 then += Im^2
                v2f b = yr * yr; b += yi * yi;
                pk[ph][pol][k] = a; pm[ph][pol][k] = b;
              }
              // bin C / 2 = P (RL / 2) + 0: its own mirror, w = -i  (meaningful in the lane j = 0 only)
              const float hr = u[RL / 2].x[0] + u[RL / 2].x[0], hi = -(u[RL / 2].y[0] + u[RL / 2].y[0]);
              float hh = hr * hr; hh += hi * hi;
              ph2[ph][pol] = hh;
            }
          if constexpr (PSC) {                   // :79-80 pol sum BEFORE the time sum
#pragma unroll
            for (int ph = 0; ph < PH; ph++) {
#pragma unroll
              for (int k = 0; k < NK; k++) { pk[ph][0][k] += pk[ph][1][k]; pm[ph][0][k] += pm[ph][1][k]; }
              ph2[ph][0] += ph2[ph][1];
            }
          }
          // time sum, parts in time order: part = PH * lane + ph
          auto tsum = [&](float& acc, const float (&val)[PH]) {
            if constexpr (LP == 1) {
#pragma unroll
              for (int ph = 0; ph < PH; ph++) acc += val[ph];
            } else {
#pragma unroll
              for (int ph = 0; ph < PH; ph++) acc += bc0(val[ph]);
#pragma unroll
              for (int ph = 0; ph < PH; ph++) acc += bc1(val[ph]);
#pragma unroll
              for (int ph = 0; ph < PH; ph++) acc += bc2(val[ph]);
#pragma unroll
              for (int ph = 0; ph < PH; ph++) acc += bc3(val[ph]);
            }
          };
#pragma unroll
          for (uint32_t q = 0; q < npol_out; q++) {
#pragma unroll
            for (int k = 0; k < NK; k++)
#pragma unroll
              for (int e = 0; e < 2; e++) {
                float vk[PH], vm[PH];
#pragma unroll
                for (int ph = 0; ph < PH; ph++) { vk[ph] = pk[ph][q][k][e]; vm[ph] = pm[ph][q][k][e]; }
                float ak = accK[jh][q][k][e], am = accM[jh][q][k][e];
                tsum(ak, vk); tsum(am, vm);
                accK[jh][q][k][e] = ak; accM[jh][q][k][e] = am;
              }
            if (jh == 0) {
              float vh[PH];
#pragma unroll
              for (int ph = 0; ph < PH; ph++) vh[ph] = ph2[ph][q];
              tsum(accH[q], vh);
            }
          }
        }
      };
      wgfft<LOGC, -1, false, false, true>(lds, ltw_off, tid, logT, x, post);
      if (gi + 1 == groups_per_out && (LP == 1 || (threadIdx.x & 3u) == 0)) {
        // every lane of a quad holds the same sums: lane 0 stores (TFP order: out[sample][bin][pol])
        float* const o = p.out + item * (uint64_t)C * npol_out;
#pragma unroll
        for (int jh = 0; jh < JH; jh++) {
          uint32_t j = jof(jh);
          asm volatile("" : "+v"(j));            // (loop-invariant addresses: keep them out of the tile loop's registers)
          const bool j0 = j == 0;
          const uint32_t pa = j, pb = j ? Pn - j : Pn / 2;
#pragma unroll
          for (uint32_t q = 0; q < npol_out; q++)
#pragma unroll
            for (int k = 0; k < NK; k++) {
              o[(Pn * k + pa) * npol_out + q] = accK[jh][q][k][0];
              o[(Pn * k + pb) * npol_out + q] = accK[jh][q][k][1];
              if (!j0) {
                o[(Pn * (RL - 1 - k) + pb) * npol_out + q] = accM[jh][q][k][0];
                o[(Pn * (RL - 1 - k) + pa) * npol_out + q] = accM[jh][q][k][1];
              } else {
                if (k) o[(Pn * (RL - k)) * npol_out + q] = accM[jh][q][k][0];
                o[(Pn * (RL - 1 - k) + pb) * npol_out + q] = accM[jh][q][k][1];            // (pb = P / 2)
              }
            }
          if (jh == 0 && j0) {
#pragma unroll
            for (uint32_t q = 0; q < npol_out; q++) o[(C / 2) * npol_out + q] = accH[q];
          }
        }
      }
      FB_ST(8, 4);
      FB_ST_TILE(8, 5);
    }
  }
  FB_ST_END(8);
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

FB_ST_READER(tfp)

extern "C" int dspsr_amd_tfp_filterbank(dspsr_amd_ctx* ctx, const dspsr_amd_tfp_config* cfg, const int8_t* raw_dev,
                                        int raw_layout, float scale, float* out_dev, uint64_t npart)
{
  if (!ctx || !cfg || !raw_dev || !out_dev) return DSPSR_AMD_EINVAL;
  const uint32_t nchan = cfg->nchan;
  if (nchan < 16 || (nchan & (nchan - 1)) || nchan > 8192)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tfp_filterbank: nchan=%u must be a power of two in [16, 8192]", nchan);
  if (cfg->npol != 2)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_tfp_filterbank: only real dual-polarisation 8-bit input is built (npol=%u)", cfg->npol);
  const uint32_t sf = cfg->tscrunch ? cfg->tscrunch : 1;
  int logC = 0;                                       // nchan complex points per polarisation and part (see k_tfp)
  while ((1u << logC) < nchan) logC++;
  const int logT = 14 - logC;                         // columns per workgroup tile = 2 polarisations x T parts
  const uint32_t T = 1u << (logT - 1);
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
  const bool coal = ((uintptr_t)raw_dev & 15) == 0;
  ktfp_t k = pick_tfp(logC, p.caspsr != 0, coal, mkseq_t<14>::type());
  // digifil's own geometry (-F 4096 with an even -t): post-processing on the last stage's registers, see k_tfp4k
  if (nchan == 4096 && coal && (sf % 2) == 0)
    k = p.caspsr ? (p.pscrunch ? k_tfp4k<true, true> : k_tfp4k<true, false>) : (p.pscrunch ? k_tfp4k<false, true> : k_tfp4k<false, false>);
  // its generalisation to -F 512 / 1024 / 2048 / 8192 (tscrunch a multiple of the tile's 16 / 8 / 4 / 1 parts), see k_tfpm
#define TFPM(L) (p.caspsr ? (p.pscrunch ? k_tfpm<L, true, true> : k_tfpm<L, true, false>) : (p.pscrunch ? k_tfpm<L, false, true> : k_tfpm<L, false, false>))
  if (coal && (sf % T) == 0) {
    if (nchan == 512) k = TFPM(9);
    else if (nchan == 1024) k = TFPM(10);
    else if (nchan == 2048) k = TFPM(11);
    else if (nchan == 8192) k = TFPM(13);
  }
#undef TFPM
  const size_t lds = lds_total_words_host(16384, logC) * sizeof(cf);
  hipError_t e = dspsr_amd_allow_lds((const void*)k, lds);      // raised once per kernel, not per call
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_tfp_filterbank: %s", hipGetErrorString(e));
  const uint64_t nitem = sf > T ? nout : (nout * sf + T - 1) >> (logT - 1);
  const uint32_t ncu = ctx->ncu;
  const uint32_t grid = (uint32_t)(nitem < ncu ? nitem : ncu);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, ctx->stream, p, ctx->tw);
  e = hipGetLastError();
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_tfp_filterbank: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}
