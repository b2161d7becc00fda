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
#include "engine_internal.h"

namespace dspsr_amd {

struct FbGeom {
  int logM, logR, logT1, logT2, logT3;
  int real_input, npol;
  uint32_t C, nfilt_pos, nkeep;
};

struct FbIn {
  int kind;  // 0: float32 rows, 1: int8 generic, 2: int8 caspsr
  const void* base;
  uint64_t pol_stride;  // float32: floats between pol rows
  uint64_t part_step;   // time samples between parts
  uint32_t nchan, ichan;
  float scale;
};

struct FbOut {
  int kind;  // 0: none (benchmark), 1: complex filterbank rows, 2: detected
  float* base;
  uint64_t chan_stride, pol_stride, part_step;  // floats
  int state;                                    // detected: coherence / stokes
  uint32_t ndim, chan0;
};

DEV float cvt8(int v, float scale) { return ((float)v + 0.5f) * scale; }

DEV cf load_sample(const FbGeom& g, const FbIn& in, uint32_t seq, uint64_t t)
{
  if (g.real_input) {
    if (in.kind == 0) {
      const float* x = (const float*)in.base;
      return make_float2(x[t], g.npol == 2 ? x[in.pol_stride + t] : 0.0f);
    }
    if (in.kind == 1) {
      const int8_t* r = (const int8_t*)in.base + (t * in.nchan + in.ichan) * g.npol;
      if (g.npol == 2) {
        if (((uintptr_t)r & 1) == 0) {
          const uint16_t u = *(const uint16_t*)r;
          return make_float2(cvt8((int8_t)(u & 0xff), in.scale), cvt8((int8_t)(u >> 8), in.scale));
        }
        return make_float2(cvt8(r[0], in.scale), cvt8(r[1], in.scale));
      }
      return make_float2(cvt8(r[0], in.scale), 0.0f);
    }
    const int8_t* r = (const int8_t*)in.base + (t >> 2) * 8 + (t & 3);
    return make_float2(cvt8(r[0], in.scale), cvt8(r[4], in.scale));
  }
  if (in.kind == 0) {
    const float2* x = (const float2*)((const float*)in.base + seq * in.pol_stride);
    return x[t];
  }
  const int8_t* r = (const int8_t*)in.base + ((t * in.nchan + in.ichan) * g.npol + seq) * 2;
  return make_float2(cvt8(r[0], in.scale), cvt8(r[1], in.scale));
}

// exp(-2*pi*i*j/2^logL), j < 2^logL, from exact float arguments
DEV cf twiddle_big(uint64_t j, int logL)
{
  float s, c;
  if (logL <= 24) {
    sincospif(-2.0f * (float)(uint32_t)j / (float)(1u << logL), &s, &c);
    return make_float2(c, s);
  }
  const uint32_t hi = (uint32_t)(j >> 12), lo = (uint32_t)(j & 4095);
  float s2, c2;
  sincospif(-2.0f * (float)hi / (float)(1ull << (logL - 12)), &s, &c);
  sincospif(-2.0f * (float)lo / (float)(1ull << logL), &s2, &c2);
  return cmul(make_float2(c, s), make_float2(c2, s2));
}

// ------------------------------------------------------------------------------------ P1
template <int LOGF>
__global__ __launch_bounds__(1024) void k_fwd_cols(const FbGeom g, const FbIn in, cf* __restrict__ A,
                                                   const cf* __restrict__ tw, const uint64_t part0)
{
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const int logT = g.logT1, logT2 = g.logT2;
  const uint32_t T = 1u << logT, T2 = 1u << logT2;
  const uint32_t tile = blockIdx.x, seq = blockIdx.y, nseq = gridDim.y;
  const uint64_t part = blockIdx.z;
  const int logL = g.logM + g.logR;
  const uint64_t L = 1ull << logL;
  const uint64_t t0 = (part0 + part) * in.part_step;

#pragma unroll 4
  for (int j = 0; j < 16; j++) {
    const uint32_t q = tid + nt * j;
    const uint32_t col = q & (T - 1), na = q >> logT;
    const uint64_t n = ((uint64_t)na << g.logR) + tile * T + col;
    lds[lds_pad(q)] = load_sample(g, in, seq, t0 + n);
  }
  __syncthreads();
  wgfft<LOGF, -1>(lds, tid, nt, logT, tw);

  cf* __restrict__ Aseq = A + (part * nseq + seq) * L;
#pragma unroll 4
  for (int j = 0; j < 16; j++) {
    const uint32_t q = tid + nt * j;
    const uint32_t klo = q & (T2 - 1), col = (q >> logT2) & (T - 1), khi = q >> (logT2 + logT);
    const uint32_t ka = (khi << logT2) | klo;
    const uint32_t nb = tile * T + col;
    cf v = lds[lds_pad((ka << logT) | col)];
    v = cmul(v, twiddle_big(((uint64_t)nb * ka) & (L - 1), logL));
    Aseq[(((((uint64_t)khi) << g.logR) + nb) << logT2) | klo] = v;
  }
}

// ------------------------------------------------------------------------------------ P2
template <int LOGF>
__global__ __launch_bounds__(1024) void k_fwd_rows(const FbGeom g, const cf* __restrict__ A, cf* __restrict__ X,
                                                   const cf* __restrict__ tw)
{
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const int logT = g.logT2, logT3 = g.logT3;
  const uint32_t T2 = 1u << logT, T3 = 1u << logT3;
  const uint32_t tile = blockIdx.x, seq = blockIdx.y, nseq = gridDim.y;
  const uint64_t part = blockIdx.z;
  const uint64_t L = 1ull << (g.logM + g.logR);
  const cf* __restrict__ Ablk = A + (part * nseq + seq) * L + (((uint64_t)tile << g.logR) << logT);

#pragma unroll 4
  for (int j = 0; j < 16; j++) {
    const uint32_t q = tid + nt * j;
    lds[lds_pad(q)] = Ablk[q];
  }
  __syncthreads();
  wgfft<LOGF, -1>(lds, tid, nt, logT, tw);

  cf* __restrict__ Xseq = X + (part * nseq + seq) * L;
#pragma unroll 4
  for (int j = 0; j < 16; j++) {
    const uint32_t q = tid + nt * j;
    const uint32_t slo = q & (T3 - 1), klo = (q >> logT3) & (T2 - 1), shi = q >> (logT3 + logT);
    const uint32_t srow = (shi << logT3) | slo;
    const cf v = lds[lds_pad((srow << logT) | klo)];
    Xseq[(((((uint64_t)shi) << g.logM) + tile * T2 + klo) << logT3) | slo] = v;
  }
}

// ------------------------------------------------------------------------------------ P3
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

template <int LOGF>
__global__ __launch_bounds__(1024) void k_inv_chan(const FbGeom g, const cf* __restrict__ X,
                                                   const cf* __restrict__ kernel, const FbOut out,
                                                   const cf* __restrict__ tw, const uint64_t part0)
{
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const int logT3 = g.logT3;
  const int logT = logT3 + (g.npol == 2 ? 1 : 0);
  const uint32_t T3 = 1u << logT3, M = 1u << g.logM, Rr = 1u << g.logR;
  const uint32_t tile = blockIdx.x;
  const uint64_t part = blockIdx.z;
  const uint64_t L = (uint64_t)M << g.logR;
  const uint32_t nseq = g.real_input ? 1 : g.npol;
  const cf* __restrict__ X0s = X + part * nseq * L;
  const uint64_t blk = ((uint64_t)M) << logT3;  // elements per X block

  for (uint32_t q = tid; q < (M << logT3); q += nt) {
    const uint32_t slo = q & (T3 - 1), m = q >> logT3;
    const uint32_t s = tile * T3 + slo;
    cf x0, x1;
    if (g.real_input) {
      const cf a = X0s[tile * blk + q];
      cf b;
      if (m > 0) {
        const uint32_t mblk = (Rr >> logT3) - 1 - tile;
        b = X0s[mblk * blk + (((uint64_t)(M - m)) << logT3) + (T3 - 1 - slo)];
      } else {
        const uint32_t r = (Rr - s) & (Rr - 1);
        b = X0s[(r >> logT3) * blk + (r & (T3 - 1))];
      }
      // W[k] = X0[k] + i X1[k] ; conj(W[L-k]) = X0[k] - i X1[k]
      x0 = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
      x1 = make_float2(0.5f * (a.y + b.y), 0.5f * (b.x - a.x));
    } else {
      x0 = X0s[tile * blk + q];
      x1 = g.npol == 2 ? X0s[L + tile * blk + q] : make_float2(0.f, 0.f);
    }
    if (kernel) {
      const cf k = kernel[((uint64_t)s << g.logM) + m];
      x0 = cmul(k, x0);
      x1 = cmul(k, x1);
    }
    if (g.npol == 2) {
      float4* dst = (float4*)&lds[lds_pad((m << logT) | (slo << 1))];
      *dst = make_float4(x0.x, x0.y, x1.x, x1.y);
    } else {
      lds[lds_pad((m << logT) | slo)] = x0;
    }
  }
  __syncthreads();
  wgfft<LOGF, +1>(lds, tid, nt, logT, tw);

  if (out.kind == 0) return;
  for (uint32_t slo = 0; slo < T3; slo++) {
    const uint32_t chan = out.chan0 + tile * T3 + slo;
    float* __restrict__ row = out.base + chan * out.chan_stride;
    for (uint32_t t = tid; t < g.nkeep; t += nt) {
      const uint32_t pos = g.nfilt_pos + t;
      cf p, q = make_float2(0.f, 0.f);
      if (g.npol == 2) {
        const float4 pq = *(const float4*)&lds[lds_pad((pos << logT) | (slo << 1))];
        p = make_float2(pq.x, pq.y);
        q = make_float2(pq.z, pq.w);
      } else {
        p = lds[lds_pad((pos << logT) | slo)];
      }
      if (out.kind == 1) {
        float2* o = (float2*)(row + (part0 + part) * out.part_step) + t;
        o[0] = p;
        if (g.npol == 2) ((float2*)((float*)o + out.pol_stride))[0] = q;
      } else {
        float r[4];
        detect4(p, q, out.state, r);
        const uint64_t idat = (part0 + part) * g.nkeep + t;
        if (out.ndim == 4) {
          ((float4*)row)[idat] = make_float4(r[0], r[1], r[2], r[3]);
        } else if (out.ndim == 2) {
          ((float2*)row)[idat] = make_float2(r[0], r[1]);
          ((float2*)(row + out.pol_stride))[idat] = make_float2(r[2], r[3]);
        } else {
          row[idat] = r[0];
          row[out.pol_stride + idat] = r[1];
          row[2 * out.pol_stride + idat] = r[2];
          row[3 * out.pol_stride + idat] = r[3];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ host
typedef void (*k1_t)(FbGeom, FbIn, cf*, const cf*, uint64_t);
typedef void (*k2_t)(FbGeom, const cf*, cf*, const cf*);
typedef void (*k3_t)(FbGeom, const cf*, const cf*, FbOut, const cf*, uint64_t);

template <int... I> struct iseq {};
template <int N, int... I> struct mkseq : mkseq<N - 1, N - 1, I...> {};
template <int... I> struct mkseq<0, I...> { typedef iseq<I...> type; };

template <int... I> static k1_t pick1(int logf, iseq<I...>) { static const k1_t t[] = {k_fwd_cols<I>...}; return t[logf]; }
template <int... I> static k2_t pick2(int logf, iseq<I...>) { static const k2_t t[] = {k_fwd_rows<I>...}; return t[logf]; }
template <int... I> static k3_t pick3(int logf, iseq<I...>) { static const k3_t t[] = {k_inv_chan<I>...}; return t[logf]; }

constexpr int MAX_LOGF = 14;
constexpr int LOG_POINTS = 14;  // points per workgroup (16 per thread, 1024 threads)

static inline int ilog2(uint64_t v) { int l = 0; while ((1ull << l) < v) l++; return l; }
static inline bool ispow2(uint64_t v) { return v && !(v & (v - 1)); }

struct dspsr_amd_filterbank_impl {
  dspsr_amd_ctx* ctx;
  dspsr_amd_filterbank_config cfg;
  FbGeom g;
  uint64_t N, L;
  uint32_t nseq, max_parts;
  uint32_t nt1, nt2, nt3;
  size_t lds1, lds2, lds3;
  cf* A = nullptr;
  cf* X = nullptr;
  cf* kernel = nullptr;
  bool kernel_set = false;
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

extern "C" int dspsr_amd_filterbank_create(dspsr_amd_ctx* ctx, const dspsr_amd_filterbank_config* cfg,
                                           dspsr_amd_filterbank** out)
{
  if (!ctx || !cfg || !out) return DSPSR_AMD_EINVAL;
  *out = nullptr;
  if (cfg->npol != 1 && cfg->npol != 2)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: npol=%u not 1 or 2", cfg->npol);
  if (!ispow2(cfg->freq_res) || cfg->freq_res < 2)
    return fb_fail(ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_create: freq_res=%u must be a power of two >= 2 "
                   "(freq_res=1 is the non-convolving filterbank, not built yet)", cfg->freq_res);
  if (!ispow2(cfg->nchan_subband))
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nchan_subband=%u must be a power of two",
                   cfg->nchan_subband);
  if (cfg->nfilt_pos + cfg->nfilt_neg >= cfg->freq_res)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nfilt_pos+nfilt_neg=%u >= freq_res=%u",
                   cfg->nfilt_pos + cfg->nfilt_neg, cfg->freq_res);
  if (cfg->input_nchan == 0) return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: input_nchan=0");

  dspsr_amd_filterbank* fb = new dspsr_amd_filterbank;
  fb->ctx = ctx;
  fb->cfg = *cfg;
  FbGeom& g = fb->g;
  const uint64_t M = cfg->freq_res, C = cfg->nchan_subband;
  fb->N = C * M;
  fb->L = cfg->real_input ? 2 * fb->N : fb->N;
  const uint64_t Rr = fb->L / M;
  g.logM = ilog2(M);
  g.logR = ilog2(Rr);
  g.real_input = cfg->real_input ? 1 : 0;
  g.npol = cfg->npol;
  g.C = (uint32_t)C;
  g.nfilt_pos = cfg->nfilt_pos;
  g.nkeep = cfg->freq_res - cfg->nfilt_pos - cfg->nfilt_neg;
  fb->nseq = cfg->real_input ? 1 : cfg->npol;
  if (g.logM > MAX_LOGF || g.logR > MAX_LOGF) {
    delete fb;
    return fb_fail(ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_create: freq_res=%llu / spectrum rows=%llu exceed the single-pass "
                   "limit 2^%d", (unsigned long long)M, (unsigned long long)Rr, MAX_LOGF);
  }
  // tiles: every workgroup holds min(2^14, available) points = 16 per thread
  g.logT1 = g.logR < LOG_POINTS - g.logM ? g.logR : LOG_POINTS - g.logM;
  g.logT2 = g.logM < LOG_POINTS - g.logR ? g.logM : LOG_POINTS - g.logR;
  const int logC = ilog2(C), logPol = cfg->npol == 2 ? 1 : 0;
  int t3 = LOG_POINTS - g.logM - logPol;
  if (t3 < 0) t3 = 0;
  g.logT3 = logC < t3 ? logC : t3;
  const uint64_t p1 = M << g.logT1, p2 = Rr << g.logT2, p3 = (M << g.logT3) << logPol;
  if (p1 < 16 || p2 < 16 || p3 < 16 || p3 > (1u << LOG_POINTS)) {
    delete fb;
    return fb_fail(ctx, DSPSR_AMD_EINVAL,
                   "dspsr_amd_filterbank_create: problem too small/large for the workgroup tiling "
                   "(points per pass %llu/%llu/%llu, need 16..16384)",
                   (unsigned long long)p1, (unsigned long long)p2, (unsigned long long)p3);
  }
  fb->nt1 = (uint32_t)(p1 / 16);
  fb->nt2 = (uint32_t)(p2 / 16);
  fb->nt3 = (uint32_t)(p3 / 16);
  fb->lds1 = lds_words_host((uint32_t)p1) * sizeof(cf);
  fb->lds2 = lds_words_host((uint32_t)p2) * sizeof(cf);
  fb->lds3 = lds_words_host((uint32_t)p3) * sizeof(cf);
  fb->max_parts = cfg->max_parts ? cfg->max_parts : 1;
  const size_t scratch = (size_t)fb->max_parts * fb->nseq * fb->L * sizeof(cf);
  if (hipMalloc((void**)&fb->A, scratch) != hipSuccess || hipMalloc((void**)&fb->X, scratch) != hipSuccess) {
    if (fb->A) (void)hipFree(fb->A);
    delete fb;
    return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_create: hipMalloc of 2 x %zu scratch bytes failed",
                   scratch);
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
  hipError_t e = hipMemcpyAsync(fb->kernel, kernel_host, expect * sizeof(cf), hipMemcpyHostToDevice, fb->ctx->stream);
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
  if (nkeep) *nkeep = fb->g.nkeep;
  return DSPSR_AMD_OK;
}

template <typename K> static hipError_t allow_lds(K kern, size_t bytes)
{
  return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static int fb_run(dspsr_amd_filterbank* fb, FbIn in, FbOut out, uint64_t npart, uint64_t in_chan_stride_bytes_or_floats)
{
  dspsr_amd_ctx* ctx = fb->ctx;
  if (!fb->kernel_set)
    return fb_fail(ctx, DSPSR_AMD_ESTATE, "dspsr_amd_filterbank_perform: set_kernel (Engine::setup) not called");
  if (npart == 0) return DSPSR_AMD_OK;
  const FbGeom& g = fb->g;
  typedef mkseq<MAX_LOGF + 1>::type seq_t;
  k1_t k1 = pick1(g.logM, seq_t());
  k2_t k2 = pick2(g.logR, seq_t());
  k3_t k3 = pick3(g.logM, seq_t());
  hipError_t e;
  if ((e = allow_lds(k1, fb->lds1)) != hipSuccess || (e = allow_lds(k2, fb->lds2)) != hipSuccess ||
      (e = allow_lds(k3, fb->lds3)) != hipSuccess)
    return fb_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_perform: hipFuncSetAttribute: %s", hipGetErrorString(e));
  const uint32_t Rr = 1u << g.logR, M = 1u << g.logM;
  const float* in_f32 = (const float*)in.base;
  for (uint32_t ichan = 0; ichan < fb->cfg.input_nchan; ichan++) {
    FbIn ci = in;
    if (in.kind == 0) ci.base = in_f32 + ichan * in_chan_stride_bytes_or_floats;
    ci.ichan = ichan;
    ci.nchan = fb->cfg.input_nchan;
    FbOut co = out;
    co.chan0 = ichan * g.C;
    const cf* kern = fb->kernel ? fb->kernel + (uint64_t)ichan * fb->N : nullptr;
    for (uint64_t part0 = 0; part0 < npart; part0 += fb->max_parts) {
      const uint32_t nb = (uint32_t)((npart - part0) < fb->max_parts ? (npart - part0) : fb->max_parts);
      hipLaunchKernelGGL(k1, dim3(Rr >> g.logT1, fb->nseq, nb), dim3(fb->nt1), fb->lds1, ctx->stream, g, ci, fb->A,
                         ctx->tw, part0);
      hipLaunchKernelGGL(k2, dim3(M >> g.logT2, fb->nseq, nb), dim3(fb->nt2), fb->lds2, ctx->stream, g, fb->A, fb->X,
                         ctx->tw);
      hipLaunchKernelGGL(k3, dim3(g.C >> g.logT3, 1, nb), dim3(fb->nt3), fb->lds3, ctx->stream, g, fb->X, kern, co,
                         ctx->tw, part0);
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
  if (out_dev && out_step < 2ull * fb->g.nkeep)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: out_step=%llu < 2*nkeep=%u",
                   (unsigned long long)out_step, 2 * fb->g.nkeep);
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
  if (raw_layout != DSPSR_AMD_RAW_CASPSR && raw_layout != DSPSR_AMD_RAW_GENERIC)
    return fb_fail(fb->ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_raw: unknown raw layout %d", raw_layout);
  uint64_t step;
  dspsr_amd_filterbank_sizes(fb, nullptr, nullptr, &step, nullptr);
  FbIn in = {raw_layout == DSPSR_AMD_RAW_CASPSR ? 2 : 1, raw_dev, 0, step, fb->cfg.input_nchan, 0, scale};
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
    in = {0, in_f32_dev, in_pol_stride, in_step / idim, fb->cfg.input_nchan, 0, 1.0f};
  } else {
    if (raw_layout == DSPSR_AMD_RAW_CASPSR &&
        !(fb->cfg.real_input && fb->cfg.npol == 2 && fb->cfg.input_nchan == 1))
      return fb_fail(fb->ctx, DSPSR_AMD_EINVAL,
                     "dspsr_amd_filterbank_perform_detect: CASPSR layout needs real dual-pol single-channel input");
    in = {raw_layout == DSPSR_AMD_RAW_CASPSR ? 2 : 1, raw_dev, 0, step, fb->cfg.input_nchan, 0, scale};
  }
  FbOut out = {2, det_dev, det_chan_stride, det_pol_stride, 0, state, ndim, 0};
  return fb_run(fb, in, out, npart, in_chan_stride);
}
