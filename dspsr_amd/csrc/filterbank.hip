// Convolving filterbank (dsp::Filterbank -F N:D), host side: geometry, kernel choice and the launch sequence per call.
// Kernels: fb_fwd_cols.hip, fb_fwd_rows.hip, fb_inv_chan*.hip, fb_two_pass.hip, fb_four_pass.hip; shared: fb_common.h
#include "fb_common.h"

namespace dspsr_amd {

constexpr int LOG_POINTS_DEFAULT = 14;  // points per workgroup (32 per thread, 512 threads)

static inline int ilog2(uint64_t v) { int l = 0; while ((1ull << l) < v) l++; return l; }
static inline bool ispow2(uint64_t v) { return v && !(v & (v - 1)); }

struct dspsr_amd_filterbank_impl {
  dspsr_amd_ctx* ctx;
  dspsr_amd_filterbank_config cfg;
  FbGeom g;
  uint64_t N, L;
  uint32_t nseq, max_parts;
  uint32_t nt1, nt2, nt3, nt4 = 0, ncu, wg3 = 1, wg1 = 1;
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
  k1_t k1_w1 = nullptr, k1_w4 = nullptr;                     // pass 1: one word per sample pair / generic loads
  k2_t k2 = nullptr;
  k3_t k3 = nullptr, k3f = nullptr, k3s = nullptr;           // inverse pass: plain, fused fold, search mode (FbOut kind 5)
  k3a_t k3a = nullptr;
  k3b_t k3b = nullptr;
  float* fpart = nullptr;    // segmented fused fold: partial profiles of the part runs 1 .. nseg-1 of a launch
  size_t fpart_floats = 0;
  uint32_t plan_cap = 0;     // fused fold: plan entries per LDS buffer behind the twiddle tables
  size_t lds3f = 0;          // dynamic LDS of the fused inverse pass
  // two-pass path of short responses (complex dual-pol 8-bit input, nchan_subband * freq_res^2 == 2^27): see fb_two_pass.hip
  uint8_t* dsub = nullptr;    // nsub > 1: the launch group's samples de-interleaved into nsub blocks (k_sub_split)
  size_t dsub_bytes = 0;
  bool two_pass = false;
  FbGeom g1t;                 // ... pass 1 of Fa < 2^14 through k_raw_transpose + k_fwd_cols: their geometry (M = Fa, Rr = Fb, T2 = freq_res)
  k1_t k1t = nullptr;
  uint32_t nt1t = 0;
  size_t lds1t = 0;
  k1c_t k1c = nullptr;
  k3_t k2r = nullptr, k2rf = nullptr, k2rs = nullptr;
  size_t lds1c = 0, lds2r = 0, lds2rf = 0;
  uint32_t plan_cap2 = 0;
  k3b_t k3bf = nullptr;      // four-pass fused fold: second inverse pass that leaves segment sums (FbOut kind 4)
  float* msum = nullptr;     // ... [chan][part][tile][t2][2] float4 of one input channel's sub-band and one block
  size_t msum_floats = 0;
  // freq_res = 3 * 2^k / 5 * 2^k (msub = 3, 5; see k_time_combine): g and everything above describe the INNER filterbank of
  // nchan_subband * msub pseudo-channels with freq_res / msub bins and the whole transform kept; cfg and these the caller's
  uint32_t msub = 0, out_C = 0, out_M = 0, out_nfilt_pos = 0, out_nkeep = 0;
  dspsr_amd_filterbank* batch = nullptr;   // dsp::Convolution on many channels (nchan_subband = 1, complex float rows): the inverse passes of a
                             // filterbank of `batch->cfg.nchan_subband` channels per group, see fb_run_batched
  int conv1_logM = -1;       // >= 0: dsp::Convolution shapes with n_fft <= 8192 on complex float rows run in ONE tile pass (fb_conv1.hip)
  int conv3_logM = -1;       // >= 0: ... with 2^14 <= n_fft <= 2^21 in THREE tile passes (fb_conv3.hip), launch groups of conv3_ch channels x
  uint32_t conv3_ch = 0, conv3_parts = 0;    //   conv3_parts parts through the scratch blocks S1 / S2
  cf* S1 = nullptr;
  cf* S2 = nullptr;
  uint64_t conv3_bytes = 0;                  //   bytes of each (allocated on first use; a failed allocation turns the path off)
  cf* kernel_nat = nullptr;  // ... its response in natural order when `kernel` is stored in the blocked order of the four-pass kernels
  int plain_logC = -1;       // >= 0: freq_res = 1, the non-convolving filterbank (fb_plain.hip): no scratch, one launch per call
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
  // (odd factors up to ODD_MAX = 127 of either length; both lengths at once as long as the product of the two factors stays within it)
  auto odd_part = [](uint32_t v) { while (v && !(v & 1)) v >>= 1; return v; };
  if (cfg->freq_res == 1) {
    // the non-convolving filterbank (Filterbank.C:614-623, `dspsr -F N`): nchan_subband-point forward transforms, bin k of a part
    // is the part's one output sample of channel k
    if (cfg->nfilt_pos || cfg->nfilt_neg)
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nfilt_pos+nfilt_neg=%u >= freq_res=1",
                     cfg->nfilt_pos + cfg->nfilt_neg);
    if (!ispow2(cfg->nchan_subband) || cfg->nchan_subband < 2 || cfg->nchan_subband > (1u << MAX_LOGF))
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: freq_res=1 (non-convolving filterbank) needs nchan_subband=%u "
                     "to be a power of two in [2, %u]", cfg->nchan_subband, 1u << MAX_LOGF);
    if (cfg->input_nchan == 0) return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: input_nchan=0");
    const int logC = ilog2(cfg->nchan_subband);
    const int rc = fb_plain_check(ctx, logC, cfg->real_input != 0, cfg->npol, nullptr);
    if (rc != DSPSR_AMD_OK)
      return fb_fail(ctx, rc, "dspsr_amd_filterbank_create: no kernel for the %u-channel non-convolving filterbank", cfg->nchan_subband);
    dspsr_amd_filterbank* fb = new dspsr_amd_filterbank;
    fb->ctx = ctx;
    fb->cfg = *cfg;
    fb->plain_logC = logC;
    fb->N = cfg->nchan_subband;
    fb->L = cfg->real_input ? 2 * fb->N : fb->N;
    fb->nseq = cfg->real_input ? 1 : cfg->npol;
    fb->ncu = ctx->ncu;
    fb->max_parts = cfg->max_parts ? cfg->max_parts : 1;
    FbGeom& g = fb->g;
    g = FbGeom();
    g.real_input = cfg->real_input ? 1 : 0;
    g.npol = cfg->npol;
    g.C = cfg->nchan_subband;
    g.nsub = 1;
    g.nfilt_pos = 0;
    g.nkeep = 1;
    g.xstride = fb->L;
    *out = fb;
    return DSPSR_AMD_OK;
  }
  // (any odd factor up to ODD_MAX: 3, 5, 7, 9, 15 have radix kernels of their own, the others -- 11, 13, 21, 25, ... -- the
  //  run-time-radix forms k_sub_combine_any / k_time_combine<0>)
  auto radix_ok = [](uint32_t r) { return (r & 1u) && r >= 3 && r <= ODD_MAX; };
  uint32_t msub = 0;
  if (!ispow2(cfg->freq_res)) {
    msub = odd_part(cfg->freq_res);
    if (!radix_ok(msub) || cfg->nchan_subband == 0 || !(ispow2(cfg->nchan_subband) || radix_ok(odd_part(cfg->nchan_subband) * msub)) ||
        cfg->freq_res / msub < 2 || cfg->force_four_pass == 1)
      return fb_fail(ctx, DSPSR_AMD_EINVAL,
                     "dspsr_amd_filterbank_create: freq_res=%u must be 2^k >= 2, or 2^k (k >= 1) times an odd number <= 127 (times the odd "
                     "factor of nchan_subband=%u: again <= 127)",
                     cfg->freq_res, cfg->nchan_subband);
  } else if (cfg->freq_res < 2)
    return fb_fail(ctx, DSPSR_AMD_EINVAL,
                   "dsp::Filterbank::make_preparations Response.ndat = 0 (freq_res=%u)", cfg->freq_res);
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
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: nchan_subband=%u must be 2^k or 2^k times an odd number <= 127",
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
  fb->nseq = cfg->real_input ? 1 : cfg->npol;
  // tiles: every workgroup holds min(2^14, available) points = 32 per thread
  constexpr int LOG_POINTS = LOG_POINTS_DEFAULT;
  auto imin = [](int a, int b) { return a < b ? a : b; };
  const int logPol = 1;   // the inverse passes always carry (pol0, pol1) column pairs
  // three passes (freq_res and the spectrum rows each fit one workgroup tile) when possible ...
  uint64_t p1 = 0, p2 = 0, p3 = 0, p4 = 0;
  bool three_ok = g.logM <= MAX_LOGF && g.logR <= MAX_LOGF;
  if (three_ok) {
    g.logT1 = imin(g.logR, LOG_POINTS - g.logM);
    g.logT2 = imin(g.logM, LOG_POINTS - g.logR);
    int t3 = LOG_POINTS - g.logM - logPol;
    if (t3 < 0) t3 = 0;
    g.logX3 = imin(logC, t3);                    // X layout: keeps the pass-2 store runs at T2*X3 elements
    g.logT3 = g.logX3;                           // pass-3 tile = one layout block
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
    const bool blocked = logT2f < 4;
    // freq_res = Ma*Mb: the split that measured fastest (same file).  The second inverse pass likes Mb = 256 (two
    // radix-16 stages, 32 adjacent output samples per run), the first one Ma <= 2^11 (>= 4 columns per tile).
    static const signed char lma_best[27] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, -1, 10, 11, 11, 11, 12, 12, 13};
    int lma = (logMf + 1) / 2;
    if (logMf >= 14 && logMf <= 26) lma = lma_best[logMf] > 0 ? lma_best[logMf] : (blocked ? 11 : 9);
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
      g.xblock = 1u << (lb + g.logT2);           // (padding between the blocks measured irrelevant: profiles/r04_experiments.txt item 12)
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
  fb->wg3 = (!g.four_pass && 2 * fb->lds3 + 1024 <= 160 * 1024) ? 2 : 1;      // workgroups per compute unit (small tiles: two)
  fb->wg1 = (2 * fb->lds1 + 1024 <= 160 * 1024 && fb->nt1 <= 256) ? 2 : 1;
  {
    // kernels of this geometry and their dynamic-LDS limits (once; perform only launches)
    const bool full1 = g.logT1 == full_logt(g.logM), full2 = g.logT2 == full_logt(g.logR),
               full3 = !g.four_pass && g.logT3 + 1 == full_logt(g.logM);
    fb->k1_w1 = fb_pick1(g.logM, 1, full1);
    fb->k1_w4 = fb_pick1(g.logM, 4, full1);
    // two-column tiles of 2^13 rows whose A runs would be half cache lines: transformed in pairs (k_fwd_cols_dual)
    if (full1 && g.logM == 13 && g.logT1 == 1 && g.logT1 + g.logT2 < 4 && g.logR >= 2) {
      fb->k1d_w1 = fb_pick1_dual(1);
      fb->k1d_w4 = fb_pick1_dual(4);
    }
    fb->k2 = fb_pick2(g.logR, full2);
    if (g.four_pass) {
      fb->k3a = fb_pick3a(g.logMa, g.xblocked != 0, g.real_input != 0, g.logTm == 13 - g.logMa && g.logMa <= 12 && fb->nt3 == 512);
      const bool full4 = g.logTt == 13 - g.logMb && g.logMb <= 12 && fb->nt4 == 512;
      fb->k3b = fb_pick3b(g.logMb, false, full4);
      fb->k3bf = fb_pick3b(g.logMb, true, full4);
    } else {
      fb->k3 = fb_pick3(g.logM, full3);
      fb->k3f = fb_pick3f(g.logM, full3);
      fb->k3s = fb_pick3s(g.logM, full3);
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
      if (e == hipSuccess && fb->k1d_w1) e = allow_lds(fb->k1d_w1, fb->lds1);
      if (e == hipSuccess && fb->k1d_w4) e = allow_lds(fb->k1d_w4, fb->lds1);
      if (e == hipSuccess) e = allow_lds(fb->k2, fb->lds2);
      if (e == hipSuccess && fb->k3) e = allow_lds(fb->k3, fb->lds3);
      if (e == hipSuccess && fb->k3f) e = allow_lds(fb->k3f, fb->lds3f);
      if (e == hipSuccess && fb->k3s) e = allow_lds(fb->k3s, fb->lds3);
      if (e == hipSuccess && fb->k3a) e = allow_lds(fb->k3a, fb->lds3);
      if (e == hipSuccess && fb->k3b) e = allow_lds(fb->k3b, fb->lds4);
      if (e == hipSuccess && fb->k3bf) e = allow_lds(fb->k3bf, fb->lds4);
    }
    if (!have || e != hipSuccess) {
      delete fb;
      return have ? fb_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_create: hipFuncSetAttribute: %s", hipGetErrorString(e))
                  : fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_create: no kernel for this geometry");
    }
  }
  // Two-pass path: forward and inverse levels together fit two workgroup tiles (fb_two_pass.hip).  Complex dual-pol input,
  // 512 <= freq_res <= 4096, Fb = 2^13 / freq_res channels per inverse tile, Fa = L / Fb with freq_res <= Fa <= 2^14, i.e.
  // Fb <= nchan_subband <= 2^27 / freq_res^2 (the 50 MHz sub-band geometry -F 512:D -x 512 is the upper end).  Taken per call
  // when the input is the generic 8-bit block (fb_run); the three-pass kernels above serve every other input form.
  // force_four_pass == 2 switches it off (comparison runs and tests).
  {
    const int lfb = 13 - logMf, lfa = logL - lfb;
    if (nsub == 1 && !g.four_pass && !cfg->real_input && cfg->npol == 2 && cfg->force_four_pass != 2 && logMf >= 9 && logMf <= 12 &&
        lfa >= logMf && lfa <= 14 && ctx->ncu > 0) {
      hipError_t e2 = hipSuccess;
      bool have1 = false;
      if (lfa == 14) {
        fb->k1c = fb_pick_col1();
        fb->lds1c = lds_total_words_host(1u << 14, 12) * sizeof(cf);
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
      fb->k2r = fb_pick_rinv(logMf, 0);
      fb->k2rf = fb_pick_rinv(logMf, 1);
      fb->k2rs = fb_pick_rinv(logMf, 2);
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
        if (e2 == hipSuccess && fb->k2rs) e2 = allow_lds(fb->k2rs, fb->lds2r);
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
  // dsp::Convolution with a short response (n_fft <= 8192) on complex float rows: the whole transform of a (channel, part) sequence,
  // both polarisations, fits one workgroup tile -- forward transform, response, backward transform, keep window and Detection in ONE
  // pass over the rows (fb_conv1.hip) instead of four.  force_four_pass != 0 keeps the four-pass kernels (tests, comparison runs).
  if (cfg->nchan_subband == 1 && !cfg->real_input && cfg->npol == 2 && cfg->force_four_pass == 0 && !msub && nsub == 1 &&
      g.logMf >= 6 && g.logMf <= 13 && fb_conv1_check(g.logMf, nullptr) == DSPSR_AMD_OK)
    fb->conv1_logM = g.logMf;
  // The same with a response of 2^14 ... 2^21 points: three tile passes instead of four -- the forward transform's second pass and the
  // inverse transform's first one run along the same rows and are one pass (fb_conv3.hip).  Launch groups of channels x parts that fill
  // about 2 GB per scratch buffer.
  if (fb->conv1_logM < 0 && cfg->nchan_subband == 1 && !cfg->real_input && cfg->npol == 2 && cfg->force_four_pass == 0 && !msub && nsub == 1 &&
      g.logMf >= CONV3_MIN_LOGM && g.logMf <= CONV3_MAX_LOGM && fb_conv3_check(g.logMf) == DSPSR_AMD_OK) {
    const uint64_t seq_bytes = (2ull << g.logMf) * sizeof(cf);           // one (channel, part), both polarisations
    uint64_t max_seq = (2ull << 30) / seq_bytes;
    uint64_t np = fb->max_parts < max_seq ? fb->max_parts : max_seq;
    if (np < 1) np = 1;
    uint64_t ch = max_seq / np;
    if (ch < 1) ch = 1;
    if (ch > cfg->input_nchan) ch = cfg->input_nchan;
    // (the two scratch buffers are allocated by the first call that takes this path: an object fed 8-bit blocks never does)
    fb->conv3_logM = g.logMf;
    fb->conv3_ch = (uint32_t)ch;
    fb->conv3_parts = (uint32_t)np;
    fb->conv3_bytes = np * ch * seq_bytes;
  }
  // dsp::Convolution behind a filterbank (nchan_subband = 1 on many input channels, `dspsr -F N`): one launch group per input channel
  // holds parts x 2 x freq_res points -- a tile or two per compute unit and four launches per channel.  The channels of a GROUP run
  // as one launch group instead (fb_run_batched): forward passes on the group's (part, pol, channel) sequences with this object's
  // per-channel geometry, inverse passes of a `group`-channel filterbank object (natural spectrum order) on the spectra laid side
  // by side.  Complex float32 rows read in place by pass 1; other inputs keep the loop over the channels.
  if (fb->conv1_logM < 0 && fb->conv3_logM < 0 && cfg->nchan_subband == 1 && !cfg->real_input && cfg->npol == 2 && cfg->input_nchan >= 4 && cfg->force_four_pass != 2 &&
      g.four_pass && !g.xblocked && !msub && nsub == 1 && !(g.logR >= 6 && g.logT1 <= 4) && fb->k1_w4) {
    uint32_t ch = 1;
    while (ch < 64 && cfg->input_nchan % (2 * ch) == 0) ch *= 2;
    // (scratch of the group object: max_parts x 2 x ch x freq_res elements per buffer; keep it near 2 GB)
    while (ch > 2 && (uint64_t)fb->max_parts * 2 * ch * fb->N * sizeof(cf) > (2ull << 30)) ch /= 2;
    // (the group object must keep its spectrum in natural order: ch * freq_res <= 2^21, see `blocked` above)
    while (ch > 1 && ilog2(ch) + g.logMf > 21) ch /= 2;
    if (ch >= 2) {
      dspsr_amd_filterbank_config c2 = *cfg;
      c2.nchan_subband = ch;
      c2.input_nchan = cfg->input_nchan / ch;
      c2.force_four_pass = 1;
      c2.max_parts = fb->max_parts;
      dspsr_amd_filterbank* inv = nullptr;
      if (dspsr_amd_filterbank_create(ctx, &c2, &inv) == DSPSR_AMD_OK && inv) {
        if (inv->g.four_pass && !inv->g.xblocked && inv->k3a && inv->k3b && inv->g.logMf == g.logMf) fb->batch = inv;
        else dspsr_amd_filterbank_destroy(inv);
      }
      ctx->error[0] = 0;                       // (a refusal of the group object is not an error of this call)
    }
  }
  *out = fb;
  return DSPSR_AMD_OK;
}

extern "C" void dspsr_amd_filterbank_destroy(dspsr_amd_filterbank* fb)
{
  if (!fb) return;
  if (fb->batch) dspsr_amd_filterbank_destroy(fb->batch);
  (void)hipStreamSynchronize(fb->ctx->stream);
  if (fb->A) (void)hipFree(fb->A);
  if (fb->X) (void)hipFree(fb->X);
  if (fb->kernel) (void)hipFree(fb->kernel);
  if (fb->kernel_nat) (void)hipFree(fb->kernel_nat);
  if (fb->S1) (void)hipFree(fb->S1);
  if (fb->S2) (void)hipFree(fb->S2);
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
  if (fb->conv3_logM >= 0) {
    // the three-pass convolution reads the response in the order of its pass-B tiles (fb_conv3_response_order: whole lines per load)
    if (!fb->kernel_nat && hipMalloc((void**)&fb->kernel_nat, expect * sizeof(cf)) != hipSuccess)
      return fb_fail(fb->ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_set_kernel: hipMalloc failed");
    std::vector<cf> ord(expect);
    for (uint32_t c = 0; c < fb->cfg.input_nchan; c++) fb_conv3_response_order(fb->conv3_logM, src + (uint64_t)c * fb->N, ord.data() + (uint64_t)c * fb->N);
    if (hipMemcpy(fb->kernel_nat, ord.data(), expect * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess)
      return fb_fail(fb->ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_set_kernel: copy of the response failed");
  } else if (fb->conv1_logM >= 0 && fb->g.xblocked) {
    // (the one-pass convolution reads the response in natural order; the four-pass kernels of this geometry take it blocked)
    if (!fb->kernel_nat && hipMalloc((void**)&fb->kernel_nat, expect * sizeof(cf)) != hipSuccess)
      return fb_fail(fb->ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_set_kernel: hipMalloc failed");
    if (hipMemcpy(fb->kernel_nat, src, expect * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess)
      return fb_fail(fb->ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_set_kernel: copy of the response failed");
  }
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

// dsp::Convolution on the channels of a filterbank, one launch group per GROUP of channels (see dspsr_amd_filterbank_create):
//   pass 1 / pass 2   this object's kernels and per-channel geometry on the virtual sequences vp = (part * 2 + pol) * CH + c of the
//                     group (FbIn::batch): spectra X[vp][freq_res] = X[part][pol][c][m], natural order
//   inverse passes    those of the group object (CH channels per "input channel", four-pass, natural order): its X layout is exactly
//                     that, its chirp slice [CH][freq_res] the group's rows of this object's kernel, its output channels chan0 + c
// Same kernels, same arithmetic per channel as the loop over the channels.
static int fb_run_batched(dspsr_amd_filterbank* fb, const FbIn& in, const FbOut& out, uint64_t npart, uint64_t chan_stride)
{
  dspsr_amd_ctx* ctx = fb->ctx;
  dspsr_amd_filterbank* inv = fb->batch;
  const FbGeom& gf = fb->g;
  const FbGeom& gi = inv->g;
  const uint32_t CH = inv->cfg.nchan_subband, ngroup = fb->cfg.input_nchan / CH;
  const uint32_t Rr = 1u << gf.logR, M = 1u << gf.logM;
  const float* in_f32 = (const float*)in.base;
  for (uint32_t grp = 0; grp < ngroup; grp++) {
    const cf* kern = fb->kernel ? fb->kernel + (uint64_t)grp * CH * fb->N : nullptr;
    FbOut co = out;
    co.chan0 = grp * CH;
    uint32_t nb_step = 0;
    for (uint64_t part0 = 0; part0 < npart; part0 += nb_step) {
      uint32_t nb = (uint32_t)((npart - part0) < inv->max_parts ? (npart - part0) : inv->max_parts);
      {
        uint64_t per_part = (uint64_t)(Rr >> gf.logT1) * 2 * CH;
        const uint64_t i2 = (uint64_t)(M >> gf.logT2) * 2 * CH, i3a = (uint64_t)gi.C << (gi.logMb - gi.logTm),
                       i3b = (uint64_t)gi.C << (gi.logMa - gi.logTt);
        if (i2 > per_part) per_part = i2;
        if (i3a > per_part) per_part = i3a;
        if (i3b > per_part) per_part = i3b;
        while (nb > 1 && per_part * nb >= (1ull << 31)) nb /= 2;
        nb_step = nb;
      }
      const uint32_t nvp = nb * 2 * CH;
      FbIn ci = in;
      ci.base = in_f32 + (uint64_t)grp * CH * chan_stride;
      ci.batch = CH;
      ci.chan_stride_c = chan_stride / 2;
      ci.nchan = 1; ci.ichan = 0;
      const uint64_t n1 = (uint64_t)(Rr >> gf.logT1) * nvp, n2 = (uint64_t)(M >> gf.logT2) * nvp;
      hipLaunchKernelGGL(fb->k1_w4, dim3(grid_for(n1, fb->ncu * fb->wg1)), dim3(fb->nt1), fb->lds1, ctx->stream, gf, ci, inv->A, ctx->tw, part0,
                         nvp, 1u, 32u);
      hipLaunchKernelGGL(fb->k2, dim3(grid_for(n2, fb->ncu)), dim3(fb->nt2), fb->lds2, ctx->stream, gf, inv->A, inv->X, ctx->tw, nvp, 1u, 4u);
      const uint64_t n3a = ((uint64_t)gi.C << (gi.logMb - gi.logTm)) * nb, n3b = ((uint64_t)gi.C << (gi.logMa - gi.logTt)) * nb;
      hipLaunchKernelGGL(inv->k3a, dim3(grid_for(n3a, fb->ncu)), dim3(inv->nt3), inv->lds3, ctx->stream, gi, inv->X, kern, inv->A, ctx->tw, nb, 8u);
      hipLaunchKernelGGL(inv->k3b, dim3(grid_for(n3b, fb->ncu)), dim3(inv->nt4), inv->lds4, ctx->stream, gi, inv->A, co, ctx->tw, part0, nb, 8u);
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fb_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_filterbank_perform: launch failed: %s", hipGetErrorString(e));
  return DSPSR_AMD_OK;
}

static int fb_run(dspsr_amd_filterbank* fb, FbIn in, FbOut out, uint64_t npart, uint64_t in_chan_stride_bytes_or_floats)
{
  dspsr_amd_ctx* ctx = fb->ctx;
  if (!fb->kernel_set)
    return fb_fail(ctx, DSPSR_AMD_ESTATE, "dspsr_amd_filterbank_perform: set_kernel (Engine::setup) not called");
  if (npart == 0) return DSPSR_AMD_OK;
  const FbGeom& g = fb->g;
  if (fb->plain_logC >= 0) {
    if (out.kind != 0 && out.kind != 1 && out.kind != 2)
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: output kind %d has no non-convolving form", out.kind);
    const int rc = fb_plain_launch(ctx, fb->plain_logC, g.real_input != 0, (uint32_t)g.npol, fb->cfg.input_nchan, fb->kernel, in, out,
                                   in_chan_stride_bytes_or_floats, npart);
    if (rc != DSPSR_AMD_OK) return fb_fail(ctx, rc, "dspsr_amd_filterbank_perform: launch of the non-convolving filterbank failed");
    return DSPSR_AMD_OK;
  }
  if (fb->conv1_logM >= 0 && in.kind == 0 && (out.kind == 0 || out.kind == 1 || out.kind == 2) && (in_chan_stride_bytes_or_floats % 2) == 0 &&
      (in.pol_stride % 2) == 0 && ((uintptr_t)in.base % 8) == 0) {
    const cf* kern = fb->kernel ? (fb->g.xblocked ? fb->kernel_nat : fb->kernel) : nullptr;
    const int rc = fb_conv1_launch(ctx, fb->conv1_logM, (const float*)in.base, in_chan_stride_bytes_or_floats, in.pol_stride, 2 * in.part_step, kern, out,
                                   fb->cfg.input_nchan, g.nfilt_pos, g.nkeep, npart);
    if (rc != DSPSR_AMD_OK) return fb_fail(ctx, rc, "dspsr_amd_filterbank_perform: launch of the one-pass convolution failed");
    return DSPSR_AMD_OK;
  }
  if (fb->conv3_logM >= 0 && in.kind == 0 && (out.kind == 0 || out.kind == 1 || out.kind == 2) && (in_chan_stride_bytes_or_floats % 2) == 0 &&
      (in.pol_stride % 2) == 0 && ((uintptr_t)in.base % 8) == 0) {
    if (!fb->S1) {
      if (hipMalloc((void**)&fb->S1, fb->conv3_bytes) != hipSuccess || hipMalloc((void**)&fb->S2, fb->conv3_bytes) != hipSuccess) {
        if (fb->S1) (void)hipFree(fb->S1);
        fb->S1 = fb->S2 = nullptr;
        (void)hipGetLastError();
        return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform: hipMalloc of 2 x %llu bytes of scratch for the three-pass convolution failed",
                       (unsigned long long)fb->conv3_bytes);
      }
    }
    const cf* kern = fb->kernel ? fb->kernel_nat : nullptr;             // (in pass-B order, see set_kernel)
    for (uint32_t c0 = 0; c0 < fb->cfg.input_nchan; c0 += fb->conv3_ch) {
      const uint32_t nc = fb->cfg.input_nchan - c0 < fb->conv3_ch ? fb->cfg.input_nchan - c0 : fb->conv3_ch;
      FbOut co = out;
      co.chan0 = out.chan0 + c0;
      for (uint64_t part0 = 0; part0 < npart; part0 += fb->conv3_parts) {
        const uint32_t np = (uint32_t)(npart - part0 < fb->conv3_parts ? npart - part0 : fb->conv3_parts);
        const int rc = fb_conv3_launch(ctx, fb->conv3_logM, (const float*)in.base + (uint64_t)c0 * in_chan_stride_bytes_or_floats,
                                       in_chan_stride_bytes_or_floats, in.pol_stride, 2 * in.part_step, kern ? kern + (uint64_t)c0 * fb->N : nullptr, co,
                                       nc, g.nfilt_pos, g.nkeep, part0, np, fb->S1, fb->S2);
        if (rc != DSPSR_AMD_OK) return fb_fail(ctx, rc, "dspsr_amd_filterbank_perform: launch of the three-pass convolution failed");
      }
    }
    return DSPSR_AMD_OK;
  }
  if (fb->batch && in.kind == 0 && (out.kind == 0 || out.kind == 1 || out.kind == 2) && (in_chan_stride_bytes_or_floats % 2) == 0)
    return fb_run_batched(fb, in, out, npart, in_chan_stride_bytes_or_floats);
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
  bool pret = two || ((fast8 || fastc) && g.logR >= 2 && g.logT1 <= 5);   // rows of >= 128 B need no regrouping
  if (pret && in.kind == 2 && (in.part_step % 4) != 0) pret = false;
  if (pret && !fb->Rt) {
    if (hipMalloc((void**)&fb->Rt, (size_t)fb->max_parts * fb->nseq * fb->L * sizeof(uint16_t)) != hipSuccess)
      return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform: hipMalloc of the 8-bit regroup buffer failed");
  }
  // float32 rows (what Filterbank::Engine::perform is given): regrouped likewise, 8-byte elements, into the idle X scratch
  const bool pretf = in.kind == 0 && g.npol == 2 && g.logR >= 6 && g.logM >= 1 && g.logT1 >= 1 && g.logT1 <= 4 &&
                     ((uintptr_t)in.base % 16) == 0 && (in.part_step % 4) == 0 && (in.pol_stride % 4) == 0 &&
                     (in_chan_stride_bytes_or_floats % 4) == 0;
  const int raww = (pret || (fast8 && in.kind == 1)) ? 1 : 4;
  k1_t k1 = raww == 1 ? fb->k1_w1 : fb->k1_w4;
  const k1_t k1d = raww == 1 ? fb->k1d_w1 : fb->k1d_w4;
  k2_t k2 = fb->k2;
  k3_t k3 = out.kind == 3 ? fb->k3f : out.kind == 5 ? fb->k3s : fb->k3;
  k3a_t k3a = fb->k3a;
  k3b_t k3b = out.kind == 4 ? fb->k3bf : fb->k3b;
  if (!k1 || !k2 || (g.four_pass ? (!k3a || !k3b) : !k3))
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: no kernel for this geometry");
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
        // The sub-sequences of a launch share ONE de-interleaved block, so its parts must start a multiple of R samples apart.  The
        // part step is a multiple of the odd factor of nchan_subband always, of the factor of freq_res only when the kept length
        // allows it: otherwise the group runs as rp interleaved sub-groups -- parts q, q + rp, q + 2 rp, ... start rp steps apart,
        // a multiple of R -- each with its own de-interleave (round 4 fell back to ONE part per launch here: a cliff for persistent
        // kernels that amortise ramp-up and tail over 32-256 parts).
        auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; };
        const uint32_t rp = (in.part_step % R) ? R / gcd((uint32_t)(in.part_step % R), R) : 1u;
        const uint32_t nb_all = nb;
        for (uint32_t q = 0; q < rp && q < nb_all; q++) {
        const uint32_t nb = (nb_all - q + rp - 1) / rp;                                  // parts of this sub-group
        const uint64_t part0q = part0 + q;
        // (sub-groups: the parts do not follow each other -- every part is a window of its own in the de-interleaved block, L / R
        //  samples per sub-sequence, instead of one contiguous range that would cover the other sub-groups' samples as well)
        const uint64_t step = in.part_step * rp;                                         // (a multiple of R, like L)
        const uint64_t wlen = rp > 1 ? fb->L / R : ((uint64_t)(nb - 1) * step + fb->L) / R, nper = rp > 1 ? (uint64_t)nb * wlen : wlen;
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
                       fb->cfg.input_nchan, ichan, (uint32_t)g.npol, ndim, part0q * in.part_step, nper, R, sub_stride,
                       rp > 1 ? nb : 1u, wlen, step};
        fb_launch_sub_split(ctx->stream, sp, fb->dsub, fb->ncu);
        const uint64_t Ls = fb->L / R;
        for (uint32_t c = 0; c < R; c++) {
          FbIn cs = in;
          cs.kind = in.kind == 0 ? 0 : 1;                           // (the split writes the generic byte order)
          cs.base = fb->dsub + (size_t)c * sub_stride;
          cs.pol_stride = in.kind == 0 ? nper * ndim : 0;
          cs.part_step = rp > 1 ? wlen : step / R;
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
          if (!k1s) return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: no kernel for this geometry");
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
          hipLaunchKernelGGL(k2, dim3(grid_for(n2s, fb->ncu)), dim3(fb->nt2), fb->lds2, ctx->stream, g, fb->A, fb->X + c * Ls,
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
          (void)fb_launch_sub_combine(ctx->stream, g, fb->X, nb * fb->nseq, fb->ncu, nullptr, fb->Xp, fb->out_M, fb->msub);
          // Y[pseudo-channel][pol][part of the group][Mi] complex: rows (pseudo-channel, pol), parts 2*Mi floats apart
          FbOut yo = {1, (float*)fb->Y, (uint64_t)g.npol * fb->max_parts * Mi * 2, (uint64_t)fb->max_parts * Mi * 2, Mi * 2, 0, 2, 0};
          hipLaunchKernelGGL(fb->k3, dim3(grid_for(n3s, fb->ncu * fb->wg3)), dim3(fb->nt3), fb->lds3, ctx->stream, g, fb->Xp, kern, yo, ctx->tw,
                             0ull, nb, nb);
          TimeCombine tc = {fb->Y, (uint64_t)g.npol * fb->max_parts * Mi, (uint64_t)fb->max_parts * Mi, (uint32_t)g.logM, fb->out_M,
                            fb->out_nfilt_pos, fb->out_nkeep, fb->out_C, (uint32_t)g.npol, part0q, nb, rp, make_odd_tw(fb->msub)};
          FbOut cu = co;
          cu.chan0 = ichan * fb->out_C;
          if (cu.kind == 1 || cu.kind == 2) fb_launch_time_combine(ctx->stream, tc, cu, fb->msub, fb->ncu);
          continue;                                                                    // (next sub-group)
        }
        if (rp != 1) return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform: part step %llu is not a multiple of %u",
                                    (unsigned long long)in.part_step, R);            // (cannot happen: see above)
        // (factors without a radix kernel of their own combine out of place, into the A scratch -- idle behind pass 2)
        const cf* Xc = fb_launch_sub_combine(ctx->stream, g, fb->X, nb * fb->nseq, fb->ncu, fb->A);
        if (co.kind == 3) {
          const int rc = fb_launch_fused(fb, k3, Xc, kern, co, part0, nb, fused_segmented);
          if (rc != DSPSR_AMD_OK) return rc;
        } else {
          // (search mode: one workgroup per tile of channels, walking the group's parts in order)
          hipLaunchKernelGGL(k3, dim3(grid_for(co.kind == 5 ? n3s / nb : n3s, fb->ncu * fb->wg3)), dim3(fb->nt3), fb->lds3, ctx->stream, g, Xc,
                             kern, co, ctx->tw, part0, nb, nb);
        }
        }
        continue;
      }
      if (two) {
        // Two passes (fb_two_pass.hip): regroup per column, whole-column forward pass, rows + inverse pass -- the spectrum never
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
        } else if (co.kind == 5) {
          hipLaunchKernelGGL(fb->k2rs, dim3(grid_for(tiles, fb->ncu)), dim3(512), fb->lds2r, ctx->stream, g, fb->A, kern, co, ctx->tw, part0,
                             nb, nb);
        } else {
          hipLaunchKernelGGL(fb->k2r, dim3(grid_for((uint64_t)tiles * nb, fb->ncu)), dim3(512), fb->lds2r, ctx->stream, g, fb->A, kern, co,
                             ctx->tw, part0, nb, nb);
        }
        continue;
      }
      // persistent grids: one workgroup per CU (LDS-limited), a multiple of 8 so the XCD-aware item order applies
      const uint64_t n1 = (uint64_t)(Rr >> g.logT1) * fb->nseq * nb, n2 = (uint64_t)(M >> g.logT2) * fb->nseq * nb,
                     n3 = g.four_pass ? 0 : (uint64_t)(g.C >> g.logT3) * nb;
      // XCD dealing of the persistent items (wgfft.h persistent_item): runs of consecutive items per XCD
      const uint32_t run1 = 32, run2 = 4, run3 = nb;
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
      uint64_t p23auto = (512ull << 20) / (fb->part_elems * sizeof(cf));
      if (p23auto < 1) p23auto = 1;
      // (the fused kernel gains less, +1.4 % Msamples/s measured in three alternating runs, but consistently)
      // (segmented fused fold -- geometries with fewer channel tiles than compute units: every launch of the fused kernel
      //  brings a memset and a combine pass over the partial profiles, so whole launches win: 50 MHz sub-band geometry
      //  43.5k -> 46.0k Msamples/s, -F 256:D 60.3k -> 60.6-61.3k, tools/exp_p23.sh)
      const uint32_t p23sub = (g.four_pass || (co.kind == 3 && fused_segmented)) ? nb : (uint32_t)(p23auto < nb ? p23auto : nb);
      if (p23sub < nb) {
        const size_t lds3s = co.kind == 3 ? fb->lds3f : fb->lds3;
        if (co.kind == 3) co.plan_cap = fb->plan_cap;
        for (uint32_t s0 = 0; s0 < nb; s0 += p23sub) {
          const uint32_t ns = nb - s0 < p23sub ? nb - s0 : p23sub;
          const uint64_t off = (uint64_t)s0 * fb->part_elems;
          const uint64_t n2s = (uint64_t)(M >> g.logT2) * fb->nseq * ns;
          const uint64_t n3s = (co.kind == 3 || co.kind == 5) ? (uint64_t)(g.C >> g.logT3) : (uint64_t)(g.C >> g.logT3) * ns;
          hipLaunchKernelGGL(k2, dim3(grid_for(n2s, fb->ncu)), dim3(fb->nt2), fb->lds2, ctx->stream, g, fb->A + off,
                             fb->X + off, ctx->tw, ns, fb->nseq, run2);
          if (co.kind == 3) {
            const int rc = fb_launch_fused(fb, k3, fb->X + off, kern, co, part0 + s0, ns, fused_segmented);
            if (rc != DSPSR_AMD_OK) return rc;
          } else {
            hipLaunchKernelGGL(k3, dim3(grid_for(n3s, fb->ncu * fb->wg3)), dim3(fb->nt3), lds3s, ctx->stream, g, fb->X + off, kern, co,
                               ctx->tw, part0 + s0, ns, ns);
          }
        }
        continue;
      }
      hipLaunchKernelGGL(k2, dim3(grid_for(n2, fb->ncu)), dim3(fb->nt2), fb->lds2, ctx->stream, g, fb->A, fb->X,
                         ctx->tw, nb, fb->nseq, run2);
      if (!g.four_pass) {
        // fused fold: one workgroup owns a tile (T3 channels) for all parts of the launch
        const uint64_t items3 = (co.kind == 3 || co.kind == 5) ? (uint64_t)(g.C >> g.logT3) : n3;
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
        hipLaunchKernelGGL(k3a, dim3(grid_for(n3a, fb->ncu)), dim3(fb->nt3), fb->lds3, ctx->stream, g, fb->X, kern,
                           fb->A, ctx->tw, nb, 8u);
        if (out.kind == 4 && fb->plan_wait) {      // the fused second pass reads the segment plan
          const int rc = fold_plan_wait(out.fold, fb->plan_wait);
          fb->plan_wait = nullptr;
          if (rc != DSPSR_AMD_OK) return rc;
        }
        hipLaunchKernelGGL(k3b, dim3(grid_for(n3b, fb->ncu)), dim3(fb->nt4), fb->lds4, ctx->stream, g, fb->A, co,
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

// search mode inside the inverse pass: the staged tile (fb_common.h ts_stage) must fit the exchange buffer and the 32-bit index
// arithmetic of the epilogue must hold
static bool fb_search_fits(const dspsr_amd_filterbank* fb, uint32_t npo, uint32_t sf, uint32_t* G_out)
{
  const FbGeom& g = fb->g;
  if (g.four_pass || fb->msub || !fb->k3s || g.nkeep >= 65536 || sf >= 32768) return false;
  const uint64_t ngmax = ((uint64_t)g.nkeep + 2ull * sf - 2) / sf;
  const uint32_t G = (uint32_t)ngmax | 1u;
  const uint64_t floats = ((uint64_t)npo << g.logT3) * sf * G;
  if (floats > 2ull * fb->nt3 * PTS) return false;                       // (the buffer's padding is slack)
  if (((uint64_t)g.nkeep + sf) * sf >= (1ull << 32)) return false;       // t / sf by multiplication (ts_magic)
  if (G_out) *G_out = G;
  return true;
}

extern "C" int dspsr_amd_filterbank_search_is_fused(const dspsr_amd_filterbank* fb)
{
  return fb && fb_search_fits(fb, 2, 1, nullptr) ? 1 : 0;
}

extern "C" int dspsr_amd_filterbank_perform_search(dspsr_amd_filterbank* fb, const float* in_f32_dev, uint64_t in_chan_stride,
                                                   uint64_t in_pol_stride, uint64_t in_step, const int8_t* raw_dev, int raw_layout,
                                                   float scale, int out_state, uint32_t tscrunch, float* out_dev, uint64_t out_chan_stride,
                                                   uint64_t out_pol_stride, float* carry_dev, uint32_t* carry_count, uint64_t npart,
                                                   uint64_t* nout)
{
  if (!fb || (!in_f32_dev == !raw_dev) || !carry_count || !nout) return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = fb->ctx;
  if (out_state != DSPSR_AMD_INTENSITY && out_state != DSPSR_AMD_PPQQ)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: out_state=%d is neither Intensity nor PPQQ", out_state);
  if (out_state == DSPSR_AMD_PPQQ && fb->cfg.npol != 2)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: PPQQ needs two polarisations (npol=%u)", fb->cfg.npol);
  if (!tscrunch) return fb_fail(ctx, DSPSR_AMD_EINVAL, "dsp::TScrunch::get_factor scrunch factor not set");
  if (*carry_count >= tscrunch)
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: carry_count=%u must be < tscrunch=%u", *carry_count, tscrunch);
  const uint32_t npo = out_state == DSPSR_AMD_PPQQ ? 2u : 1u, nkeep = fb_out_nkeep(fb);
  const uint64_t ndat = npart * nkeep, total = (uint64_t)*carry_count + ndat;
  if (total + tscrunch >= (1ull << 32))
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: %llu parts x %u samples exceed the 32-bit sample index of a call",
                   (unsigned long long)npart, nkeep);
  *nout = total / tscrunch;
  const uint32_t rem = (uint32_t)(total % tscrunch);
  if (!npart) return DSPSR_AMD_OK;
  if ((!out_dev && *nout) || !carry_dev) return DSPSR_AMD_EINVAL;
  const uint64_t nchan_out = (uint64_t)fb->cfg.input_nchan * fb_out_C(fb);
  if (*nout && ((npo > 1 && out_pol_stride < *nout) || (nchan_out > 1 && out_chan_stride < (npo - 1) * out_pol_stride + *nout)))
    return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: output rows of %llu floats overlap (chan stride %llu, pol "
                   "stride %llu)", (unsigned long long)*nout, (unsigned long long)out_chan_stride, (unsigned long long)out_pol_stride);
  uint64_t step;
  dspsr_amd_filterbank_sizes(fb, nullptr, nullptr, &step, nullptr);
  FbIn in;
  if (in_f32_dev) {
    const uint32_t idim = fb->cfg.real_input ? 1 : 2;
    if (in_step % idim)
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: in_step=%llu not a multiple of ndim", (unsigned long long)in_step);
    in = {0, in_f32_dev, in_pol_stride, in_step / idim, fb->cfg.input_nchan, 0, 1.0f};
  } else {
    if (raw_layout == DSPSR_AMD_RAW_CASPSR && !(fb->cfg.real_input && fb->cfg.npol == 2 && fb->cfg.input_nchan == 1))
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: CASPSR layout needs real dual-pol single-channel input");
    if (raw_layout == DSPSR_AMD_RAW_UWB16 && (fb->cfg.real_input || fb->cfg.input_nchan != 1))
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: UWB 16-bit layout needs complex single-channel input");
    if (raw_layout != DSPSR_AMD_RAW_CASPSR && raw_layout != DSPSR_AMD_RAW_GENERIC && raw_layout != DSPSR_AMD_RAW_UWB16)
      return fb_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_filterbank_perform_search: unknown raw layout %d", raw_layout);
    in = {raw_kind(raw_layout), raw_dev, 0, step, fb->cfg.input_nchan, 0, scale};
  }
  uint32_t G = 0;
  if (fb_search_fits(fb, npo, tscrunch, &G)) {
    // Filterbank + Detection::square_law + TScrunch in one launch group: the detected stream never reaches HBM
    FbOut out = {};
    out.kind = 5; out.base = out_dev; out.chan_stride = out_chan_stride; out.pol_stride = out_pol_stride;
    out.state = out_state; out.ndim = 1;
    out.ts_sf = tscrunch; out.ts_magic = (uint32_t)(((1ull << 32) + tscrunch - 1) / tscrunch); out.ts_phase0 = *carry_count;
    out.ts_G = G; out.ts_carry = carry_dev;
    if (tscrunch == 1) out.ts_magic = 0xffffffffu;                       // (2^32 does not fit: t / 1 = t is the special case below)
    const int rc = fb_run(fb, in, out, npart, in_chan_stride);
    if (rc != DSPSR_AMD_OK) return rc;
    *carry_count = rem;
    return DSPSR_AMD_OK;
  }
  // Other geometries (freq_res > 8192, freq_res with an odd factor, tiles the staged samples do not fit): the three operations one
  // after the other on a block owned by the object -- complex rows [chan][pol], detected rows behind them
  const uint64_t crow = 2 * ndat, drow = ndat;
  const size_t need = (size_t)nchan_out * (fb->cfg.npol * crow + npo * drow);
  if (need > fb->det_floats) {
    (void)hipStreamSynchronize(ctx->stream);
    if (fb->det) (void)hipFree(fb->det);
    fb->det = nullptr; fb->det_floats = 0;
    if (hipMalloc((void**)&fb->det, need * sizeof(float)) != hipSuccess)
      return fb_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_filterbank_perform_search: hipMalloc of %zu bytes failed", need * sizeof(float));
    fb->det_floats = need;
  }
  float* cplx = fb->det;
  float* det = fb->det + (size_t)nchan_out * fb->cfg.npol * crow;
  FbOut cout = {1, cplx, fb->cfg.npol * crow, crow, 2ull * nkeep, 0, 2, 0};
  int rc = fb_run(fb, in, cout, npart, in_chan_stride);
  if (rc != DSPSR_AMD_OK) return rc;
  rc = dspsr_amd_detect_square_law(ctx, out_state == DSPSR_AMD_INTENSITY, cplx, fb->cfg.npol * crow, crow, det, npo * drow, drow,
                                   (uint32_t)nchan_out, fb->cfg.npol, ndat);
  if (rc != DSPSR_AMD_OK) return rc;
  return dspsr_amd_tscrunch_fpt(ctx, det, npo * drow, drow, out_dev, out_chan_stride, out_pol_stride, (uint32_t)nchan_out, npo, 1, ndat,
                                tscrunch, carry_dev, carry_count, nout);
}

extern "C" int dspsr_amd_filterbank_npass(const dspsr_amd_filterbank* fb, int raw_input)
{
  if (!fb) return 0;
  if (fb->plain_logC >= 0) return 1;
  if (fb->conv1_logM >= 0 && !raw_input) return 1;
  if (fb->conv3_logM >= 0 && !raw_input) return 3;
  if (fb->g.four_pass) return 4;
  return fb->two_pass && raw_input ? 2 : 3;
}

extern "C" int dspsr_amd_filterbank_fold_is_fused(const dspsr_amd_filterbank* fb)
{
  // (short responses of dsp::Convolution shapes: Detection inside the one-pass kernel, Fold as a launch of its own -- the segment-sum
  //  fold belongs to the four-pass kernels)
  if (fb && (fb->conv1_logM >= 0 || fb->conv3_logM >= 0)) return 0;     // (the three-pass convolution likewise: Detection in pass C, Fold behind it)
  if (!fb || fb->msub || fb->plain_logC >= 0) return 0;            // (freq_res = 3 * 2^k / 5 * 2^k: the last step is a pass of its own, k_time_combine)
  // (segment sums pay when most of the transform is kept: at -F 64:D -x 16384 only 1817 of 16384 samples are, the unfused pass
  //  writes just those, and the fused one measured 541 against 458 us per 8 parts)
  if (fb->g.four_pass)
    return fb->cfg.fused_fold != DSPSR_AMD_FUSED_NEVER && fb->k3bf && fb->g.logTt >= 3 &&
           (2ull * fb->g.nkeep >= (1ull << fb->g.logMf) || fb->cfg.fused_fold == DSPSR_AMD_FUSED_ALWAYS) ? 3 : 0;
  if (fb->g.nkeep >= 65536) return 0;
  if (fb->cfg.fused_fold == DSPSR_AMD_FUSED_ALWAYS) return 1;
  if (fb->cfg.fused_fold == DSPSR_AMD_FUSED_NEVER) return 0;
  // (a call that takes the two-pass path has the same tile count: Fb = 2^13 / freq_res channels per tile is the three-pass T3
  //  whenever nchan_subband >= Fb, which the two-pass path requires)
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
  if ((fmode != 1 && fmode != 2) || !prof_vec4 || fold_plan_max_run(fold) >= FOLD_FUSED_MAX_RUN) {
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

