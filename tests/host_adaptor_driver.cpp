// Runs the host adaptor classes (dspsr_amd/host/dspsr_amd_engines.h) on a real device, driven in the call order of the
// reference: Filterbank::Engine::setup/perform (Filterbank.C:219-225,547-553), Detection::Engine::polarimetry in place
// (LoadToFold1.C:545-546), then Fold.C's engine sequence prepare_output -> set_nbin -> set_ndat -> set_bins -> fold ->
// get_result (synch) -> reset (zero).  Containers are the functional miniatures of tests/host_mock (the real ones need
// PSRCHIVE); every device buffer is allocated through HIP::DeviceMemory.  Built and run by tests/test_host_adaptor.py.
// Exit code 77 = no HIP device (the CPU suite only builds it).
#include <math.h>
#include <stdio.h>

#include "dspsr_amd_engines.h"

#define REQUIRE(cond, ...) do { if (!(cond)) { fprintf (stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf (stderr, __VA_ARGS__); \
                                               fprintf (stderr, "\n"); return 1; } } while (0)

static unsigned lcg = 12345u;
static float rnd () { lcg = lcg * 1664525u + 1013904223u; return (float) ((int) (lcg >> 8) % 2001 - 1000) / 1000.0f; }

static void h2d (dspsr_amd_ctx* ctx, dsp::TimeSeries& dev, const dsp::TimeSeries& host)
{
  HIP::check (ctx, dspsr_amd_copy (ctx, dev.internal_get_buffer (), host.internal_get_buffer (), host.internal_get_size (), DSPSR_AMD_H2D), "h2d");
  HIP::check (ctx, dspsr_amd_stream_sync (ctx), "h2d");
}
static void d2h (dspsr_amd_ctx* ctx, dsp::TimeSeries& host, const dsp::TimeSeries& dev)
{
  HIP::check (ctx, dspsr_amd_copy (ctx, host.internal_get_buffer (), dev.internal_get_buffer (), dev.internal_get_size (), DSPSR_AMD_D2H), "d2h");
  HIP::check (ctx, dspsr_amd_stream_sync (ctx), "d2h");
}

// ---------------------------------------------------------------------------------------------------------------------
// Deferred mode (HIP::Chain) and the raw-input side channel: the SAME reference call order -- Filterbank::Engine::perform
// (Filterbank.C:547-553), Detection::Engine::polarimetry in place (Detection.C:325-334, LoadToFold1.C:545-546), then
// Fold::fold's set_nbin / set_ndat / set_bins / Engine::fold (Fold.C:724-741,817-829) -- through eager adaptors, deferred
// adaptors on float32 rows and deferred adaptors on the packed 8-bit block must give the same PhaseSeries bit for bit.
// nfold 2: TWO dsp::Fold (different periods and phases, as dspsr folds several pulsars) read the ONE detected series
// (LoadToFold1.C:917-924,948-955,1206) -- the second profile is appended to `profile` / `hits`.  zeroed: the detected series
// carries zeroed samples and the PhaseSeries per-channel hits (Fold.C:853-866): hits then holds C*nbin counts.
struct Variant { const char* name; bool chain, deferred, raw; int fused_mode; bool pieces, finish_each; int nfold; bool zeroed; };

static int run_variant (dspsr_amd_ctx* ctx, dsp::Memory* dmem, const Variant& v, const std::vector<signed char>& raw_h, float scale,
                        unsigned C, unsigned M, unsigned pos, unsigned neg, unsigned npart, unsigned nblock, const dsp::Response& resp,
                        unsigned nbin, std::vector<float>& profile, std::vector<unsigned>& hits, double& length, uint64_t counters[3])
{
  const unsigned nkeep = M - pos - neg;
  const uint64_t N = uint64_t (C) * M, nsamp_fft = 2 * N, overlap = 2 * (pos + neg) * C, step = nsamp_fft - overlap;
  const uint64_t ndat_in = npart * step + overlap, ndat = uint64_t (npart) * nkeep;
  HIP::Chain* chain = v.chain ? new HIP::Chain (ctx) : 0;
  if (chain) chain->set_deferred (v.deferred);
  dsp::TimeSeries in_h, in_d, out_d;
  in_h.set_nchan (1); in_h.set_npol (2); in_h.set_ndim (1); in_h.set_state (Signal::Nyquist); in_h.set_rate (1e6);
  in_h.resize (ndat_in);
  in_d.set_memory (dmem); in_d.internal_match (&in_h);
  dsp::BitSeries bits_h, bits_d;
  bits_d.set_memory (dmem);
  dsp::Filterbank fbk;
  fbk.nchan_subband = C; fbk.freq_res = M; fbk.input = &in_d; fbk.response = &resp;
  HIP::FilterbankEngine fbe (ctx, chain);
  fbe.set_fused_fold (v.fused_mode);
  fbe.setup (&fbk);
  HIP::DetectionEngine dete (ctx, chain);
  out_d.set_nchan (C); out_d.set_npol (2); out_d.set_ndim (2); out_d.set_rate (1e6 / (2 * C));
  out_d.set_memory (dmem); out_d.resize (ndat);
  dsp::Fold fold, fold2;
  HIP::FoldEngine* eng = new HIP::FoldEngine (ctx, chain);
  fold.set_input (&out_d); fold.set_engine (eng); fold.set_nbin (nbin);
  if (v.zeroed) { out_d.set_zeroed_data (true); eng->get_profiles ()->set_hits_nchan (C); }
  if (v.nfold == 2) { fold2.set_input (&out_d); fold2.set_engine (new HIP::FoldEngine (ctx, chain)); fold2.set_nbin (nbin); }
  const double pfold = 37.7 / out_d.get_rate (), pfold2 = 23.3 / out_d.get_rate ();
  for (unsigned b = 0; b < nblock; b++)
  {
    // block b of the stream: packed bytes ((t*nchan + c)*npol + p)*ndim + d = 2*t + p, and their unpacked image
    const signed char* rb = &raw_h[size_t (b) * step * 2];
    for (unsigned p = 0; p < 2; p++) for (uint64_t t = 0; t < ndat_in; t++) in_h.get_datptr (0, p)[t] = (float (rb[2 * t + p]) + 0.5f) * scale;
    const int64_t input_sample = int64_t (b) * step;
    in_d.set_input_sample (input_sample);
    h2d (ctx, in_d, in_h);
    if (v.raw)
    {
      bits_h.resize (ndat_in, 2); bits_h.set_input_sample (input_sample);
      memcpy (bits_h.get_rawptr (), rb, ndat_in * 2);
      bits_d.resize (ndat_in, 2);
      HIP::transfer_bitseries (ctx, &bits_h, &bits_d, &fbe, DSPSR_AMD_RAW_GENERIC, scale);
      // the float rows must then be dead weight: poison them (a pipeline that hands over the BitSeries does not unpack)
      for (unsigned p = 0; p < 2; p++) for (uint64_t t = 0; t < ndat_in; t++) in_h.get_datptr (0, p)[t] = 1e30f;
      h2d (ctx, in_d, in_h);
    }
    out_d.set_state (Signal::Analytic); out_d.reshape (2, 2);
    fbe.perform (&in_d, &out_d, npart, step, 2 * nkeep);
    if (v.finish_each) fbe.finish ();                              // Operation::record_time: Filterbank.C:551
    dete.polarimetry (2, &out_d, &out_d);                          // in place, LoadToFold1.C:545-546
    out_d.reshape (2, 2); out_d.set_state (Signal::Coherence);     // Detection::resize_output AFTER the engine call (:136-138)
    fold.prepare_output ();
    const double phi = 0.11 + 0.37 * b;
    if (v.pieces && b == 1) {                                      // a sub-integration boundary inside block 1 (Subint.h:234-309)
      fold.fold (phi, pfold, 0, 700);
      fold.fold (phi + 0.5, pfold, 700, ndat - 700);
    } else
      fold.fold (phi, pfold, 0, ndat);
    if (v.nfold == 2) { fold2.prepare_output (); fold2.fold (0.73 - 0.21 * b, pfold2, 0, ndat); }   // fold[1] of the same block
  }
  profile.clear (); hits.clear ();
  for (int f = 0; f < v.nfold; f++)
  {
    dsp::PhaseSeries* res = (f ? fold2 : fold).get_result ();
    REQUIRE (res->get_nbin () == nbin && res->get_nchan () == C && res->get_npol () == 2 && res->get_ndim () == 2, "%s: result shape", v.name);
    const size_t at = profile.size ();
    profile.resize (at + size_t (C) * 2 * nbin * 2);
    for (unsigned c = 0; c < C; c++) for (unsigned p = 0; p < 2; p++)
      memcpy (&profile[at + (size_t (c) * 2 + p) * nbin * 2], res->get_datptr (c, p), nbin * 2 * sizeof (float));
    hits.insert (hits.end (), res->get_hits (), res->get_hits () + nbin * res->get_hits_nchan ());
    if (f == 0) length = res->integration_length;
  }
  counters[0] = chain ? chain->get_fused_blocks () : 0;
  counters[1] = chain ? chain->get_eager_blocks () : 0;
  counters[2] = chain ? chain->get_dropped_blocks () : 0;
  return 0;
}

static int deferred_section (dspsr_amd_ctx* ctx, dsp::Memory* dmem)
{
  const unsigned C = 64, M = 256, pos = 5, neg = 7, npart = 5, nblock = 3, nbin = 32;
  const uint64_t N = uint64_t (C) * M, step = 2 * N - 2 * (pos + neg) * C, ndat_in = npart * step + 2 * (pos + neg) * C;
  std::vector<signed char> raw_h ((size_t (nblock - 1) * step + ndat_in) * 2);
  for (size_t i = 0; i < raw_h.size (); i++) { lcg = lcg * 1664525u + 1013904223u; raw_h[i] = (signed char) ((int) (lcg >> 24) - 128); }
  dsp::Response resp;
  resp.impulse_pos = pos; resp.impulse_neg = neg; resp.nchan = C; resp.ndat = M;
  resp.kernel.resize (2 * N);
  for (uint64_t k = 0; k < N; k++) { const float a = 3.0f * rnd (); resp.kernel[2 * k] = cosf (a); resp.kernel[2 * k + 1] = sinf (a); }
  const float scale = 0.0123f;
  const Variant variants[] = {
    // name                                                           chain  deferred raw    fused mode               pieces finish
    {"eager adaptors, no chain",                                       false, false,   false, DSPSR_AMD_FUSED_AUTO,   false, false, 1, false},
    {"chain, not deferred",                                            true,  false,   false, DSPSR_AMD_FUSED_AUTO,   false, false, 1, false},
    {"deferred, float32 rows, fused kernel (one workgroup per tile)",  true,  true,    false, DSPSR_AMD_FUSED_ALWAYS, false, false, 1, false},
    {"deferred, packed 8-bit block, fused kernel",                     true,  true,    true,  DSPSR_AMD_FUSED_ALWAYS, false, false, 1, false},
    {"deferred, packed 8-bit block, library's choice of launches",     true,  true,    true,  DSPSR_AMD_FUSED_AUTO,   false, false, 1, false},
    {"eager, packed 8-bit block (perform_raw inside perform)",         true,  false,   true,  DSPSR_AMD_FUSED_AUTO,   false, false, 1, false},
    {"deferred + finish() after every perform (record_time)",          true,  true,    true,  DSPSR_AMD_FUSED_ALWAYS, false, true,  1, false},
  };
  std::vector<float> want, got;
  std::vector<unsigned> whits, ghits;
  double wlen = 0, glen = 0;
  uint64_t cnt[3];
  for (size_t i = 0; i < sizeof variants / sizeof variants[0]; i++)
  {
    const Variant& v = variants[i];
    if (run_variant (ctx, dmem, v, raw_h, scale, C, M, pos, neg, npart, nblock, resp, nbin, i ? got : want, i ? ghits : whits, i ? glen : wlen, cnt)) return 1;
    if (i == 0) {
      double power = 0;
      for (size_t k = 0; k < want.size (); k++) power += double (want[k]) * want[k];
      REQUIRE (power > 0, "deferred section: the eager profile is all zero");
      continue;
    }
    REQUIRE (ghits == whits && glen == wlen, "%s: hits / integration_length differ from the eager run", v.name);
    for (size_t k = 0; k < want.size (); k++)
      REQUIRE (got[k] == want[k], "%s: profile[%zu] = %.9g != %.9g (eager)", v.name, k, got[k], want[k]);
    if (v.deferred && !v.finish_each) REQUIRE (cnt[0] == nblock && cnt[1] == 0 && cnt[2] == 0, "%s: %llu fused / %llu eager / %llu dropped blocks",
                                              v.name, (unsigned long long) cnt[0], (unsigned long long) cnt[1], (unsigned long long) cnt[2]);
    if (v.chain && (!v.deferred || v.finish_each)) REQUIRE (cnt[0] == 0 && cnt[1] == nblock, "%s: %llu fused / %llu eager blocks", v.name,
                                                            (unsigned long long) cnt[0], (unsigned long long) cnt[1]);
    printf ("deferred section: %-66s == eager  (%llu fused, %llu eager)\n", v.name, (unsigned long long) cnt[0], (unsigned long long) cnt[1]);
  }
  // a sub-integration boundary inside a block: that block falls back to the separate launches, the others stay fused
  const Variant pe = {"eager, block 1 folded in two pieces", false, false, false, DSPSR_AMD_FUSED_AUTO, true, false, 1, false};
  const Variant pd = {"deferred, block 1 folded in two pieces", true, true, true, DSPSR_AMD_FUSED_ALWAYS, true, false, 1, false};
  if (run_variant (ctx, dmem, pe, raw_h, scale, C, M, pos, neg, npart, nblock, resp, nbin, want, whits, wlen, cnt)) return 1;
  if (run_variant (ctx, dmem, pd, raw_h, scale, C, M, pos, neg, npart, nblock, resp, nbin, got, ghits, glen, cnt)) return 1;
  REQUIRE (ghits == whits && glen == wlen, "pieces: hits / integration_length differ");
  for (size_t k = 0; k < want.size (); k++) REQUIRE (got[k] == want[k], "pieces: profile[%zu] = %.9g != %.9g", k, got[k], want[k]);
  REQUIRE (cnt[0] == nblock - 1 && cnt[1] == 1 && cnt[2] == 0, "pieces: %llu fused / %llu eager blocks", (unsigned long long) cnt[0], (unsigned long long) cnt[1]);
  printf ("deferred section: block with a sub-integration boundary falls back to eager, result identical\n");

  // TWO Folds on one deferred chain (dspsr folding several pulsars from one detected series, LoadToFold1.C:917-955,1206):
  // the chain must never fuse -- the second Fold reads the detected TimeSeries, which a fused block never writes
  {
    const Variant te = {"eager, two Folds of one detected series", false, false, false, DSPSR_AMD_FUSED_AUTO, false, false, 2, false};
    const Variant td = {"deferred chain, two Folds of one detected series", true, true, true, DSPSR_AMD_FUSED_ALWAYS, false, false, 2, false};
    if (run_variant (ctx, dmem, te, raw_h, scale, C, M, pos, neg, npart, nblock, resp, nbin, want, whits, wlen, cnt)) return 1;
    if (run_variant (ctx, dmem, td, raw_h, scale, C, M, pos, neg, npart, nblock, resp, nbin, got, ghits, glen, cnt)) return 1;
    REQUIRE (want.size () == 2 * size_t (C) * 2 * nbin * 2 && got.size () == want.size (), "two folds: profile sizes");
    REQUIRE (ghits == whits && glen == wlen, "two folds: hits / integration_length differ");
    double p2 = 0;
    for (size_t k = want.size () / 2; k < want.size (); k++) p2 += double (want[k]) * want[k];
    REQUIRE (p2 > 0, "two folds: the second eager profile is all zero");
    for (size_t k = 0; k < want.size (); k++) REQUIRE (got[k] == want[k], "two folds: profile[%zu] = %.9g != %.9g (fold %zu)", k, got[k], want[k], k / (want.size () / 2));
    REQUIRE (cnt[0] == 0 && cnt[1] == nblock && cnt[2] == 0, "two folds: %llu fused / %llu eager blocks (must never fuse)", (unsigned long long) cnt[0], (unsigned long long) cnt[1]);
    printf ("deferred section: two Folds on one deferred chain: never fused, both profiles == eager\n");
  }
  // zeroed (RFI-excised) input with per-channel hits in deferred mode: the eager fold counts the hits from the data
  {
    const Variant ze = {"eager, zeroed samples", false, false, false, DSPSR_AMD_FUSED_AUTO, false, false, 1, true};
    const Variant zd = {"deferred chain, zeroed samples", true, true, true, DSPSR_AMD_FUSED_ALWAYS, false, false, 1, true};
    if (run_variant (ctx, dmem, ze, raw_h, scale, C, M, pos, neg, npart, nblock, resp, nbin, want, whits, wlen, cnt)) return 1;
    if (run_variant (ctx, dmem, zd, raw_h, scale, C, M, pos, neg, npart, nblock, resp, nbin, got, ghits, glen, cnt)) return 1;
    REQUIRE (whits.size () == size_t (C) * nbin && ghits == whits, "zeroed deferred: per-channel hits differ");
    uint64_t hsum = 0;
    for (size_t k = 0; k < whits.size (); k++) hsum += whits[k];
    REQUIRE (hsum > 0, "zeroed deferred: no hits counted");
    for (size_t k = 0; k < want.size (); k++) REQUIRE (got[k] == want[k], "zeroed deferred: profile[%zu]", k);
    REQUIRE (cnt[0] == 0 && cnt[1] == nblock, "zeroed deferred: %llu fused blocks (must be eager)", (unsigned long long) cnt[0]);
    printf ("deferred section: zeroed samples on a deferred chain take the eager fold, per-channel hits == eager\n");
  }
  // what the chain must refuse: (a) a recorded block that no Fold asked for, (b) a reader of a fused block's intermediates
  {
    const uint64_t ndat = uint64_t (npart) * (M - pos - neg);
    HIP::Chain* chain = new HIP::Chain (ctx);
    Reference::To<HIP::Chain> keep = chain;
    chain->set_deferred (true);
    dsp::TimeSeries in_d, out_d, copy_d;
    in_d.set_nchan (1); in_d.set_npol (2); in_d.set_ndim (1); in_d.set_state (Signal::Nyquist); in_d.set_rate (1e6);
    in_d.set_memory (dmem); in_d.resize (ndat_in); in_d.zero ();
    dsp::Filterbank fbk;
    fbk.nchan_subband = C; fbk.freq_res = M; fbk.input = &in_d; fbk.response = &resp;
    HIP::FilterbankEngine fbe (ctx, chain);
    fbe.set_fused_fold (DSPSR_AMD_FUSED_ALWAYS);
    fbe.setup (&fbk);
    HIP::DetectionEngine dete (ctx, chain);
    out_d.set_nchan (C); out_d.set_npol (2); out_d.set_ndim (2); out_d.set_rate (1e6 / (2 * C));
    out_d.set_memory (dmem); out_d.resize (ndat);
    copy_d.set_memory (dmem); copy_d.internal_match (&out_d);
    dsp::Fold fold;
    fold.set_input (&out_d); fold.set_engine (new HIP::FoldEngine (ctx, chain)); fold.set_nbin (nbin);
    auto block = [&] (bool do_fold) {
      out_d.set_state (Signal::Analytic);
      fbe.perform (&in_d, &out_d, npart, step, 2 * (M - pos - neg));
      dete.polarimetry (2, &out_d, &out_d);
      out_d.set_state (Signal::Coherence);
      if (do_fold) { fold.prepare_output (); fold.fold (0.3, 37.7 / out_d.get_rate (), 0, ndat); }
    };
    // a block Fold skipped (Subint.h:270: `if (!divider.get_is_valid()) continue;`) is dropped and counted by default ...
    REQUIRE (chain->get_drop_unfolded (), "dropping unfolded blocks is the default");
    block (false);                                         // recorded, never folded
    block (true);
    REQUIRE (chain->get_dropped_blocks () == 1 && chain->get_fused_blocks () == 1, "default: %llu dropped / %llu fused",
             (unsigned long long) chain->get_dropped_blocks (), (unsigned long long) chain->get_fused_blocks ());
    // ... and an Error in strict mode
    chain->set_drop_unfolded (false);
    block (false);
    bool threw = false;
    try { block (true); } catch (Error& e) { threw = true; }
    REQUIRE (threw, "strict mode: an unfolded recorded block must be an Error at the next perform()");
    chain->set_drop_unfolded (true);
    block (true);
    REQUIRE (chain->get_dropped_blocks () == 1 && chain->get_fused_blocks () == 2, "after strict mode: %llu dropped / %llu fused",
             (unsigned long long) chain->get_dropped_blocks (), (unsigned long long) chain->get_fused_blocks ());
    // the last block was fused: out_d was never written; a reader that goes through an engine of the chain is refused
    HIP::TimeSeriesEngine tse (ctx, chain);
    tse.prepare (&copy_d);
    threw = false;
    try { tse.copy_data_fpt (&out_d, 0, 16); } catch (Error& e) { threw = true; }
    REQUIRE (threw, "copy_data_fpt from the never-written intermediate of a fused block must be an Error");
    threw = false;
    try { dete.polarimetry (2, &out_d, &copy_d); } catch (Error& e) { threw = true; }
    REQUIRE (threw, "a second Detection of a fused block's intermediate must be an Error");
    block (true);                                          // the next block clears the mark
    printf ("deferred section: unfolded block / reader of a fused block's intermediates are refused loudly\n");
  }
  return 0;
}

int main ()
{
  dspsr_amd_ctx* ctx = 0;
  if (dspsr_amd_ctx_create (0, DSPSR_AMD_NEW_STREAM, &ctx) != DSPSR_AMD_OK) { printf ("no HIP device\n"); return 77; }
  try
  {
    dsp::Memory* dmem = new HIP::DeviceMemory (ctx);
    REQUIRE (!dmem->on_host (), "DeviceMemory must not be host memory");

    // ---------------------------------------------------------------- Filterbank::Engine
    const unsigned C = 8, M = 64, pos = 5, neg = 7, nkeep = M - pos - neg, npart = 3;
    const uint64_t N = uint64_t (C) * M, nsamp_fft = 2 * N, overlap = 2 * (pos + neg) * C, step = nsamp_fft - overlap;
    dsp::TimeSeries in_h, in_d;
    in_h.set_nchan (1); in_h.set_npol (2); in_h.set_ndim (1); in_h.set_state (Signal::Nyquist); in_h.set_rate (1e6);
    in_h.resize (npart * step + overlap);
    in_d.set_memory (dmem); in_d.internal_match (&in_h);
    for (unsigned p = 0; p < 2; p++) for (uint64_t i = 0; i < in_h.get_ndat (); i++) in_h.get_datptr (0, p)[i] = rnd ();
    h2d (ctx, in_d, in_h);
    dsp::Response resp;
    resp.impulse_pos = pos; resp.impulse_neg = neg; resp.nchan = C; resp.ndat = M;
    resp.kernel.resize (2 * N);
    for (uint64_t k = 0; k < N; k++) { const float a = 3.0f * rnd (); resp.kernel[2 * k] = cosf (a); resp.kernel[2 * k + 1] = sinf (a); }
    dsp::Filterbank fbk;
    fbk.nchan_subband = C; fbk.freq_res = M; fbk.input = &in_d; fbk.response = &resp;
    HIP::FilterbankEngine fbe (ctx);
    fbe.setup (&fbk);
    REQUIRE (fbk.passband_cleared, "setup must null the passband (FilterbankCUDA.cu:78)");
    dsp::TimeSeries out_d, out2_d, out_h, out2_h;
    out_d.set_nchan (C); out_d.set_npol (2); out_d.set_ndim (2); out_d.set_state (Signal::Analytic); out_d.set_rate (1e6 / (2 * C));
    out_d.set_memory (dmem); out_d.resize (npart * nkeep);
    out2_d.set_memory (dmem); out2_d.internal_match (&out_d);
    out_h.internal_match (&out_d); out2_h.internal_match (&out_d);
    fbe.perform (&in_d, &out_d, npart, step, 2 * nkeep);
    fbe.finish ();
    {  // the same through the bare C-ABI
      dspsr_amd_filterbank_config cfg = {C, M, pos, neg, 1, 2, 1, 0, 0, DSPSR_AMD_FUSED_AUTO};
      dspsr_amd_filterbank* fb = 0;
      HIP::check (ctx, dspsr_amd_filterbank_create (ctx, &cfg, &fb), "create");
      HIP::check (ctx, dspsr_amd_filterbank_set_kernel (fb, &resp.kernel[0], N), "set_kernel");
      HIP::check (ctx, dspsr_amd_filterbank_perform (fb, in_d.get_datptr (0, 0), 0, in_d.get_datptr (0, 1) - in_d.get_datptr (0, 0),
               out2_d.get_datptr (0, 0), out2_d.get_datptr (1, 0) - out2_d.get_datptr (0, 0),
               out2_d.get_datptr (0, 1) - out2_d.get_datptr (0, 0), npart, step, 2 * nkeep), "perform");
      HIP::check (ctx, dspsr_amd_stream_sync (ctx), "sync");
      dspsr_amd_filterbank_destroy (fb);
    }
    d2h (ctx, out_h, out_d); d2h (ctx, out2_h, out2_d);
    double power = 0;
    for (unsigned c = 0; c < C; c++) for (unsigned p = 0; p < 2; p++) for (unsigned i = 0; i < 2 * npart * nkeep; i++) {
      REQUIRE (out_h.get_datptr (c, p)[i] == out2_h.get_datptr (c, p)[i], "FilterbankEngine::perform differs from the C-ABI at %u %u %u", c, p, i);
      power += double (out_h.get_datptr (c, p)[i]) * out_h.get_datptr (c, p)[i];
    }
    REQUIRE (power > 0, "filterbank output is all zero");

    // ---------------------------------------------------------------- Detection::Engine, in place (ndim 2, npol 2 -> 2)
    HIP::DetectionEngine dete (ctx);
    out_d.set_state (Signal::Coherence);
    dete.polarimetry (2, &out_d, &out_d);
    HIP::check (ctx, dspsr_amd_stream_sync (ctx), "sync");
    dsp::TimeSeries det_h;
    det_h.internal_match (&out_d);
    d2h (ctx, det_h, out_d);
    for (unsigned c = 0; c < C; c++) for (unsigned i = 0; i < npart * nkeep; i++) {
      const float pr = out_h.get_datptr (c, 0)[2 * i], pi = out_h.get_datptr (c, 0)[2 * i + 1];
      const float qr = out_h.get_datptr (c, 1)[2 * i], qi = out_h.get_datptr (c, 1)[2 * i + 1];
      const float want[4] = {pr * pr + pi * pi, qr * qr + qi * qi, pr * qr + pi * qi, pr * qi - pi * qr};    // cross_detect.ic:23-43
      const float got[4] = {det_h.get_datptr (c, 0)[2 * i], det_h.get_datptr (c, 0)[2 * i + 1], det_h.get_datptr (c, 1)[2 * i],
                            det_h.get_datptr (c, 1)[2 * i + 1]};
      for (int k = 0; k < 4; k++)
        REQUIRE (fabsf (got[k] - want[k]) <= 4e-7f * (fabsf (want[0]) + fabsf (want[1]) + 1e-30f), "polarimetry %u %u %d: %g vs %g", c, i, k, got[k], want[k]);
    }

    // ---------------------------------------------------------------- Fold::Engine in Fold.C's order
    const unsigned nbin = 24;
    const uint64_t ndat = npart * nkeep;                       // 156 detected samples per (chan, pol) row
    dsp::Fold fold;
    HIP::FoldEngine* eng = new HIP::FoldEngine (ctx);
    fold.set_input (&out_d);                                   // the detected device TimeSeries (npol 2, ndim 2)
    fold.set_engine (eng);
    fold.set_nbin (nbin);
    REQUIRE (fold.get_output () == eng->get_profiles (), "Fold::get_output must be the engine's PhaseSeries (Fold.C:88-94)");
    fold.prepare_output ();                                    // resizes and zeroes the DEVICE PhaseSeries through its Memory
    REQUIRE (!eng->get_profiles ()->get_memory ()->on_host (), "engine profiles must live in device memory");
    REQUIRE (eng->get_profiles ()->get_nfloat_span () > uint64_t (nbin) * 2, "the miniature pads rows: span != nbin*ndim is exercised");
    const double pfold = 37.7 / out_d.get_rate (), pps = (1.0 / out_d.get_rate ()) / pfold;
    std::vector<float> want (size_t (C) * 2 * nbin * 2, 0.0f);
    std::vector<unsigned> whits (nbin, 0);
    uint64_t wtotal = 0;
    auto cpu_fold = [&] (double phi, uint64_t i0, uint64_t n) {       // Fold.C:744-787 plan + :835-891 accumulate, time order
      for (uint64_t i = i0; i < i0 + n; i++) {
        phi -= floor (phi);
        const unsigned ibin = unsigned (phi * double (nbin));
        phi += pps;
        whits[ibin]++;
        for (unsigned c = 0; c < C; c++) for (unsigned p = 0; p < 2; p++) for (unsigned d = 0; d < 2; d++)
          want[((size_t (c) * 2 + p) * nbin + ibin) * 2 + d] += det_h.get_datptr (c, p)[2 * i + d];
      }
      wtotal += n;
    };
    fold.fold (0.37, pfold, 3, 70);  cpu_fold (0.37, 3, 70);
    fold.fold (0.81, pfold, 73, ndat - 73);  cpu_fold (0.81, 73, ndat - 73);
    auto check_result = [&] (const char* what) -> int {
      dsp::PhaseSeries* res = fold.get_result ();             // engine->synch (output): internal_match + copy_configuration + buffer
      REQUIRE (res != eng->get_profiles () && res->get_memory ()->on_host (), "%s: the result is the host PhaseSeries", what);
      REQUIRE (res->get_nbin () == nbin && res->get_nchan () == C && res->get_npol () == 2 && res->get_ndim () == 2, "%s: result shape", what);
      REQUIRE (res->ndat_total == wtotal, "%s: ndat_total %llu != %llu", what, (unsigned long long) res->ndat_total, (unsigned long long) wtotal);
      REQUIRE (fabs (res->integration_length - double (wtotal) / out_d.get_rate ()) <= 1e-12, "%s: integration_length", what);
      for (unsigned b = 0; b < nbin; b++) REQUIRE (res->get_hits ()[b] == whits[b], "%s: hits[%u] = %u != %u", what, b, res->get_hits ()[b], whits[b]);
      for (unsigned c = 0; c < C; c++) for (unsigned p = 0; p < 2; p++) for (unsigned k = 0; k < nbin * 2; k++)
        REQUIRE (res->get_datptr (c, p)[k] == want[(size_t (c) * 2 + p) * nbin * 2 + k], "%s: profile[%u][%u][%u] = %g != %g", what, c, p, k,
                 res->get_datptr (c, p)[k], want[(size_t (c) * 2 + p) * nbin * 2 + k]);
      return 0;
    };
    if (check_result ("two folds")) return 1;
    if (check_result ("synch again (idempotent)")) return 1;
    fold.fold (0.05, pfold, 0, 40);  cpu_fold (0.05, 0, 40);       // more data after a synch: the device profile moves on
    if (check_result ("fold after synch")) return 1;
    fold.reset ();                                                  // Engine::zero -> PhaseSeries::zero on the device, and the host copy
    want.assign (want.size (), 0.0f); whits.assign (nbin, 0); wtotal = 0;
    fold.fold (0.5, pfold, 10, 100);  cpu_fold (0.5, 10, 100);
    if (check_result ("after reset")) return 1;

    // ---------------------------------------------------------------- zeroed samples: per-channel hits counted on the device
    {
      // RFI-excised input (TimeSeries::get_zeroed_data, hits_nchan == nchan): Fold.C:853-866 counts, per channel, the samples
      // of polarisation 0 whose first float is not zero; CUDA twin fold1bin*hits (FoldCUDA.cu:415-576)
      dsp::TimeSeries z_h, z_d;
      z_h.internal_match (&out_d); z_d.set_memory (dmem); z_d.internal_match (&out_d);
      d2h (ctx, z_h, out_d);
      for (unsigned c = 0; c < C; c++) for (uint64_t i = c; i < ndat; i += 3 + c)          // another pattern in every channel
        for (unsigned p = 0; p < 2; p++) z_h.get_datptr (c, p)[2 * i] = z_h.get_datptr (c, p)[2 * i + 1] = 0.0f;
      h2d (ctx, z_d, z_h);
      z_d.set_zeroed_data (true);
      dsp::Fold zfold;
      HIP::FoldEngine* zeng = new HIP::FoldEngine (ctx);
      zeng->get_profiles ()->set_hits_nchan (C);
      zfold.set_input (&z_d); zfold.set_engine (zeng); zfold.set_nbin (nbin);
      zfold.prepare_output ();
      zfold.fold (0.37, pfold, 3, 70);
      zfold.fold (0.81, pfold, 73, ndat - 73);
      dsp::PhaseSeries* zres = zfold.get_result ();
      REQUIRE (zres->get_hits_nchan () == C, "zeroed samples: result hits_nchan");
      std::vector<unsigned> zh (size_t (C) * nbin, 0);
      std::vector<float> zp (size_t (C) * 2 * nbin * 2, 0.0f);
      auto zcpu = [&] (double phi, uint64_t i0, uint64_t n) {
        for (uint64_t i = i0; i < i0 + n; i++) {
          phi -= floor (phi);
          const unsigned ibin = unsigned (phi * double (nbin));
          phi += pps;
          for (unsigned c = 0; c < C; c++) {
            if (z_h.get_datptr (c, 0)[2 * i] != 0) zh[size_t (c) * nbin + ibin]++;
            for (unsigned p = 0; p < 2; p++) for (unsigned d = 0; d < 2; d++)
              zp[((size_t (c) * 2 + p) * nbin + ibin) * 2 + d] += z_h.get_datptr (c, p)[2 * i + d];
          }
        }
      };
      zcpu (0.37, 3, 70); zcpu (0.81, 73, ndat - 73);
      for (unsigned c = 0; c < C; c++) for (unsigned b = 0; b < nbin; b++)
        REQUIRE (zres->get_hits (c)[b] == zh[size_t (c) * nbin + b], "zeroed samples: hits[%u][%u] = %u != %u", c, b, zres->get_hits (c)[b], zh[size_t (c) * nbin + b]);
      for (unsigned c = 0; c < C; c++) for (unsigned p = 0; p < 2; p++) for (unsigned k = 0; k < nbin * 2; k++)
        REQUIRE (zres->get_datptr (c, p)[k] == zp[(size_t (c) * 2 + p) * nbin * 2 + k], "zeroed samples: profile[%u][%u][%u]", c, p, k);
      printf ("zeroed samples: per-channel hits and sums == CPU loop\n");
    }

    // ---------------------------------------------------------------- deferred mode + raw side channel
    if (deferred_section (ctx, dmem)) return 1;

    // ---------------------------------------------------------------- TimeSeries::Engine::copy_data_fpt
    HIP::TimeSeriesEngine tse (ctx);
    out2_d.zero ();
    tse.prepare (&out2_d);
    tse.copy_data_fpt (&out_d, 5, 20);
    d2h (ctx, out2_h, out2_d);
    for (unsigned c = 0; c < C; c++) for (unsigned p = 0; p < 2; p++) for (unsigned k = 0; k < 40; k++)
      REQUIRE (out2_h.get_datptr (c, p)[k] == det_h.get_datptr (c, p)[10 + k], "copy_data_fpt %u %u %u", c, p, k);

    // ---------------------------------------------------------------- TScrunch::Engine / FScrunch::Engine (digifil, LoadToFITS.C:435)
    {
      const unsigned nch = 12, npl = 2, nd = 1003, sf = 7, fsf = 3;
      for (unsigned ndim = 1; ndim <= 2; ndim++)
      {
        dsp::TimeSeries x_h, x_d, t_h, t_d, f_h, f_d;
        x_h.set_nchan (nch); x_h.set_npol (npl); x_h.set_ndim (ndim); x_h.resize (nd);
        for (unsigned c = 0; c < nch; c++) for (unsigned p = 0; p < npl; p++) for (unsigned k = 0; k < nd * ndim; k++)
          x_h.get_datptr (c, p)[k] = rnd () * rnd ();
        x_d.set_memory (dmem); x_d.internal_match (&x_h); h2d (ctx, x_d, x_h);
        t_h.set_nchan (nch); t_h.set_npol (npl); t_h.set_ndim (ndim); t_h.resize (nd / sf);
        t_d.set_memory (dmem); t_d.internal_match (&t_h);
        HIP::TScrunchEngine tsc (ctx);
        tsc.fpt_tscrunch (&x_d, &t_d, sf);
        d2h (ctx, t_h, t_d);
        for (unsigned c = 0; c < nch; c++) for (unsigned p = 0; p < npl; p++) for (unsigned o = 0; o < nd / sf; o++) for (unsigned d = 0; d < ndim; d++)
        {
          float acc = x_h.get_datptr (c, p)[(o * sf) * ndim + d];                 // TScrunch.C:165-172
          for (unsigned j = 1; j < sf; j++) acc += x_h.get_datptr (c, p)[(o * sf + j) * ndim + d];
          REQUIRE (t_h.get_datptr (c, p)[o * ndim + d] == acc, "fpt_tscrunch ndim %u [%u][%u][%u]", ndim, c, p, o);
        }
        f_h.set_nchan (nch / fsf); f_h.set_npol (npl); f_h.set_ndim (ndim); f_h.resize (nd);
        f_d.set_memory (dmem); f_d.internal_match (&f_h);
        HIP::FScrunchEngine fsc (ctx);
        fsc.fpt_fscrunch (&x_d, &f_d, fsf);
        d2h (ctx, f_h, f_d);
        for (unsigned c = 0; c < nch / fsf; c++) for (unsigned p = 0; p < npl; p++) for (unsigned k = 0; k < nd * ndim; k++)
        {
          float acc = x_h.get_datptr (c * fsf, p)[k];                              // FScrunch.C:128-141
          for (unsigned j = 1; j < fsf; j++) acc += x_h.get_datptr (c * fsf + j, p)[k];
          REQUIRE (f_h.get_datptr (c, p)[k] == acc, "fpt_fscrunch ndim %u [%u][%u][%u]", ndim, c, p, k);
        }
        bool threw = false;
        try { tsc.fpt_tscrunch (&x_d, &x_d, sf); } catch (Error& e) { threw = true; }
        REQUIRE (threw, "fpt_tscrunch in place must be refused (as CUDA::TScrunchEngine does)");
      }
      printf ("TScrunch / FScrunch engines == CPU loops (ndim 1, 2)\n");
    }

    // ---------------------------------------------------------------- `dspsr -F N` (Filterbank::Config::After, LoadToFold1.C:296-380):
    // the NON-CONVOLVING filterbank (freq_res = 1, no response: Filterbank.C:614-623) through HIP::FilterbankEngine, then
    // dsp::Convolution on its channels through HIP::ConvolutionEngine, in the reference's call order -- against direct DFTs in double
    {
      const unsigned Cp = 8, nfb = 330;                                   // 8 channels: 16 real samples per output sample
      const unsigned Mc = 64, cpos = 5, cneg = 7, cstep = Mc - cpos - cneg, ncv = (nfb - (cpos + cneg)) / cstep;
      dsp::TimeSeries x_h, x_d;
      x_h.set_nchan (1); x_h.set_npol (2); x_h.set_ndim (1); x_h.set_state (Signal::Nyquist); x_h.set_rate (16e6);
      x_h.resize (uint64_t (nfb) * 2 * Cp);
      x_d.set_memory (dmem); x_d.internal_match (&x_h);
      for (unsigned p = 0; p < 2; p++) for (uint64_t i = 0; i < x_h.get_ndat (); i++) x_h.get_datptr (0, p)[i] = rnd ();
      h2d (ctx, x_d, x_h);
      dsp::Filterbank plain;
      plain.nchan_subband = Cp; plain.freq_res = 1; plain.input = &x_d; plain.response = 0;
      HIP::FilterbankEngine pfe (ctx);
      pfe.setup (&plain);
      uint64_t pfft = 0, povl = 0, pstep = 0; unsigned pkeep = 0;
      dsp::TimeSeries c_d, c_h, y_d, y_h;
      c_d.set_nchan (Cp); c_d.set_npol (2); c_d.set_ndim (2); c_d.set_state (Signal::Analytic); c_d.set_rate (1e6);
      c_d.set_memory (dmem); c_d.resize (nfb);
      c_h.internal_match (&c_d);
      pfe.perform (&x_d, &c_d, nfb, 2 * Cp, 2);                           // in_step = nsamp_step, out_step = 2 * nkeep = 2 (Filterbank.C:517-519)
      pfe.finish ();
      (void) pfft; (void) povl; (void) pstep; (void) pkeep;
      d2h (ctx, c_h, c_d);
      double worst = 0, scale_fb = 0;
      for (unsigned p = 0; p < 2; p++) for (unsigned t = 0; t < nfb; t += 37) for (unsigned k = 0; k < Cp; k++) {
        double re = 0, im = 0;                                            // frc1d: X[k] = sum_n x[n] exp(-2 pi i n k / 2C), k < C
        for (unsigned n = 0; n < 2 * Cp; n++) {
          const double a = -2.0 * M_PI * double (n) * k / double (2 * Cp), v = x_h.get_datptr (0, p)[uint64_t (t) * 2 * Cp + n];
          re += v * cos (a); im += v * sin (a);
        }
        const double dr = c_h.get_datptr (k, p)[2 * t] - re, di = c_h.get_datptr (k, p)[2 * t + 1] - im;
        if (fabs (dr) > worst) worst = fabs (dr);
        if (fabs (di) > worst) worst = fabs (di);
        if (fabs (re) > scale_fb) scale_fb = fabs (re);
      }
      REQUIRE (worst <= 2e-6 * scale_fb && scale_fb > 0, "non-convolving filterbank differs from the direct transform by %g (scale %g)", worst, scale_fb);
      // Convolution on the filterbank's channels: nchan responses of Mc bins each (Convolution.C:338-461)
      dsp::Response cresp;
      cresp.impulse_pos = cpos; cresp.impulse_neg = cneg; cresp.nchan = Cp; cresp.ndat = Mc;
      cresp.kernel.resize (2 * size_t (Cp) * Mc);
      for (size_t k = 0; k < size_t (Cp) * Mc; k++) { const float a = 3.0f * rnd (); cresp.kernel[2 * k] = cosf (a); cresp.kernel[2 * k + 1] = sinf (a); }
      dsp::Convolution conv;
      conv.response = &cresp; conv.input = &c_d; conv.nsamp_fft = Mc; conv.nsamp_overlap = cpos + cneg;
      HIP::ConvolutionEngine cve (ctx);
      cve.prepare (&conv);
      y_d.set_nchan (Cp); y_d.set_npol (2); y_d.set_ndim (2); y_d.set_state (Signal::Analytic); y_d.set_rate (1e6);
      y_d.set_memory (dmem); y_d.resize (uint64_t (ncv) * cstep);
      y_h.internal_match (&y_d);
      cve.perform (&c_d, &y_d, ncv);
      HIP::check (ctx, dspsr_amd_stream_sync (ctx), "sync");
      d2h (ctx, y_h, y_d);
      double cworst = 0, cscale = 0;
      std::vector<double> sr (Mc), si (Mc);
      for (unsigned c = 0; c < Cp; c += 3) for (unsigned p = 0; p < 2; p++) for (unsigned part = 0; part < ncv; part += 2) {
        const float* xin = c_h.get_datptr (c, p) + 2 * size_t (part) * cstep;
        for (unsigned k = 0; k < Mc; k++) {                                // fcc1d, times the channel's response
          double re = 0, im = 0;
          for (unsigned n = 0; n < Mc; n++) {
            const double a = -2.0 * M_PI * double (n) * k / double (Mc);
            re += xin[2 * n] * cos (a) - xin[2 * n + 1] * sin (a);
            im += xin[2 * n] * sin (a) + xin[2 * n + 1] * cos (a);
          }
          const double kr = cresp.kernel[2 * (size_t (c) * Mc + k)], ki = cresp.kernel[2 * (size_t (c) * Mc + k) + 1];
          sr[k] = re * kr - im * ki; si[k] = re * ki + im * kr;
        }
        for (unsigned t = 0; t < cstep; t += 5) {                           // bcc1d (unnormalised), samples [nfilt_pos, nfilt_pos + step)
          double re = 0, im = 0;
          for (unsigned k = 0; k < Mc; k++) {
            const double a = 2.0 * M_PI * double (k) * (cpos + t) / double (Mc);
            re += sr[k] * cos (a) - si[k] * sin (a);
            im += sr[k] * sin (a) + si[k] * cos (a);
          }
          const float* got = y_h.get_datptr (c, p) + 2 * (size_t (part) * cstep + t);
          if (fabs (got[0] - re) > cworst) cworst = fabs (got[0] - re);
          if (fabs (got[1] - im) > cworst) cworst = fabs (got[1] - im);
          if (fabs (re) > cscale) cscale = fabs (re);
        }
      }
      REQUIRE (cworst <= 4e-6 * cscale && cscale > 0, "Filterbank + Convolution differs from the direct transforms by %g (scale %g)", cworst, cscale);
      printf ("dspsr -F N: non-convolving FilterbankEngine (freq_res 1) + ConvolutionEngine == direct transforms (%.1e, %.1e of the scale)\n",
              worst / scale_fb, cworst / cscale);
    }
    // ---------------------------------------------------------------- the same ConvolutionEngine with a response of 16384 points: the
    // library's three-pass form (csrc/fb_conv3.hip).  Impulses in, so that every output sample is a short sum over the inverse
    // transform h[m] = sum_k H[k] exp(+2 pi i k m / M) of the channel's response, evaluated directly in double
    {
      const unsigned Cq = 3, Mq = 16384, qpos = 900, qneg = 1100, qstep = Mq - qpos - qneg, nq = 3;
      const uint64_t nin = uint64_t (nq) * qstep + qpos + qneg;
      dsp::TimeSeries c_h, c_d, y_h, y_d;
      c_d.set_nchan (Cq); c_d.set_npol (2); c_d.set_ndim (2); c_d.set_state (Signal::Analytic); c_d.set_rate (1e6);
      c_d.set_memory (dmem); c_d.resize (nin);
      c_h.internal_match (&c_d);
      for (unsigned c = 0; c < Cq; c++) for (unsigned p = 0; p < 2; p++) for (uint64_t i = 0; i < 2 * nin; i++) c_h.get_datptr (c, p)[i] = 0.f;
      struct Imp { uint64_t n; float re, im; };
      std::vector<Imp> imps;                                             // (the same impulses in every channel and polarisation, other amplitudes)
      for (unsigned i = 0; i < 7; i++) { Imp m; m.n = (uint64_t (i) * 5231 + 77) % nin; m.re = 1.f + 0.25f * i; m.im = -0.5f + 0.125f * i; imps.push_back (m); }
      for (unsigned c = 0; c < Cq; c++) for (unsigned p = 0; p < 2; p++) for (size_t i = 0; i < imps.size (); i++) {
        c_h.get_datptr (c, p)[2 * imps[i].n] = imps[i].re * float (1 + c) * (p ? -1.f : 1.f);
        c_h.get_datptr (c, p)[2 * imps[i].n + 1] = imps[i].im * float (1 + c);
      }
      h2d (ctx, c_d, c_h);
      dsp::Response qresp;
      qresp.impulse_pos = qpos; qresp.impulse_neg = qneg; qresp.nchan = Cq; qresp.ndat = Mq;
      qresp.kernel.resize (2 * size_t (Cq) * Mq);
      for (size_t k = 0; k < size_t (Cq) * Mq; k++) { const float a = 3.0f * rnd (); qresp.kernel[2 * k] = cosf (a); qresp.kernel[2 * k + 1] = sinf (a); }
      dsp::Convolution conv;
      conv.response = &qresp; conv.input = &c_d; conv.nsamp_fft = Mq; conv.nsamp_overlap = qpos + qneg;
      HIP::ConvolutionEngine cve (ctx);
      cve.prepare (&conv);
      y_d.set_nchan (Cq); y_d.set_npol (2); y_d.set_ndim (2); y_d.set_state (Signal::Analytic); y_d.set_rate (1e6);
      y_d.set_memory (dmem); y_d.resize (uint64_t (nq) * qstep);
      y_h.internal_match (&y_d);
      cve.perform (&c_d, &y_d, nq);
      HIP::check (ctx, dspsr_amd_stream_sync (ctx), "sync");
      d2h (ctx, y_h, y_d);
      double qworst = 0, qscale = 0;
      for (unsigned c = 0; c < Cq; c++) for (unsigned p = 0; p < 2; p++) for (unsigned part = 0; part < nq; part++)
        for (unsigned t = (c + p + part) % 11; t < qstep; t += 997) {
          // y[t] of this part = sum over the impulses inside its transform window of a_i h[(t + qpos - n_i) mod M]
          double re = 0, im = 0;
          const uint64_t w0 = uint64_t (part) * qstep;
          for (size_t i = 0; i < imps.size (); i++) {
            if (imps[i].n < w0 || imps[i].n >= w0 + Mq) continue;
            const uint64_t m = (uint64_t (t) + qpos + Mq - (imps[i].n - w0)) % Mq;
            double hr = 0, hi = 0;
            for (unsigned k = 0; k < Mq; k++) {
              const double a = 2.0 * M_PI * double ((uint64_t (k) * m) % Mq) / double (Mq);
              const double kr = qresp.kernel[2 * (size_t (c) * Mq + k)], ki = qresp.kernel[2 * (size_t (c) * Mq + k) + 1];
              hr += kr * cos (a) - ki * sin (a); hi += kr * sin (a) + ki * cos (a);
            }
            const double ar = imps[i].re * double (1 + c) * (p ? -1.0 : 1.0), ai = imps[i].im * double (1 + c);
            re += ar * hr - ai * hi; im += ar * hi + ai * hr;
          }
          const float* got = y_h.get_datptr (c, p) + 2 * (size_t (part) * qstep + t);
          if (fabs (got[0] - re) > qworst) qworst = fabs (got[0] - re);
          if (fabs (got[1] - im) > qworst) qworst = fabs (got[1] - im);
          if (fabs (re) > qscale) qscale = fabs (re);
        }
      REQUIRE (qworst <= 4e-6 * qscale && qscale > 0, "ConvolutionEngine with a 16384-point response differs from the direct sums by %g (scale %g)", qworst, qscale);
      printf ("dspsr -F N: ConvolutionEngine, response of 16384 points (three tile passes) == direct sums (%.1e of the scale)\n", qworst / qscale);
    }
  }
  catch (Error& error)
  {
    fprintf (stderr, "Error: %s\n", error.message.c_str ());
    return 1;
  }
  dspsr_amd_ctx_destroy (ctx);
  printf ("host adaptor driver ok\n");
  return 0;
}
