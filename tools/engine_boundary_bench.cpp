// engine_boundary_bench: the hot path timed THROUGH THE REFERENCE'S PLUG-IN SURFACE, i.e. through the C++ adaptors of
// dspsr_amd/host/dspsr_amd_engines.h driven in DSPSR's own call order per block --
//   Filterbank::Engine::perform (Filterbank.C:547-553) -> Detection::Engine::polarimetry, ndim 2, in place
//   (Detection.C:325-334, LoadToFold1.C:545-546,1105-1109) -> Fold::fold: set_nbin / set_ndat / set_bins / Engine::fold
//   (Fold.C:724-741,817-829)
// -- on the headline geometry (dspsr -F 1024:D -x 4096 -D 1000, header.dada band), block resident in HBM, in four modes:
//   float_eager    : unpacked float32 rows, every Engine call launches at once (what a DSPSR build that "just installs the
//                    engines" gets)
//   float_deferred : float32 rows, adaptors sharing a HIP::Chain in deferred mode (fused launch group at Fold::Engine::fold)
//   raw_eager      : the packed 8-bit block handed over (set_raw_input), separate launches
//   raw_deferred   : packed block + deferred chain = the launch group bench.py's headline times, reached from the Engine API
// Containers are the functional miniatures of tests/host_mock (DSPSR's own need PSRCHIVE).  Prints one JSON object.
// Built by __graft_entry__.build() into dspsr_amd/host/engine_boundary_bench; run by bench.py (config.engine_boundary).
#include <chrono>
#include <math.h>
#include <stdio.h>

#include "dspsr_amd_engines.h"

static double now () { return std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count (); }

int main (int argc, char** argv)
{
  const unsigned npart = argc > 1 ? atoi (argv[1]) : 64, iters = argc > 2 ? atoi (argv[2]) : 6, max_parts = argc > 3 ? atoi (argv[3]) : 32;
  const unsigned C = 1024, M = 4096, nbin = 1024;
  dspsr_amd_ctx* ctx = 0;
  if (dspsr_amd_ctx_create (0, DSPSR_AMD_NEW_STREAM, &ctx) != DSPSR_AMD_OK) { printf ("{\"error\": \"no HIP device\"}\n"); return 77; }
  try
  {
    // Dedispersion::prepare + build (Dedispersion.C:216-556) for the headline band
    dspsr_amd_dedispersion_config dc = {1382.0, -400.0, 1000.0, 1, C, 1, -1, 0, 0, M, 0, 0};
    dspsr_amd_dedispersion_info di;
    char err[256];
    if (dspsr_amd_dedispersion_prepare (&dc, &di, err, sizeof err) != DSPSR_AMD_OK) { printf ("{\"error\": \"%s\"}\n", err); return 1; }
    dsp::Response resp;
    resp.impulse_pos = di.impulse_pos; resp.impulse_neg = di.impulse_neg; resp.nchan = C; resp.ndat = M;
    resp.kernel.resize (size_t (2) * C * M);
    if (dspsr_amd_dedispersion_build (&dc, M, &resp.kernel[0]) != DSPSR_AMD_OK) { printf ("{\"error\": \"dedispersion_build\"}\n"); return 1; }
    const unsigned nkeep = M - di.impulse_pos - di.impulse_neg;
    const uint64_t N = uint64_t (C) * M, overlap = 2ull * (di.impulse_pos + di.impulse_neg) * C, step = 2 * N - overlap;
    const uint64_t ndat_in = npart * step + overlap, ndat = uint64_t (npart) * nkeep;

    dsp::Memory* dmem = new HIP::DeviceMemory (ctx);
    dsp::TimeSeries in_d, out_d;
    in_d.set_nchan (1); in_d.set_npol (2); in_d.set_ndim (1); in_d.set_state (Signal::Nyquist); in_d.set_rate (800e6);
    in_d.set_memory (dmem); in_d.resize (ndat_in); in_d.set_input_sample (0);
    out_d.set_nchan (C); out_d.set_npol (2); out_d.set_ndim (2); out_d.set_rate (800e6 / (2 * C));
    out_d.set_memory (dmem); out_d.resize (ndat);
    dsp::BitSeries bits_d;
    bits_d.set_memory (dmem); bits_d.resize (ndat_in, 2); bits_d.set_input_sample (0);
    {  // synthetic block: seeded noise (sigma = 24 LSB as bytes, unit variance as floats), one 2^22-sample piece repeated
       // over the block -- the content does not change the work
      const uint64_t piece = 1ull << 22;
      std::vector<signed char> rb (piece * 2);
      std::vector<float> xf (piece);
      unsigned lcg = 20100413u;
      for (unsigned p = 0; p < 2; p++) {
        for (uint64_t t = 0; t < piece; t++) {
          float s = 0;
          for (int k = 0; k < 4; k++) { lcg = lcg * 1664525u + 1013904223u; s += float (lcg >> 8) * (1.0f / 16777216.0f) - 0.5f; }
          xf[t] = s * 1.7320508f;
          rb[2 * t + p] = (signed char) lrintf (fmaxf (-128.f, fminf (127.f, xf[t] * 24.0f)));
        }
        for (uint64_t t0 = 0; t0 < ndat_in; t0 += piece)
          HIP::check (ctx, dspsr_amd_copy (ctx, in_d.get_datptr (0, p) + t0, &xf[0], (ndat_in - t0 < piece ? ndat_in - t0 : piece) * sizeof (float),
                                           DSPSR_AMD_H2D), "h2d");
        HIP::check (ctx, dspsr_amd_stream_sync (ctx), "h2d");
      }
      for (uint64_t t0 = 0; t0 < ndat_in; t0 += piece)
        HIP::check (ctx, dspsr_amd_copy (ctx, bits_d.get_rawptr () + 2 * t0, &rb[0], (ndat_in - t0 < piece ? ndat_in - t0 : piece) * 2, DSPSR_AMD_H2D), "h2d");
      HIP::check (ctx, dspsr_amd_stream_sync (ctx), "h2d");
    }
    const float scale = (float) dspsr_amd_eight_bit_scale (0.02957);

    struct Mode { const char* name; bool deferred, raw; };
    const Mode modes[4] = {{"float_eager", false, false}, {"float_deferred", true, false}, {"raw_eager", false, true}, {"raw_deferred", true, true}};
    double rate[4], ms[4];
    uint64_t fused[4];
    for (int m = 0; m < 4; m++)
    {
      HIP::Chain* chain = new HIP::Chain (ctx);
      chain->set_deferred (modes[m].deferred);
      dsp::Filterbank fbk;
      fbk.nchan_subband = C; fbk.freq_res = M; fbk.input = &in_d; fbk.response = &resp;
      HIP::FilterbankEngine* fbe = new HIP::FilterbankEngine (ctx, chain);
      fbe->set_max_parts (max_parts);
      fbe->setup (&fbk);
      HIP::DetectionEngine* dete = new HIP::DetectionEngine (ctx, chain);
      dsp::Fold fold;
      HIP::FoldEngine* eng = new HIP::FoldEngine (ctx, chain);
      fold.set_input (&out_d); fold.set_engine (eng); fold.set_nbin (nbin);
      const double pfold = 0.0893;
      double t0 = 0;
      for (unsigned it = 0; it < iters + 2; it++)
      {
        if (it == 2) { HIP::check (ctx, dspsr_amd_stream_sync (ctx), "sync"); t0 = now (); }     // two untimed blocks
        if (modes[m].raw) fbe->set_raw_input (bits_d.get_rawptr (), ndat_in, 0, DSPSR_AMD_RAW_CASPSR, scale);
        out_d.set_state (Signal::Analytic);
        fbe->perform (&in_d, &out_d, npart, step, 2 * nkeep);
        dete->polarimetry (2, &out_d, &out_d);
        out_d.set_state (Signal::Coherence);
        fold.prepare_output ();
        fold.fold (0.1 + 0.37 * it, pfold, 0, ndat);
      }
      HIP::check (ctx, dspsr_amd_stream_sync (ctx), "sync");
      const double dt = (now () - t0) / iters;
      ms[m] = dt * 1e3;
      rate[m] = double (npart) * step / dt / 1e6;
      fused[m] = chain->get_fused_blocks ();
      dsp::PhaseSeries* res = fold.get_result ();
      uint64_t h = 0;
      for (unsigned b = 0; b < nbin; b++) h += res->get_hits ()[b];
      if (h != uint64_t (iters + 2) * ndat) { printf ("{\"error\": \"%s: hits %llu != %llu\"}\n", modes[m].name, (unsigned long long) h, (unsigned long long) ((iters + 2) * ndat)); return 1; }
      delete eng; delete dete; delete fbe;                         // frees this mode's scratch before the next
    }
    printf ("{\"geometry\": \"-F 1024:D -x 4096 -D 1000, %u parts per block (%u per launch group), ndim 2 in-place detection, nbin %u\", "
            "\"unit\": \"Msamples/s\"", npart, max_parts, nbin);
    for (int m = 0; m < 4; m++) printf (", \"%s\": %.1f, \"%s_ms_per_block\": %.3f, \"%s_fused_blocks\": %llu", modes[m].name, rate[m], modes[m].name, ms[m],
                                        modes[m].name, (unsigned long long) fused[m]);
    printf ("}\n");
  }
  catch (Error& error)
  {
    printf ("{\"error\": \"%s\"}\n", error.message.c_str ());
    return 1;
  }
  dspsr_amd_ctx_destroy (ctx);
  return 0;
}
