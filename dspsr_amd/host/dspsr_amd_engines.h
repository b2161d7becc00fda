// Host adaptor classes: bind the C-ABI of include/dspsr_amd.h to DSPSR's Engine plug-in interfaces.
// A DSPSR maintainer compiles this header inside the dspsr tree (it only needs the dsp headers named
// below and -ldspsr_amd) and installs the engines where the CUDA ones are installed today
// (INTEGRATION.md).  Each method mirrors the CUDA twin it replaces:
//   HIP::DeviceMemory      <- CUDA::DeviceMemory      Kernel/Classes/MemoryCUDA.C:47-106
//   HIP::TimeSeriesEngine  <- CUDA::TimeSeriesEngine  Kernel/Classes/TimeSeriesCUDA.cu:31-200
//   HIP::FilterbankEngine  <- CUDA::FilterbankEngine  Signal/General/FilterbankCUDA.cu:73-304
//   HIP::ConvolutionEngine <- CUDA::ConvolutionEngine Signal/General/ConvolutionCUDA.cu:202-800
//   HIP::DetectionEngine   <- CUDA::DetectionEngine   Signal/General/DetectionCUDA.cu:127-322
//   HIP::FoldEngine        <- CUDA::FoldEngine        Signal/Pulsar/FoldCUDA.cu:64-697
// Errors: every non-zero C-ABI status is rethrown as the reference's `Error` with the library's message.
#ifndef DSPSR_AMD_ENGINES_H
#define DSPSR_AMD_ENGINES_H

#include "dsp/Memory.h"
#include "dsp/FilterbankEngine.h"   // dsp::Filterbank, dsp::Filterbank::Engine, dsp::Response, dsp::TimeSeries
#include "dsp/Convolution.h"        // dsp::Convolution, dsp::Convolution::Engine
#include "dsp/Detection.h"          // dsp::Detection::Engine
#include "dsp/Fold.h"               // dsp::Fold::Engine, dsp::PhaseSeries
#include "Error.h"

#include "dspsr_amd.h"

namespace HIP
{
  inline void check (dspsr_amd_ctx* ctx, int status, const char* method)
  {
    if (status != DSPSR_AMD_OK)
      throw Error (status == DSPSR_AMD_EINVAL ? InvalidParam : InvalidState, method,
                   dspsr_amd_last_error (ctx));
  }

  //! dsp::Memory on the MI355X: allocation bound to one context/stream (SingleThread.C:237-244)
  class DeviceMemory : public dsp::Memory
  {
  public:
    DeviceMemory (dspsr_amd_ctx* _ctx) : ctx (_ctx) { }
    void* do_allocate (size_t nbytes)
    { void* p = 0; check (ctx, dspsr_amd_malloc (ctx, nbytes, &p), "HIP::DeviceMemory::do_allocate"); return p; }
    void do_free (void* ptr) { check (ctx, dspsr_amd_free (ctx, ptr), "HIP::DeviceMemory::do_free"); }
    void do_zero (void* ptr, size_t nbytes)
    { check (ctx, dspsr_amd_zero (ctx, ptr, nbytes), "HIP::DeviceMemory::do_zero"); }
    void do_copy (void* to, const void* from, size_t nbytes)
    { check (ctx, dspsr_amd_copy (ctx, to, from, nbytes, DSPSR_AMD_D2D), "HIP::DeviceMemory::do_copy"); }
    bool on_host () const { return false; }
    dspsr_amd_ctx* get_context () const { return ctx; }
  protected:
    dspsr_amd_ctx* ctx;
  };

  //! dsp::TimeSeries::Engine (TimeSeries.h:211-223): device-side row copies for InputBuffering / prepend
  class TimeSeriesEngine : public dsp::TimeSeries::Engine
  {
  public:
    TimeSeriesEngine (dspsr_amd_ctx* _ctx) : ctx (_ctx), to (0) { }
    void prepare (dsp::TimeSeries* parent) { to = parent; }
    void prepare_buffer (unsigned) { }           // no staging buffer: rows are copied directly

    //! TimeSeriesCUDA.cu:75-200 (same-device case); copies ndat samples of every (chan, pol) row of
    //! `from`, starting at idat_start, to the start of the rows of the parent
    void copy_data_fpt (const dsp::TimeSeries* from, uint64_t idat_start = 0, uint64_t ndat = 0)
    {
      const unsigned nchan = to->get_nchan (), npol = to->get_npol (), ndim = to->get_ndim ();
      float* obase = to->get_datptr (0, 0);
      const float* ibase = from->get_datptr (0, 0);
      check (ctx, dspsr_amd_copy_fpt (ctx, obase, nchan > 1 ? to->get_datptr (1, 0) - obase : 0,
               npol > 1 ? to->get_datptr (0, 1) - obase : 0, ibase + idat_start * ndim,
               nchan > 1 ? from->get_datptr (1, 0) - ibase : 0, npol > 1 ? from->get_datptr (0, 1) - ibase : 0,
               nchan, npol, ndat * ndim), "HIP::TimeSeriesEngine::copy_data_fpt");
    }
  protected:
    dspsr_amd_ctx* ctx;
    dsp::TimeSeries* to;
  };

  //! dsp::Filterbank::Engine (FilterbankEngine.h:15-44)
  class FilterbankEngine : public dsp::Filterbank::Engine
  {
  public:
    FilterbankEngine (dspsr_amd_ctx* _ctx) : ctx (_ctx), fb (0) { }
    ~FilterbankEngine () { dspsr_amd_filterbank_destroy (fb); }

    //! reads exactly what CUDA::FilterbankEngine::setup reads (FilterbankCUDA.cu:73-168)
    void setup (dsp::Filterbank* filterbank)
    {
      filterbank->set_passband (NULL);          // the engine does not maintain the passband
      dspsr_amd_filterbank_config cfg;
      cfg.nchan_subband = filterbank->get_nchan_subband ();
      cfg.freq_res = filterbank->get_freq_res ();
      cfg.input_nchan = filterbank->get_input()->get_nchan ();
      cfg.npol = filterbank->get_input()->get_npol ();
      cfg.real_input = filterbank->get_input()->get_state () == Signal::Nyquist;
      cfg.nfilt_pos = cfg.nfilt_neg = 0;
      cfg.max_parts = 0;
      cfg.force_four_pass = 0;
      cfg.fused_fold = DSPSR_AMD_FUSED_AUTO;
      const float* kernel = 0;
      uint64_t ncomplex = 0;
      if (filterbank->has_response ())
      {
        const dsp::Response* response = filterbank->get_response ();
        cfg.nfilt_pos = response->get_impulse_pos ();
        cfg.nfilt_neg = response->get_impulse_neg ();
        kernel = response->get_datptr (0, 0);   // host-built, already swapped (Response.C:132-181)
        ncomplex = uint64_t (response->get_nchan ()) * response->get_ndat ();
      }
      dspsr_amd_filterbank_destroy (fb); fb = 0;
      check (ctx, dspsr_amd_filterbank_create (ctx, &cfg, &fb), "HIP::FilterbankEngine::setup");
      check (ctx, dspsr_amd_filterbank_set_kernel (fb, kernel, ncomplex), "HIP::FilterbankEngine::setup");
    }

    void set_scratch (float* _scratch) { scratch = _scratch; }   // unused: the library owns its scratch

    //! FilterbankCUDA.cu:181-304; pointers come from DeviceMemory-backed TimeSeries
    void perform (const dsp::TimeSeries* in, dsp::TimeSeries* out,
                  uint64_t npart, const uint64_t in_step, const uint64_t out_step)
    {
      const float* ibase = in->get_datptr (0, 0);
      const uint64_t ics = in->get_nchan () > 1 ? in->get_datptr (1, 0) - ibase : 0;
      const uint64_t ips = in->get_npol () > 1 ? in->get_datptr (0, 1) - ibase : 0;
      float* obase = out ? out->get_datptr (0, 0) : 0;      // out == NULL: benchmark only (:265)
      const uint64_t ocs = out && out->get_nchan () > 1 ? out->get_datptr (1, 0) - obase : 0;
      const uint64_t ops = out && out->get_npol () > 1 ? out->get_datptr (0, 1) - obase : 0;
      check (ctx, dspsr_amd_filterbank_perform (fb, ibase, ics, ips, obase, ocs, ops, npart, in_step, out_step),
             "HIP::FilterbankEngine::perform");
    }

    void finish () { check (ctx, dspsr_amd_stream_sync (ctx), "HIP::FilterbankEngine::finish"); }

    //! optional side channel: 8-bit input straight from the BitSeries (fused unpack)
    void perform_raw (const int8_t* raw, int layout, float scale, dsp::TimeSeries* out, uint64_t npart, uint64_t out_step)
    {
      float* obase = out->get_datptr (0, 0);
      const uint64_t ocs = out->get_nchan () > 1 ? out->get_datptr (1, 0) - obase : 0;
      const uint64_t ops = out->get_npol () > 1 ? out->get_datptr (0, 1) - obase : 0;
      check (ctx, dspsr_amd_filterbank_perform_raw (fb, raw, layout, scale, obase, ocs, ops, npart, out_step),
             "HIP::FilterbankEngine::perform_raw");
    }

  protected:
    dspsr_amd_ctx* ctx;
    dspsr_amd_filterbank* fb;
  };

  //! dsp::Convolution::Engine (Convolution.h:158-167): the same library object with nchan_subband = 1,
  //! i.e. one forward FFT, response multiply and backward FFT of response->get_ndat() points per
  //! (channel, polarisation, part), all channels and parts batched in one launch group
  class ConvolutionEngine : public dsp::Convolution::Engine
  {
  public:
    ConvolutionEngine (dspsr_amd_ctx* _ctx) : ctx (_ctx), fb (0), in_step (0), out_step (0) { }
    ~ConvolutionEngine () { dspsr_amd_filterbank_destroy (fb); }

    void set_scratch (void*) { }                 // the library owns its scratch

    //! reads what CUDA::ConvolutionEngine::prepare reads (ConvolutionCUDA.cu:208-244)
    void prepare (dsp::Convolution* convolution)
    {
      if (!convolution->has_response ())
        throw Error (InvalidState, "HIP::ConvolutionEngine::prepare", "no response");
      const dsp::Response* response = convolution->get_response ();
      const dsp::TimeSeries* input = convolution->get_input ();
      dspsr_amd_filterbank_config cfg;
      cfg.nchan_subband = 1;
      cfg.freq_res = response->get_ndat ();                       // npt_bwd
      cfg.nfilt_pos = response->get_impulse_pos ();
      cfg.nfilt_neg = response->get_impulse_neg ();
      cfg.input_nchan = input->get_nchan ();
      cfg.npol = input->get_npol ();
      cfg.real_input = input->get_state () == Signal::Nyquist;    // CUFFT_R2C vs C2C (:219-222)
      cfg.max_parts = 0;
      cfg.force_four_pass = 0;
      cfg.fused_fold = DSPSR_AMD_FUSED_AUTO;
      const uint64_t nsamp_step = convolution->get_minimum_samples () - convolution->get_minimum_samples_lost ();
      in_step = nsamp_step * (cfg.real_input ? 1 : 2);             // floats between parts (Convolution.C:386)
      out_step = in_step;                                          // the output is written at the same float offset (:441)
      dspsr_amd_filterbank_destroy (fb); fb = 0;
      check (ctx, dspsr_amd_filterbank_create (ctx, &cfg, &fb), "HIP::ConvolutionEngine::prepare");
      check (ctx, dspsr_amd_filterbank_set_kernel (fb, response->get_datptr (0, 0),
               uint64_t (response->get_nchan ()) * response->get_ndat ()), "HIP::ConvolutionEngine::prepare");
    }

    //! ConvolutionCUDA.cu:552-800
    void perform (const dsp::TimeSeries* in, dsp::TimeSeries* out, unsigned npart)
    {
      if (npart == 0) return;
      const float* ibase = in->get_datptr (0, 0);
      const uint64_t ics = in->get_nchan () > 1 ? in->get_datptr (1, 0) - ibase : 0;
      const uint64_t ips = in->get_npol () > 1 ? in->get_datptr (0, 1) - ibase : 0;
      float* obase = out->get_datptr (0, 0);
      const uint64_t ocs = out->get_nchan () > 1 ? out->get_datptr (1, 0) - obase : 0;
      const uint64_t ops = out->get_npol () > 1 ? out->get_datptr (0, 1) - obase : 0;
      check (ctx, dspsr_amd_filterbank_perform (fb, ibase, ics, ips, obase, ocs, ops, npart, in_step, out_step),
             "HIP::ConvolutionEngine::perform");
    }

  protected:
    dspsr_amd_ctx* ctx;
    dspsr_amd_filterbank* fb;
    uint64_t in_step, out_step;
  };

  //! dsp::Detection::Engine (Detection.h:98-106)
  class DetectionEngine : public dsp::Detection::Engine
  {
  public:
    DetectionEngine (dspsr_amd_ctx* _ctx) : ctx (_ctx) { }

    void polarimetry (unsigned ndim, const dsp::TimeSeries* in, dsp::TimeSeries* out)
    {
      if (in->get_ndat () != out->get_ndat ())
        throw Error (InvalidParam, "HIP::DetectionEngine::polarimetry", "input ndat != output ndat");
      const float* ibase = in->get_datptr (0, 0);
      float* obase = out->get_datptr (0, 0);
      const int state = out->get_state () == Signal::Stokes ? DSPSR_AMD_STOKES : DSPSR_AMD_COHERENCE;
      check (ctx, dspsr_amd_detect_polarimetry (ctx, state, ndim, ibase,
               in->get_nchan () > 1 ? in->get_datptr (1, 0) - ibase : 0, in->get_datptr (0, 1) - ibase,
               obase, out->get_nchan () > 1 ? out->get_datptr (1, 0) - obase : 0,
               out->get_npol () > 1 ? out->get_datptr (0, 1) - obase : 0,
               in->get_nchan (), in->get_ndat ()), "HIP::DetectionEngine::polarimetry");
    }

    void square_law (const dsp::TimeSeries* in, dsp::TimeSeries* out)
    {
      const float* ibase = in->get_datptr (0, 0);
      float* obase = out->get_datptr (0, 0);
      check (ctx, dspsr_amd_detect_square_law (ctx, out->get_state () == Signal::Intensity, ibase,
               in->get_nchan () > 1 ? in->get_datptr (1, 0) - ibase : 0,
               in->get_npol () > 1 ? in->get_datptr (0, 1) - ibase : 0, obase,
               out->get_nchan () > 1 ? out->get_datptr (1, 0) - obase : 0,
               out->get_npol () > 1 ? out->get_datptr (0, 1) - obase : 0,
               in->get_nchan (), in->get_npol (), in->get_ndat ()), "HIP::DetectionEngine::square_law");
    }

  protected:
    dspsr_amd_ctx* ctx;
  };

  //! dsp::Fold::Engine (Fold.h:249-312), the twin of CUDA::FoldEngine (FoldCUDA.cu:31-198, dsp/FoldCUDA.h:27-80).
  //! The engine OWNS the device-resident PhaseSeries (get_profiles()): Fold::get_output() returns it, so
  //! Fold::prepare_output / zero / mixable and the hits[] bookkeeping of Fold::fold act on it (Fold.C:88-94,495-508,
  //! 722,781-785).  Its sums live in device memory (HIP::DeviceMemory); hits[] stay on the host, as with the reference's
  //! default hits_on_gpu = false.  fold() binds the library to that buffer (Fold::Engine::setup's output / output_span),
  //! synch() is CUDA::FoldEngine::synch + TransferPhaseSeriesCUDA (TransferPhaseSeriesCUDA.C:23-84).
  class FoldEngine : public dsp::Fold::Engine
  {
  public:
    FoldEngine (dspsr_amd_ctx* _ctx) : ctx (_ctx), fold_handle (0)
    {
      use_set_bins = true;                       // plan built inside the library (Fold.C:730-740)
      check (ctx, dspsr_amd_fold_create (ctx, &fold_handle), "HIP::FoldEngine");
      d_profiles = new dsp::PhaseSeries;
      d_profiles->set_memory (new DeviceMemory (ctx));
      synchronized = true;                       // no data on either the host or the device (FoldCUDA.cu:52-53)
    }
    ~FoldEngine () { dspsr_amd_fold_destroy (fold_handle); }

    void set_nbin (unsigned nbin)
    { nbin_hits.assign (nbin, 0); check (ctx, dspsr_amd_fold_set_nbin (fold_handle, nbin), "HIP::FoldEngine::set_nbin"); }
    void set_ndat (uint64_t ndat, uint64_t idat_start)
    { check (ctx, dspsr_amd_fold_set_ndat (fold_handle, ndat, idat_start), "HIP::FoldEngine::set_ndat"); }
    void set_bin (uint64_t idat, double ibin, double bins_per_samp)
    { check (ctx, dspsr_amd_fold_set_bin (fold_handle, idat, ibin, bins_per_samp), "HIP::FoldEngine::set_bin"); }
    //! the double recurrence of Fold.C:744-787 inside the library; Fold::fold adds get_bin_hits to hits[] (:733-736)
    uint64_t set_bins (double phi, double phase_per_sample, uint64_t ndat, uint64_t idat_start)
    {
      uint64_t folded = 0;
      check (ctx, dspsr_amd_fold_set_bins (fold_handle, phi, phase_per_sample, ndat, idat_start,
                                           nbin_hits.empty() ? 0 : &nbin_hits[0], &folded), "HIP::FoldEngine::set_bins");
      return folded;
    }
    uint64_t get_bin_hits (int ibin) { return nbin_hits[ibin]; }
    uint64_t get_ndat_folded () const { return dspsr_amd_fold_get_ndat_folded (fold_handle); }
    dsp::PhaseSeries* get_profiles () { return d_profiles; }

    void fold ()
    {
      setup ();                                  // Fold.C:968-1011: input, input_span, output, output_span, nchan, npol, ndim
      check (ctx, dspsr_amd_fold_bind_profile (fold_handle, output, output_span, nchan, npol, ndim, d_profiles->get_nbin ()),
             "HIP::FoldEngine::fold");
      // rows of the input are input_span floats apart, (ichan*npol + ipol)-th row, as fold1bin* index them
      check (ctx, dspsr_amd_fold_fold (fold_handle, input, uint64_t (npol) * input_span, input_span),
             "HIP::FoldEngine::fold");
      synchronized = false;                      // the device profile is ahead of the host copy (FoldCUDA.cu:689)
    }

    //! FoldCUDA.cu:127-152 + TransferPhaseSeriesCUDA.C:23-84: shape and attributes (hits[] included, both on the host),
    //! then the whole buffer, then wait
    void synch (dsp::PhaseSeries* out)
    {
      if (synchronized) return;
      out->internal_match (d_profiles);
      out->copy_configuration (d_profiles);
      check (ctx, dspsr_amd_copy (ctx, out->internal_get_buffer (), d_profiles->internal_get_buffer (),
                                  d_profiles->internal_get_size (), DSPSR_AMD_D2H), "HIP::FoldEngine::synch");
      check (ctx, dspsr_amd_stream_sync (ctx), "HIP::FoldEngine::synch");
      synchronized = true;
    }

    void zero () { get_profiles ()->zero (); }   // dsp/FoldCUDA.h:49: PhaseSeries::zero through its DeviceMemory

  protected:
    dspsr_amd_ctx* ctx;
    dspsr_amd_fold* fold_handle;
    Reference::To<dsp::PhaseSeries> d_profiles;
    std::vector<unsigned> nbin_hits;
  };
}

#endif
