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
//
// DEFERRED MODE (HIP::Chain).  The three operations of the hot path are separate Engine calls in DSPSR, made in this order
// per block: Filterbank::Engine::perform (Filterbank.C:547-553), Detection::Engine::polarimetry (Detection.C:325-334),
// Fold::Engine::set_nbin / set_ndat / set_bins (Fold.C:724-741) and Fold::Engine::fold (Fold.C:817-829).  Run one by one
// ("eager") the channelised block makes three round trips through HBM.  When the adaptors of one pipeline thread share a
// HIP::Chain with set_deferred(true), perform() and polarimetry() only RECORD their arguments; FoldEngine::fold(), which
// arrives last and with the bin plan complete, launches dspsr_amd_filterbank_perform_fold: filterbank, detection and fold
// in one launch group, the detected time series never leaves the chip.  DSPSR's call order is untouched.  The chain falls
// back to eager execution, block by block, whenever the fusion would change what another caller sees:
//   * Filterbank::Engine::finish() (Operation::record_time, Filterbank.C:551,664) or any synch executes what is pending;
//   * Fold folds only a piece of the block (a sub-integration boundary inside it, Subint.h:234-309), a Fold input that is
//     not the Detection output, a detected shape other than (npol 2, ndim 2) / (npol 1, ndim 4), square-law detection;
//   * TimeSeriesEngine::copy_data_fpt reads the pending output (a second consumer that goes through an engine).
//   * more than one FoldEngine is registered on the chain (dspsr folds N pulsars from ONE detected series: every
//     fold[ifold]->set_input (to_fold), LoadToFold1.C:917-924,948-955,1206): the first fold() of a block executes the recorded
//     calls, every Fold then reads the detected series that really exists -- never fused;
//   * the Fold input carries zeroed (RFI-excised) samples with per-channel hits (Fold.C:853-866): eager, the engine counts
//     the hits from the data.
// What a fused block leaves behind is a detected TimeSeries that was NEVER WRITTEN.  Any later engine call that reads it
// before the next perform() -- a FoldEngine that was not registered on the chain, a second Detection, copy_data_fpt --
// throws Error (InvalidState) instead of returning stale data.  A recorded block that reaches the next perform() without
// having been folded cannot be executed any more (DSPSR has refilled its input).  The reference skips Fold for a block
// legitimately -- Subint<Fold>::transformation: `if (!divider.get_is_valid()) continue;` in front of Op::transformation()
// (Signal/Pulsar/dsp/Subint.h:270), e.g. data before the requested start -- and nobody else reads the intermediates in the
// pipelines deferred mode is for, so such a block is DROPPED and counted (get_dropped_blocks); set_drop_unfolded (false)
// makes it an Error (InvalidState) instead, for callers that want to hear about it.
// A consumer of the intermediate TimeSeries that bypasses the engines (dsp::Dump, a TransferCUDA to the host) cannot be
// seen from here: deferred mode is for pipelines in which Detection is the only reader of the Filterbank output and the
// Folds on the chain the only readers of the Detection output (dspsr's default fold pipeline); it is opt-in for that reason.
//
// RAW INPUT (FilterbankEngine::set_raw_input).  The reference unpacks 8-bit data to float32 in front of the Filterbank (4x
// the bytes).  The twin of dsp::TransferBitSeriesCUDA (Signal/General/TransferBitSeriesCUDA.C:23-70) hands the packed
// device block to the engine; perform() then reads the bytes instead of the floats whenever the TimeSeries it is given
// is exactly the unpacked image of that block (same first input sample and length: dspsr -overlap, i.e.
// config->input_buffering == false, LoadToFold1.C:813-821), and the float rows otherwise.
#ifndef DSPSR_AMD_ENGINES_H
#define DSPSR_AMD_ENGINES_H

#include "dsp/Memory.h"
#include "dsp/FilterbankEngine.h"   // dsp::Filterbank, dsp::Filterbank::Engine, dsp::Response, dsp::TimeSeries
#include "dsp/Convolution.h"        // dsp::Convolution, dsp::Convolution::Engine
#include "dsp/Detection.h"          // dsp::Detection::Engine
#include "dsp/Fold.h"               // dsp::Fold::Engine, dsp::PhaseSeries
#include "dsp/TScrunch.h"           // dsp::TScrunch::Engine
#include "dsp/FScrunch.h"           // dsp::FScrunch::Engine
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

  //! What the adaptors of one pipeline thread share in deferred mode (see the top of this file)
  class Chain : public Reference::Able
  {
  public:
    Chain (dspsr_amd_ctx* _ctx) : ctx (_ctx), deferred (false), fused_blocks (0), eager_blocks (0), dropped_blocks (0),
                                  nfold (0), drop_unfolded (true), consumed_fb (0), consumed_det (0)
    { fbk.pending = false; det.pending = false; fbk.raw = 0; }

    void set_deferred (bool flag) { if (!flag) flush (); deferred = flag; }
    bool get_deferred () const { return deferred; }

    //! FoldEngines of this pipeline thread (they register themselves); with more than one, fold() never fuses
    void register_fold () { nfold ++; }
    void unregister_fold () { if (nfold) nfold --; }
    unsigned get_nfold () const { return nfold; }

    //! accept that a recorded block which no Fold asked for is discarded at the next perform() (default: Error)
    void set_drop_unfolded (bool flag) { drop_unfolded = flag; }
    bool get_drop_unfolded () const { return drop_unfolded; }

    //! the fused launch group has consumed the recorded block: its intermediate TimeSeries were never written
    void mark_consumed () { consumed_fb = fbk.out; consumed_det = det.out; }
    void clear_consumed () { consumed_fb = 0; consumed_det = 0; }
    //! throws if `series` is such a never-written intermediate (a reader the fusion could not see)
    void require_written (const dsp::TimeSeries* series, const char* method) const
    {
      if (series && (series == consumed_fb || series == consumed_det))
        throw Error (InvalidState, method, "this TimeSeries is an intermediate of a block that the deferred HIP::Chain ran as one "
                     "fused launch group: it was never written (register every FoldEngine on the chain, or do not defer)");
    }

    //! blocks that went through the fused launch group / through the separate launches / recorded and never consumed
    uint64_t get_fused_blocks () const { return fused_blocks; }
    uint64_t get_eager_blocks () const { return eager_blocks; }
    uint64_t get_dropped_blocks () const { return dropped_blocks; }

    //! execute, as the separate launches DSPSR asked for, whatever has only been recorded
    void flush ()
    {
      if (fbk.pending)
      {
        fbk.pending = false;
        run_filterbank ();
        eager_blocks ++;
      }
      if (det.pending)
      {
        det.pending = false;
        run_detection ();
      }
    }
    //! the same if `series` is the output of a recorded call (a second reader)
    void flush_if (const dsp::TimeSeries* series)
    { if ((fbk.pending && series == fbk.out) || (det.pending && (series == det.out || series == det.in))) flush (); }
    //! a recorded block is still pending when the next one arrives: DSPSR has refilled its input, it cannot run any more
    void unfolded_block ()
    {
      fbk.pending = false; det.pending = false;
      if (!drop_unfolded)
        throw Error (InvalidState, "HIP::FilterbankEngine::perform", "the block recorded by the deferred HIP::Chain was never "
                     "folded and cannot be executed any more (its input has been refilled): another reader of the Filterbank / "
                     "Detection output would see stale data (strict mode, Chain::set_drop_unfolded (false))");
      dropped_blocks ++;
    }

    // ---- arguments recorded by FilterbankEngine::perform
    struct FilterbankCall
    {
      bool pending;
      dspsr_amd_filterbank* fb;
      const dsp::TimeSeries* in;
      dsp::TimeSeries* out;
      const float* ibase; uint64_t ics, ips;      // float input rows (device)
      float* obase; uint64_t ocs, ops;            // complex output rows (device)
      uint64_t npart, in_step, out_step;
      const int8_t* raw; int raw_layout; float raw_scale;   // raw != 0: the packed block holds the same samples
    } fbk;
    // ---- arguments recorded by DetectionEngine::polarimetry
    struct DetectionCall
    {
      bool pending;
      unsigned ndim; int state;
      const dsp::TimeSeries* in;
      dsp::TimeSeries* out;
      const float* ibase; uint64_t ics, ips;
      float* obase; uint64_t ocs, ops;
      unsigned nchan; uint64_t ndat;
    } det;

    void run_filterbank ()
    {
      if (fbk.raw)
        check (ctx, dspsr_amd_filterbank_perform_raw (fbk.fb, fbk.raw, fbk.raw_layout, fbk.raw_scale, fbk.obase, fbk.ocs, fbk.ops,
                                                     fbk.npart, fbk.out_step), "HIP::FilterbankEngine::perform");
      else
        check (ctx, dspsr_amd_filterbank_perform (fbk.fb, fbk.ibase, fbk.ics, fbk.ips, fbk.obase, fbk.ocs, fbk.ops, fbk.npart,
                                                 fbk.in_step, fbk.out_step), "HIP::FilterbankEngine::perform");
    }
    void run_detection ()
    {
      check (ctx, dspsr_amd_detect_polarimetry (ctx, det.state, det.ndim, det.ibase, det.ics, det.ips, det.obase, det.ocs,
                                                det.ops, det.nchan, det.ndat), "HIP::DetectionEngine::polarimetry");
    }

    dspsr_amd_ctx* ctx;
    bool deferred;
    uint64_t fused_blocks, eager_blocks, dropped_blocks;
    unsigned nfold;
    bool drop_unfolded;
    const dsp::TimeSeries* consumed_fb;           // intermediates of the last fused block (never written), until the next perform()
    const dsp::TimeSeries* consumed_det;
  };

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
    TimeSeriesEngine (dspsr_amd_ctx* _ctx, Chain* _chain = 0) : ctx (_ctx), to (0), chain (_chain) { }
    void prepare (dsp::TimeSeries* parent) { to = parent; }
    void prepare_buffer (unsigned) { }           // no staging buffer: rows are copied directly

    //! TimeSeriesCUDA.cu:75-200 (same-device case); copies ndat samples of every (chan, pol) row of
    //! `from`, starting at idat_start, to the start of the rows of the parent
    void copy_data_fpt (const dsp::TimeSeries* from, uint64_t idat_start = 0, uint64_t ndat = 0)
    {
      if (chain)
      {
        chain->flush_if (from); chain->flush_if (to);                // a reader of a recorded block: execute it first
        chain->require_written (from, "HIP::TimeSeriesEngine::copy_data_fpt");
      }
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
    Reference::To<Chain> chain;
  };

  //! dsp::Filterbank::Engine (FilterbankEngine.h:15-44)
  class FilterbankEngine : public dsp::Filterbank::Engine
  {
  public:
    FilterbankEngine (dspsr_amd_ctx* _ctx, Chain* _chain = 0)
      : ctx (_ctx), fb (0), chain (_chain), fused_fold (DSPSR_AMD_FUSED_AUTO), max_parts (0)
    { raw.ptr = 0; }
    ~FilterbankEngine () { if (chain) chain->fbk.pending = false; dspsr_amd_filterbank_destroy (fb); }

    //! before setup: DSPSR_AMD_FUSED_AUTO / _ALWAYS / _NEVER (dspsr_amd_filterbank_config::fused_fold) and the number of
    //! overlap-save parts per launch group (0 = library default; the scratch is sized for it)
    void set_fused_fold (int mode) { fused_fold = mode; }
    void set_max_parts (unsigned parts) { max_parts = parts; }

    //! Raw-input side channel, the hand-over a dsp::TransferBitSeriesCUDA twin makes per block: `raw_dev` = device copy of
    //! BitSeries::get_rawptr(), holding `ndat` time samples that start at input sample `input_sample`
    //! (BitSeries::get_input_sample, Kernel/Classes/dsp/BitSeries.h:83-86); layout / scale as dspsr_amd_filterbank_perform_raw.
    //! One shot: consumed (or discarded) by the next perform().
    void set_raw_input (const void* raw_dev, uint64_t ndat, int64_t input_sample, int layout, float scale)
    { raw.ptr = (const int8_t*) raw_dev; raw.ndat = ndat; raw.input_sample = input_sample; raw.layout = layout; raw.scale = scale; }

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
      cfg.max_parts = max_parts;
      cfg.force_four_pass = 0;
      cfg.fused_fold = fused_fold;
      const float* kernel = 0;
      uint64_t ncomplex = 0;
      if (chain) chain->flush ();
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
      // the packed block may stand in for the float rows only if `in` is exactly its unpacked image
      uint64_t step = 0;
      dspsr_amd_filterbank_sizes (fb, 0, 0, &step, 0);
      const bool use_raw = raw.ptr && in->get_input_sample () == raw.input_sample && in->get_ndat () == raw.ndat
                           && in_step == step * in->get_ndim ();
      const int8_t* rawp = use_raw ? raw.ptr : 0;
      raw.ptr = 0;
      if (chain && out)
      {
        Chain::FilterbankCall& c = chain->fbk;
        chain->clear_consumed ();                                    // `out` is about to be written (or recorded) anew
        if (c.pending) chain->unfolded_block ();                     // never folded (Subint.h:270): dropped and counted; Error in strict mode
        c.fb = fb; c.in = in; c.out = out;
        c.ibase = ibase; c.ics = ics; c.ips = ips; c.obase = obase; c.ocs = ocs; c.ops = ops;
        c.npart = npart; c.in_step = in_step; c.out_step = out_step;
        c.raw = rawp; c.raw_layout = raw.layout; c.raw_scale = raw.scale;
        if (chain->get_deferred ()) { c.pending = true; return; }
        chain->run_filterbank ();
        chain->eager_blocks ++;
        return;
      }
      if (rawp && out)
        check (ctx, dspsr_amd_filterbank_perform_raw (fb, rawp, raw.layout, raw.scale, obase, ocs, ops, npart, out_step),
               "HIP::FilterbankEngine::perform");
      else
        check (ctx, dspsr_amd_filterbank_perform (fb, ibase, ics, ips, obase, ocs, ops, npart, in_step, out_step),
               "HIP::FilterbankEngine::perform");
    }

    //! Filterbank.C:551,664 (Operation::record_time): what was only recorded runs now, as its own launches
    void finish ()
    {
      if (chain) chain->flush ();
      check (ctx, dspsr_amd_stream_sync (ctx), "HIP::FilterbankEngine::finish");
    }

    //! direct form of the side channel: 8-bit input straight from the BitSeries (fused unpack), always eager
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
    Reference::To<Chain> chain;
    int fused_fold;
    unsigned max_parts;
    struct { const int8_t* ptr; uint64_t ndat; int64_t input_sample; int layout; float scale; } raw;
  };

  //! Twin of dsp::TransferBitSeriesCUDA::transformation (Signal/General/TransferBitSeriesCUDA.C:23-70) with the hand-over
  //! added: the packed block goes host -> device as it is (asynchronous, on the context's stream) and is announced to the
  //! filterbank engine.  BitSeriesT = dsp::BitSeries (a template only so that this header does not pull in
  //! dsp/BitSeries.h); `device` must have been resized by the caller through a HIP::DeviceMemory
  //! (output->internal_match (input), as TransferBitSeriesCUDA::prepare does).
  template <class BitSeriesT>
  void transfer_bitseries (dspsr_amd_ctx* ctx, const BitSeriesT* host, BitSeriesT* device, FilterbankEngine* engine,
                           int layout, float scale)
  {
    check (ctx, dspsr_amd_copy (ctx, device->get_rawptr (), host->get_rawptr (), host->get_size (), DSPSR_AMD_H2D),
           "HIP::transfer_bitseries");
    engine->set_raw_input (device->get_rawptr (), host->get_ndat (), host->get_input_sample (), layout, scale);
  }

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
    DetectionEngine (dspsr_amd_ctx* _ctx, Chain* _chain = 0) : ctx (_ctx), chain (_chain), stokes (-1) { }

    //! Detection::Engine::polarimetry carries no output state, and an in-place Detection only sets it on the TimeSeries
    //! AFTER the engine call (Detection.C:113-138,203): tell the engine (Detection::set_output_state) -- otherwise the
    //! state is read from `out`, which is right for out-of-place use
    void set_output_state (Signal::State state) { stokes = state == Signal::Stokes ? 1 : 0; }

    void polarimetry (unsigned ndim, const dsp::TimeSeries* in, dsp::TimeSeries* out)
    {
      if (in->get_ndat () != out->get_ndat ())
        throw Error (InvalidParam, "HIP::DetectionEngine::polarimetry", "input ndat != output ndat");
      const float* ibase = in->get_datptr (0, 0);
      float* obase = out->get_datptr (0, 0);
      const int state = (stokes >= 0 ? stokes == 1 : out->get_state () == Signal::Stokes) ? DSPSR_AMD_STOKES : DSPSR_AMD_COHERENCE;
      if (chain)
      {
        Chain::DetectionCall& c = chain->det;
        if (c.pending) chain->flush ();
        chain->require_written (in, "HIP::DetectionEngine::polarimetry");
        c.ndim = ndim; c.state = state; c.in = in; c.out = out;
        c.ibase = ibase; c.ics = in->get_nchan () > 1 ? in->get_datptr (1, 0) - ibase : 0; c.ips = in->get_datptr (0, 1) - ibase;
        c.obase = obase; c.ocs = out->get_nchan () > 1 ? out->get_datptr (1, 0) - obase : 0;
        // (the planes of an in-place detection lie where the polarisation rows were, DetectionCUDA.cu:145-149)
        c.ops = in == out ? c.ips : (out->get_npol () > 1 ? out->get_datptr (0, 1) - obase : 0);
        c.nchan = in->get_nchan (); c.ndat = in->get_ndat ();
        // recorded only behind a recorded filterbank block whose output this is, in a shape the fused fold writes
        if (chain->fbk.pending && in == chain->fbk.out && (ndim == 2 || ndim == 4)) { c.pending = true; return; }
        chain->flush ();
        chain->run_detection ();
        return;
      }
      check (ctx, dspsr_amd_detect_polarimetry (ctx, state, ndim, ibase,
               in->get_nchan () > 1 ? in->get_datptr (1, 0) - ibase : 0, in->get_datptr (0, 1) - ibase,
               obase, out->get_nchan () > 1 ? out->get_datptr (1, 0) - obase : 0,
               in == out ? uint64_t (in->get_datptr (0, 1) - ibase) : (out->get_npol () > 1 ? out->get_datptr (0, 1) - obase : 0),
               in->get_nchan (), in->get_ndat ()), "HIP::DetectionEngine::polarimetry");
    }

    void square_law (const dsp::TimeSeries* in, dsp::TimeSeries* out)
    {
      if (chain) { chain->flush (); chain->require_written (in, "HIP::DetectionEngine::square_law"); }
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
    Reference::To<Chain> chain;
    int stokes;
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
    FoldEngine (dspsr_amd_ctx* _ctx, Chain* _chain = 0)
      : ctx (_ctx), fold_handle (0), chain (_chain), plan_ndat (0), plan_idat_start (0), d_hits (0), d_hits_size (0)
    {
      use_set_bins = true;                       // plan built inside the library (Fold.C:730-740)
      check (ctx, dspsr_amd_fold_create (ctx, &fold_handle), "HIP::FoldEngine");
      d_profiles = new dsp::PhaseSeries;
      d_profiles->set_memory (new DeviceMemory (ctx));
      synchronized = true;                       // no data on either the host or the device (FoldCUDA.cu:52-53)
      if (chain) chain->register_fold ();
    }
    ~FoldEngine ()
    {
      if (chain) chain->unregister_fold ();
      if (d_hits) dspsr_amd_free (ctx, d_hits);
      dspsr_amd_fold_destroy (fold_handle);
    }

    void set_nbin (unsigned nbin)
    { nbin_hits.assign (nbin, 0); check (ctx, dspsr_amd_fold_set_nbin (fold_handle, nbin), "HIP::FoldEngine::set_nbin"); }
    void set_ndat (uint64_t ndat, uint64_t idat_start)
    {
      plan_ndat = ndat; plan_idat_start = idat_start;
      check (ctx, dspsr_amd_fold_set_ndat (fold_handle, ndat, idat_start), "HIP::FoldEngine::set_ndat");
    }
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
      if (chain && chain->fbk.pending)
      {
        // Deferred mode: the filterbank and the detection of this block have only been recorded.  If this call folds the
        // WHOLE detected block (no sub-integration boundary inside it: Subint.h:234-309 would fold it in pieces) in a shape
        // the fused kernel writes, the three operations run as ONE launch group and the channelised data stay on the chip;
        // otherwise they run now as the separate launches DSPSR asked for and the fold below reads their output.
        const Chain::FilterbankCall& f = chain->fbk;
        const Chain::DetectionCall& d = chain->det;
        const dsp::TimeSeries* in = parent->get_input ();
        // (one Fold per chain only: a second Fold of the same detected series -- dspsr folding several pulsars,
        //  LoadToFold1.C:917-955,1206 -- needs the series itself; zeroed samples with per-channel hits are counted from the
        //  data by the eager kernel, Fold.C:853-866)
        const bool whole = d.pending && in == d.out && plan_idat_start == 0 && plan_ndat == in->get_ndat ()
                           && in->get_ndat () == d.ndat && ((npol == 2 && ndim == 2 && d.ndim == 2) || (npol == 1 && ndim == 4 && d.ndim == 4))
                           && chain->get_nfold () <= 1 && !(zeroed_samples && hits_nchan == nchan);
        if (whole)
        {
          chain->mark_consumed ();
          chain->fbk.pending = false; chain->det.pending = false;
          check (ctx, dspsr_amd_filterbank_perform_fold (f.fb, f.raw ? 0 : f.ibase, f.ics, f.ips, f.in_step, f.raw, f.raw_layout,
                                                        f.raw_scale, d.state, fold_handle, f.npart), "HIP::FoldEngine::fold");
          chain->fused_blocks ++;
          synchronized = false;
          return;
        }
        chain->flush ();
      }
      else if (chain)
        chain->flush ();
      if (chain) chain->require_written (parent->get_input (), "HIP::FoldEngine::fold");
      if (zeroed_samples && hits_nchan == nchan)
      {
        // the input carries zeroed (RFI-excised) samples: hits[] per channel, counted on the device from the data
        // (Fold.C:853-866; CUDA twin fold1bin*hits, FoldCUDA.cu:415-576,622 with hits_on_gpu).  The counts live in a device
        // buffer of this engine and are added to the host PhaseSeries' hits in synch()
        const uint64_t nhits = uint64_t (nchan) * d_profiles->get_nbin ();
        if (nhits != d_hits_size)
        {
          if (d_hits) check (ctx, dspsr_amd_free (ctx, d_hits), "HIP::FoldEngine::fold");
          void* p = 0;
          check (ctx, dspsr_amd_malloc (ctx, nhits * sizeof (uint32_t), &p), "HIP::FoldEngine::fold");
          check (ctx, dspsr_amd_zero (ctx, p, nhits * sizeof (uint32_t)), "HIP::FoldEngine::fold");
          d_hits = (uint32_t*) p; d_hits_size = nhits;
        }
        check (ctx, dspsr_amd_fold_fold_zeroed (fold_handle, input, uint64_t (npol) * input_span, input_span, d_hits),
               "HIP::FoldEngine::fold");
        synchronized = false;
        return;
      }
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
      if (d_hits && out->get_hits_nchan () * uint64_t (out->get_nbin ()) == d_hits_size)
        // zeroed samples: the per-channel counts made on the device (TransferPhaseSeriesCUDA with transfer_hits)
        check (ctx, dspsr_amd_copy (ctx, out->get_hits (0), d_hits, d_hits_size * sizeof (uint32_t), DSPSR_AMD_D2H),
               "HIP::FoldEngine::synch");
      check (ctx, dspsr_amd_stream_sync (ctx), "HIP::FoldEngine::synch");
      synchronized = true;
    }

    //! dsp/FoldCUDA.h:49: PhaseSeries::zero through its DeviceMemory (and the device hits of a zeroed input)
    void zero ()
    {
      get_profiles ()->zero ();
      if (d_hits) check (ctx, dspsr_amd_zero (ctx, d_hits, d_hits_size * sizeof (uint32_t)), "HIP::FoldEngine::zero");
    }

  protected:
    dspsr_amd_ctx* ctx;
    dspsr_amd_fold* fold_handle;
    Reference::To<dsp::PhaseSeries> d_profiles;
    std::vector<unsigned> nbin_hits;
    Reference::To<Chain> chain;
    uint64_t plan_ndat, plan_idat_start;
    uint32_t* d_hits;                            // zeroed samples: [nchan][nbin] counts on the device
    uint64_t d_hits_size;
  };

  //! dsp::TScrunch::Engine (Signal/General/dsp/TScrunch.h:61-69; the reference's twin: CUDA::TScrunchEngine, TScrunchCUDA.cu:204-300,
  //! wired by LoadToFITS.C:435): out[o] = in[o*sfactor] + in[o*sfactor + 1] + ..., added in that order, FPT rows of ndim 1 or 2.
  //! Out of place like the CUDA engine.  The ndat % sfactor samples the block leaves over are the buffering policy's
  //! (TScrunch.C:110-111 re-presents them); the engine call covers ndat / sfactor whole output samples.
  class TScrunchEngine : public dsp::TScrunch::Engine
  {
  public:
    TScrunchEngine (dspsr_amd_ctx* _ctx) : ctx (_ctx), carry (0), carry_floats (0) { }
    ~TScrunchEngine () { if (carry) dspsr_amd_free (ctx, carry); }

    void fpt_tscrunch (const dsp::TimeSeries* in, dsp::TimeSeries* out, unsigned sfactor)
    {
      if (in->get_ndim () != out->get_ndim ())
        throw Error (InvalidParam, "HIP::TScrunchEngine::fpt_tscrunch", "cannot handle input ndim=%u != output ndim=%u",
                     in->get_ndim (), out->get_ndim ());
      if (out == in)
        throw Error (InvalidParam, "HIP::TScrunchEngine::fpt_tscrunch", "only out-of-place transformation implemented");
      if (in->get_ndat () == 0)
        return;
      const unsigned nchan = in->get_nchan (), npol = in->get_npol (), ndim = in->get_ndim ();
      // (the C-ABI treats the rows as a stream; here every call starts a fresh one: nothing carried in, the left-over partial sums
      //  land in a scratch of [nchan][npol][ndim] floats and are dropped)
      const uint64_t need = uint64_t (nchan) * npol * ndim;
      if (need > carry_floats)
      {
        if (carry) dspsr_amd_free (ctx, carry);
        void* p = 0;
        check (ctx, dspsr_amd_malloc (ctx, need * sizeof (float), &p), "HIP::TScrunchEngine::fpt_tscrunch");
        carry = (float*) p; carry_floats = need;
      }
      const float* ibase = in->get_datptr (0, 0);
      float* obase = out->get_datptr (0, 0);
      uint32_t count = 0;
      uint64_t nout = 0;
      check (ctx, dspsr_amd_tscrunch_fpt (ctx, ibase, nchan > 1 ? in->get_datptr (1, 0) - ibase : 0, npol > 1 ? in->get_datptr (0, 1) - ibase : 0,
                                          obase, nchan > 1 ? out->get_datptr (1, 0) - obase : 0, npol > 1 ? out->get_datptr (0, 1) - obase : 0,
                                          nchan, npol, ndim, in->get_ndat (), sfactor, carry, &count, &nout),
             "HIP::TScrunchEngine::fpt_tscrunch");
    }

  protected:
    dspsr_amd_ctx* ctx;
    float* carry;
    uint64_t carry_floats;
  };

  //! dsp::FScrunch::Engine (Signal/General/dsp/FScrunch.h:56-64; CUDA::FScrunchEngine, FScrunchCUDA.cu:50-90): output channel c =
  //! input channels c*sfactor ... (c+1)*sfactor - 1 added in order, any ndim (the CUDA engine takes ndim 2 only), out of place
  class FScrunchEngine : public dsp::FScrunch::Engine
  {
  public:
    FScrunchEngine (dspsr_amd_ctx* _ctx) : ctx (_ctx) { }

    void fpt_fscrunch (const dsp::TimeSeries* in, dsp::TimeSeries* out, unsigned sfactor)
    {
      if (out == in)
        throw Error (InvalidParam, "HIP::FScrunchEngine::fpt_fscrunch", "only out-of-place transformation implemented");
      if (in->get_ndat () == 0)
        return;
      const unsigned nchan = in->get_nchan (), npol = in->get_npol ();
      const float* ibase = in->get_datptr (0, 0);
      float* obase = out->get_datptr (0, 0);
      check (ctx, dspsr_amd_fscrunch_fpt (ctx, ibase, nchan > 1 ? in->get_datptr (1, 0) - ibase : 0, npol > 1 ? in->get_datptr (0, 1) - ibase : 0,
                                          obase, nchan / sfactor > 1 ? out->get_datptr (1, 0) - obase : 0,
                                          npol > 1 ? out->get_datptr (0, 1) - obase : 0, nchan, npol,
                                          in->get_ndat () * uint64_t (in->get_ndim ()), sfactor),
             "HIP::FScrunchEngine::fpt_fscrunch");
    }

  protected:
    dspsr_amd_ctx* ctx;
  };
}

#endif
