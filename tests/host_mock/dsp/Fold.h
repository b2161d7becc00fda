// FUNCTIONAL MINIATURE (see ../Error.h): dsp::Fold::Engine exactly as declared in Signal/Pulsar/dsp/Fold.h:249-312, with
// Engine::setup as Signal/Pulsar/Fold.C:968-1011, and a Fold that makes the engine calls of Fold.C in their order:
// get_output :88-94, prepare_output :495-508, fold :718-741,792-829, get_result :123-135, reset :137-148.
#pragma once
#include <math.h>
#include "dsp/Memory.h"
namespace dsp {
  class Fold : public Reference::Able {
  public:
    class Engine;
    Fold () : input (0), output (new PhaseSeries), folding_nbin (0) {}
    void set_input (const TimeSeries* in) { input = in; }
    const TimeSeries* get_input () const { return input; }
    void set_engine (Engine* e);
    void set_nbin (unsigned n) { folding_nbin = n; }
    PhaseSeries* get_output () const;
    void prepare_output ()                                   // Fold.C:495-508 (first use or shape change: not mixable)
    {
      PhaseSeries* out = get_output ();
      if (out->get_nbin () != folding_nbin || out->get_nchan () != input->get_nchan () || out->get_npol () != input->get_npol ()
          || out->get_ndim () != input->get_ndim ()) {
        out->Observation::copy_configuration (input);
        out->resize (folding_nbin);
        out->zero ();
      }
    }
    void fold (double phi, double pfold, uint64_t idat_start, uint64_t ndat_fold);
    PhaseSeries* get_result () const;
    void reset ();
  protected:
    const TimeSeries* input;
    Reference::To<PhaseSeries> output;
    Reference::To<Engine> engine;
    unsigned folding_nbin;
  };
  class Fold::Engine : public Reference::Able {
  public:
    Engine () : use_set_bins (false), parent (0), synchronized (true) {}
    void set_parent (Fold* f) { parent = f; }
    virtual void set_nbin (unsigned nbin) = 0;
    virtual void set_bin (uint64_t idat, double ibin, double bins_per_samp) = 0;
    virtual uint64_t set_bins (double phi, double phase_per_sample, uint64_t ndat, uint64_t idat_start) = 0;
    bool use_set_bins;
    virtual uint64_t get_bin_hits (int ibin) = 0;
    virtual uint64_t get_ndat_folded () const = 0;
    virtual PhaseSeries* get_profiles () = 0;
    virtual void fold () = 0;
    virtual void synch (PhaseSeries*) = 0;
    virtual void zero () = 0;
    virtual void set_ndat (uint64_t, uint64_t) {}
  protected:
    float* output; unsigned output_span;
    const float* input; unsigned input_span;
    unsigned* hits; unsigned hits_nchan; bool zeroed_samples;
    unsigned ndat_fold; uint64_t idat_start;
    unsigned nchan, npol, ndim;
    void setup ()                                            // Fold.C:968-1011
    {
      if (!parent) throw Error (InvalidState, "dsp::Fold::Engine::setup", "no parent");
      const TimeSeries* in = parent->get_input ();
      nchan = in->get_nchan (); npol = in->get_npol (); ndim = in->get_ndim ();
      input = in->get_datptr (0, 0);
      input_span = (unsigned) in->get_nfloat_span ();
      PhaseSeries* out = get_profiles ();
      output = out->get_datptr (0, 0);
      output_span = (unsigned) out->get_nfloat_span ();
      hits = out->get_hits ();
      hits_nchan = out->get_hits_nchan ();
      zeroed_samples = in->get_zeroed_data ();
    }
    Fold* parent;
    bool synchronized;
  };
  inline void Fold::set_engine (Engine* e) { engine = e; e->set_parent (this); }
  inline PhaseSeries* Fold::get_output () const { return engine ? engine->get_profiles () : output.get (); }
  inline void Fold::fold (double phi, double pfold, uint64_t idat_start, uint64_t ndat_fold)   // Fold.C:718-741,792-829
  {
    const double sampling_interval = 1.0 / input->get_rate ();
    const double phase_per_sample = sampling_interval / pfold;
    unsigned* hits = get_output ()->get_hits ();
    uint64_t ndat_folded = 0;
    engine->set_nbin (folding_nbin);
    engine->set_ndat (ndat_fold, idat_start);
    if (engine->use_set_bins) {
      ndat_folded = engine->set_bins (phi, phase_per_sample, ndat_fold, idat_start);
      for (unsigned ibin = 0; ibin < folding_nbin; ibin++) hits[ibin] += engine->get_bin_hits (ibin);
    } else {
      const double double_nbin = double (folding_nbin);
      for (uint64_t idat = idat_start; idat < idat_start + ndat_fold; idat++) {
        phi -= floor (phi);
        const double double_ibin = phi * double_nbin;
        const unsigned ibin = unsigned (double_ibin);
        phi += phase_per_sample;
        engine->set_bin (idat, double_ibin, phase_per_sample * double_nbin);
        hits[ibin]++;
        ndat_folded++;
      }
    }
    PhaseSeries* result = get_output ();
    result->integration_length += double (ndat_folded) / input->get_rate ();
    result->ndat_total += ndat_fold;
    if (result->get_nbin () != folding_nbin)
      throw Error (InvalidParam, "dsp::Fold::fold", "folding_nbin != output->nbin (%d != %d)", folding_nbin, result->get_nbin ());
    engine->fold ();
  }
  inline PhaseSeries* Fold::get_result () const { if (engine) engine->synch (output); return output; }     // Fold.C:123-135
  inline void Fold::reset () { if (engine) engine->zero (); if (output) output->zero (); }                   // Fold.C:137-148
}
