// MOCK (see ../Error.h): dsp::Fold::Engine as declared in Signal/Pulsar/dsp/Fold.h:249-312 (names only).
#pragma once
#include "dsp/Memory.h"
namespace dsp {
  class Fold : public Reference::Able {
  public:
    class Engine;
    const TimeSeries* get_input () const { return 0; }
  };
  class Fold::Engine : public Reference::Able {
  public:
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
    void setup () {}
    Fold* parent;
    bool synchronized;
  };
}
