// MOCK (see ../Error.h): dsp::Response / dsp::Filterbank / dsp::Filterbank::Engine surface used by the adaptor.
#pragma once
#include "dsp/Memory.h"
namespace dsp {
  class Response : public Reference::Able {
  public:
    unsigned get_impulse_pos () const { return 0; }
    unsigned get_impulse_neg () const { return 0; }
    unsigned get_nchan () const { return 1; }
    unsigned get_ndat () const { return 1; }
    const float* get_datptr (unsigned, unsigned) const { return 0; }
  };
  class Filterbank : public Reference::Able {
  public:
    class Engine;
    void set_passband (Response*) {}
    unsigned get_nchan_subband () const { return 1; }
    unsigned get_freq_res () const { return 1; }
    const TimeSeries* get_input () const { return 0; }
    bool has_response () const { return false; }
    const Response* get_response () const { return 0; }
  };
  class Filterbank::Engine : public Reference::Able {
  public:
    Engine () { scratch = output = 0; }
    virtual void setup (Filterbank*) = 0;
    virtual void set_scratch (float*) = 0;
    virtual void perform (const TimeSeries* in, TimeSeries* out, uint64_t npart,
                          const uint64_t in_step, const uint64_t out_step) = 0;
    virtual void finish () {}
  protected:
    float* scratch;
    float* output;
    unsigned output_span;
  };
}
