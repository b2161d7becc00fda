// MOCK (see ../Error.h): dsp::Convolution / dsp::Convolution::Engine surface used by the adaptor
// (reference: Signal/General/dsp/Convolution.h:30-167).
#pragma once
#include "dsp/FilterbankEngine.h"
namespace dsp {
  class Convolution : public Reference::Able {
  public:
    class Engine;
    const Response* get_response () const { return 0; }
    bool has_response () const { return false; }
    const TimeSeries* get_input () const { return 0; }
    uint64_t get_minimum_samples () { return 0; }
    uint64_t get_minimum_samples_lost () { return 0; }
  };
  class Convolution::Engine : public Reference::Able {
  public:
    virtual void set_scratch (void*) = 0;
    virtual void prepare (dsp::Convolution* convolution) = 0;
    virtual void perform (const TimeSeries* in, TimeSeries* out, unsigned npart) = 0;
  };
}
