// FUNCTIONAL MINIATURE (see ../Error.h): dsp::Convolution / dsp::Convolution::Engine surface used by the adaptor
// (reference: Signal/General/dsp/Convolution.h:30-167).
#pragma once
#include "dsp/FilterbankEngine.h"
namespace dsp {
  class Convolution : public Reference::Able {
  public:
    class Engine;
    Convolution () : response (0), input (0), nsamp_fft (0), nsamp_overlap (0) {}
    const Response* get_response () const { return response; }
    bool has_response () const { return response != 0; }
    const TimeSeries* get_input () const { return input; }
    uint64_t get_minimum_samples () { return nsamp_fft; }
    uint64_t get_minimum_samples_lost () { return nsamp_overlap; }
    const Response* response;
    const TimeSeries* input;
    uint64_t nsamp_fft, nsamp_overlap;
  };
  class Convolution::Engine : public Reference::Able {
  public:
    virtual void set_scratch (void*) = 0;
    virtual void prepare (dsp::Convolution* convolution) = 0;
    virtual void perform (const TimeSeries* in, TimeSeries* out, unsigned npart) = 0;
  };
}
