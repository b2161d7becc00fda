// FUNCTIONAL MINIATURE (see ../Error.h): dsp::Response / dsp::Filterbank getters and dsp::Filterbank::Engine
// (Signal/General/dsp/FilterbankEngine.h:15-44, Filterbank.h, Response.h) as the engine's setup() reads them.
#pragma once
#include "dsp/Memory.h"
namespace dsp {
  class Response : public Reference::Able {
  public:
    Response () : impulse_pos (0), impulse_neg (0), nchan (1), ndat (1) {}
    unsigned get_impulse_pos () const { return impulse_pos; }
    unsigned get_impulse_neg () const { return impulse_neg; }
    unsigned get_nchan () const { return nchan; }
    unsigned get_ndat () const { return ndat; }
    const float* get_datptr (unsigned, unsigned) const { return kernel.empty () ? 0 : &kernel[0]; }
    unsigned impulse_pos, impulse_neg, nchan, ndat;
    std::vector<float> kernel;                              // nchan*ndat complex
  };
  class Filterbank : public Reference::Able {
  public:
    class Engine;
    Filterbank () : nchan_subband (1), freq_res (1), input (0), response (0), passband_cleared (false) {}
    void set_passband (Response*) { passband_cleared = true; }
    unsigned get_nchan_subband () const { return nchan_subband; }
    unsigned get_freq_res () const { return freq_res; }
    const TimeSeries* get_input () const { return input; }
    bool has_response () const { return response != 0; }
    const Response* get_response () const { return response; }
    unsigned nchan_subband, freq_res;
    const TimeSeries* input;
    const Response* response;
    bool passband_cleared;
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
