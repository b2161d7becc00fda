// MOCK (see ../Error.h): the virtual interface of dsp::Memory the adaptor overrides.
#pragma once
#include "Error.h"
namespace dsp {
  class Memory : public Reference::Able {
  public:
    virtual void* do_allocate (size_t nbytes) = 0;
    virtual void do_free (void*) = 0;
    virtual void do_zero (void*, size_t) = 0;
    virtual void do_copy (void* to, const void* from, size_t bytes) = 0;
    virtual bool on_host () const { return true; }
  };
  class Observation : public Reference::Able {
  public:
    unsigned get_nchan () const { return 1; }
    unsigned get_npol () const { return 2; }
    unsigned get_ndim () const { return 1; }
    uint64_t get_ndat () const { return 0; }
    Signal::State get_state () const { return Signal::Nyquist; }
  };
  class TimeSeries : public Observation {
  public:
    class Engine;
    float* get_datptr (unsigned = 0, unsigned = 0) { return 0; }
    const float* get_datptr (unsigned = 0, unsigned = 0) const { return 0; }
    void set_memory (Memory*) {}
  };
  class TimeSeries::Engine : public Reference::Able {      // Kernel/Classes/dsp/TimeSeries.h:211-223
  public:
    virtual void prepare (dsp::TimeSeries* to) = 0;
    virtual void prepare_buffer (unsigned nbytes) = 0;
    virtual void copy_data_fpt (const dsp::TimeSeries* copy, uint64_t idat_start = 0, uint64_t ndat = 0) = 0;
  };
  class PhaseSeries : public TimeSeries {
  public:
    unsigned get_nbin () const { return 0; }
  };
}
