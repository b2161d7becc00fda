// FUNCTIONAL MINIATURE (see ../Error.h): dsp::Memory (Kernel/Classes/dsp/Memory.h:18-34), dsp::Observation,
// dsp::DataSeries / dsp::TimeSeries (Kernel/Classes/dsp/DataSeries.h:30-140, TimeSeries.h:30-223) and dsp::PhaseSeries
// (Signal/Pulsar/dsp/PhaseSeries.h:28-205), reduced to what the Engine interfaces and Fold.C use.
#pragma once
#include "Error.h"
namespace dsp {
  class Memory : public Reference::Able {                   // host memory by default (Memory.C)
  public:
    virtual void* do_allocate (size_t nbytes) { return malloc (nbytes); }
    virtual void do_free (void* p) { free (p); }
    virtual void do_zero (void* p, size_t n) { memset (p, 0, n); }
    virtual void do_copy (void* to, const void* from, size_t n) { memcpy (to, from, n); }
    virtual bool on_host () const { return true; }
  };
  class Observation : public Reference::Able {              // Kernel/Classes/dsp/Observation.h
  public:
    Observation () : nchan (1), npol (1), ndim (1), ndat (0), state (Signal::Nyquist), rate (1.0) {}
    unsigned get_nchan () const { return nchan; }
    unsigned get_npol () const { return npol; }
    unsigned get_ndim () const { return ndim; }
    uint64_t get_ndat () const { return ndat; }
    Signal::State get_state () const { return state; }
    double get_rate () const { return rate; }
    void set_nchan (unsigned v) { nchan = v; }
    void set_npol (unsigned v) { npol = v; }
    void set_ndim (unsigned v) { ndim = v; }
    void set_state (Signal::State s) { state = s; }
    void set_rate (double r) { rate = r; }
    virtual void copy_configuration (const Observation* c)
    { nchan = c->nchan; npol = c->npol; ndim = c->ndim; state = c->state; rate = c->rate; }
  protected:
    unsigned nchan, npol, ndim;
    uint64_t ndat;
    Signal::State state;
    double rate;
  };
  class TimeSeries : public Observation {                   // DataSeries + TimeSeries: FPT rows behind one Memory
  public:
    class Engine;
    TimeSeries () : memory (new Memory), buffer (0), size (0), span (0) {}
    void set_memory (Memory* m) { memory = m; }
    const Memory* get_memory () const { return memory; }
    //! DataSeries::resize (DataSeries.C:107-195): rows of ndat*ndim floats, padded -- get_nfloat_span() != ndat*ndim
    virtual void resize (uint64_t nsamples)
    {
      ndat = nsamples;
      span = ((nsamples * ndim + 3) / 4) * 4 + 4;           // keeps rows 16-byte aligned, one float4 of slack
      const uint64_t need = uint64_t (nchan) * npol * span * sizeof (float);
      if (need > size) {
        if (buffer) memory->do_free (buffer);
        buffer = (unsigned char*) memory->do_allocate (need);
        size = need;
      }
    }
    float* get_datptr (unsigned ichan = 0, unsigned ipol = 0) { return (float*) buffer + (uint64_t (ichan) * npol + ipol) * span; }
    const float* get_datptr (unsigned ichan = 0, unsigned ipol = 0) const
    { return (const float*) buffer + (uint64_t (ichan) * npol + ipol) * span; }
    uint64_t get_nfloat_span () const { return span; }                            // TimeSeries.h:128
    unsigned char* internal_get_buffer () { return buffer; }                        // DataSeries.h:103-107
    const unsigned char* internal_get_buffer () const { return buffer; }
    uint64_t internal_get_size () const { return size; }
    virtual void internal_match (const TimeSeries* other)                          // TimeSeries.h:88
    { Observation::copy_configuration (other); resize (other->get_ndat ()); }
    virtual void zero () { if (buffer) memory->do_zero (buffer, size); }           // TimeSeries::zero
    bool get_zeroed_data () const { return zeroed_data; }                           // TimeSeries.h: set by the RFI excision
    void set_zeroed_data (bool z) { zeroed_data = z; }
    int64_t get_input_sample () const { return input_sample; }                     // TimeSeries.h:122-125
    void set_input_sample (uint64_t sample) { input_sample = (int64_t) sample; }
    //! TimeSeries::reshape (used by an in-place Detection, Detection.C:190-200): same buffer, other (npol, ndim) split
    void reshape (unsigned new_npol, unsigned new_ndim) { span = span * npol / new_npol; npol = new_npol; ndim = new_ndim; }
  protected:
    Reference::To<Memory> memory;
    unsigned char* buffer;
    uint64_t size, span;
    int64_t input_sample = -1;
    bool zeroed_data = false;
  };
  class BitSeries : public Observation {                    // Kernel/Classes/dsp/BitSeries.h:30-125: the packed block
  public:
    BitSeries () : memory (new Memory), data (0), size (0), input_sample (-1) {}
    void set_memory (Memory* m) { memory = m; }
    void resize (uint64_t nsamples, unsigned nbyte_per_sample)
    {
      ndat = nsamples;
      const uint64_t need = nsamples * nbyte_per_sample;
      if (need > size) { if (data) memory->do_free (data); data = (unsigned char*) memory->do_allocate (need); size = need; }
      used = need;
    }
    unsigned char* get_rawptr () { return data; }                                   // BitSeries.h:58-62
    const unsigned char* get_rawptr () const { return data; }
    uint64_t get_size () const { return used; }                                     // DataSeries.h: bytes in use
    int64_t get_input_sample (void* = 0) const { return input_sample; }             // BitSeries.h:83-86
    void set_input_sample (int64_t sample) { input_sample = sample; }
  protected:
    Reference::To<Memory> memory;
    unsigned char* data;
    uint64_t size, used = 0;
    int64_t input_sample;
  };
  class TimeSeries::Engine : public Reference::Able {      // Kernel/Classes/dsp/TimeSeries.h:211-223
  public:
    virtual void prepare (dsp::TimeSeries* to) = 0;
    virtual void prepare_buffer (unsigned nbytes) = 0;
    virtual void copy_data_fpt (const dsp::TimeSeries* copy, uint64_t idat_start = 0, uint64_t ndat = 0) = 0;
  };
  class PhaseSeries : public TimeSeries {                   // Signal/Pulsar/dsp/PhaseSeries.h
  public:
    PhaseSeries () : integration_length (0), ndat_total (0) {}
    unsigned get_nbin () const { return (unsigned) ndat; }
    void resize (uint64_t nbin) { TimeSeries::resize (nbin); hits.resize (nbin * hits_nchan, 0); }   // PhaseSeries.C:83-110
    unsigned* get_hits (unsigned ichan = 0) { return hits.empty () ? 0 : &hits[0] + uint64_t (ichan) * ndat; }
    unsigned get_hits_nchan () const { return hits_nchan; }
    void set_hits_nchan (unsigned n) { hits_nchan = n; }                                        // PhaseSeries.h: per-channel hits
    void zero () { integration_length = 0; ndat_total = 0; hits.assign (hits.size (), 0); TimeSeries::zero (); }  // PhaseSeries.C:239-256
    void copy_configuration (const Observation* c)                                              // PhaseSeries.C:258-318
    {
      TimeSeries::copy_configuration (c);
      const PhaseSeries* like = dynamic_cast<const PhaseSeries*> (c);
      if (like) { integration_length = like->integration_length; ndat_total = like->ndat_total; hits_nchan = like->hits_nchan; hits = like->hits; }
    }
    double integration_length;
    uint64_t ndat_total;
    unsigned hits_nchan = 1;
    std::vector<unsigned> hits;
  };
}
