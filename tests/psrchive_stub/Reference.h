// STUB of a PSRCHIVE header, declarations only (PSRCHIVE is an external dependency of DSPSR and is not available here).
// Purpose: let tests/test_host_adaptor.py type-check dspsr_amd/host/dspsr_amd_engines.h against the REAL dsp/*.h headers
// of the reference tree.  Nothing here is linked or executed; only the names and signatures the dsp headers use exist.
#ifndef STUB_Reference_h
#define STUB_Reference_h
#include <stddef.h>
#include <stdint.h>
#include <iostream>
#include <string>
#include <vector>
namespace Reference {
  class Able { public: Able (); Able (const Able&); virtual ~Able (); };
  class HeapTracked { public: virtual ~HeapTracked (); };
  template <class T, bool active = true> class To {
  public:
    To (T* = 0); To (const To&);
    To& operator= (T*); To& operator= (const To&);
    T* operator-> () const; T& operator* () const;
    operator T* () const; operator bool () const; bool operator! () const;
    T* get () const; T* ptr () const; T* release (); void set (T*);
  };
}
#endif
