// FUNCTIONAL MINIATURE of the few PSRCHIVE / dsp classes the adaptors touch -- test infrastructure, NOT a build of DSPSR.
// tests/test_host_adaptor.py type-checks dspsr_amd/host/dspsr_amd_engines.h against the REAL reference headers (with
// tests/psrchive_stub); these miniatures exist so that the same adaptor header can also be COMPILED AND RUN on the GPU box
// (where /root/reference does not exist): containers that allocate through dsp::Memory, and a Fold that makes the engine
// calls in the order of Signal/Pulsar/Fold.C.  Every member cites the reference declaration it mirrors.
#pragma once
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
enum ErrorCode { InvalidParam, InvalidState };
class Error {                                              // PSRCHIVE Error.h
public:
  Error (ErrorCode c, const char* method, const char* fmt, ...) : code (c)
  {
    char buf[1024];
    va_list ap; va_start (ap, fmt); vsnprintf (buf, sizeof buf, fmt, ap); va_end (ap);
    message = std::string (method) + ": " + buf;
  }
  Error& operator+= (const char* where) { message = std::string (where) + " <- " + message; return *this; }
  ErrorCode code;
  std::string message;
};
namespace Reference {                                       // PSRCHIVE Reference.h (no counting: tests leak on purpose)
  class Able { public: virtual ~Able () {} };
  template <class T> class To {
  public:
    To () : p (0) {}
    To (T* q) : p (q) {}
    To& operator= (T* q) { p = q; return *this; }
    T* operator-> () const { return p; }
    T& operator* () const { return *p; }
    operator T* () const { return p; }
    T* get () const { return p; }
  private:
    T* p;
  };
}
namespace Signal { enum State { Nyquist, Analytic, Intensity, PPQQ, Coherence, Stokes }; }   // PSRCHIVE Types.h
