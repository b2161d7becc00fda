// MOCK for syntax-checking dspsr_amd/host/dspsr_amd_engines.h only (tests/test_host_adaptor.py).
// Declares just the names the adaptor uses from PSRCHIVE's Error.h / Reference.h; no behaviour.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
enum ErrorCode { InvalidParam, InvalidState };
class Error {
public:
  Error (ErrorCode, const char* /*method*/, const char* /*fmt*/, ...) {}
};
namespace Reference {
  class Able { public: virtual ~Able () {} };
  template <class T> class To {
  public:
    To () : p (0) {}
    To& operator= (T* q) { p = q; return *this; }
    T* operator-> () const { return p; }
    operator T* () const { return p; }
  private:
    T* p;
  };
}
namespace Signal { enum State { Nyquist, Analytic, Intensity, PPQQ, Coherence, Stokes }; }
