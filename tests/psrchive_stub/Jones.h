// STUB (see Reference.h)
#ifndef STUB_Jones_h
#define STUB_Jones_h
#include <complex>
template <typename T> class Jones {
public:
  std::complex<T> j00, j01, j10, j11;
  Jones (T scalar = 0.0);
};
#endif
