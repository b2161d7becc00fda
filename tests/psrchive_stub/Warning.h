// STUB (see Reference.h)
#ifndef STUB_Warning_h
#define STUB_Warning_h
#include <iostream>
class Warning {
public:
  Warning ();
  template <class T> Warning& operator<< (const T&);
  Warning& operator<< (std::ostream& (*) (std::ostream&));
};
#endif
