// STUB (see Reference.h)
#ifndef STUB_OwnStream_h
#define STUB_OwnStream_h
#include <iostream>
#include "Reference.h"   // (PSRCHIVE headers reach ReferenceAble.h through one another)
class OwnStream {
public:
  OwnStream (); OwnStream (const OwnStream&); virtual ~OwnStream ();
  virtual void set_cerr (std::ostream&) const;
  virtual void set_cout (std::ostream&) const;
protected:
  mutable std::ostream cout;
  mutable std::ostream cerr;
};
#endif
