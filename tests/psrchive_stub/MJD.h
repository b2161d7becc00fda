// STUB (see Reference.h)
#ifndef STUB_MJD_h
#define STUB_MJD_h
#include <string>
class MJD {
public:
  static const MJD zero;
  MJD (double = 0.0); MJD (int, int, double);
  int intday () const; double fracday () const; double in_days () const; double in_seconds () const;
  std::string printdays (unsigned) const;
  MJD& operator+= (const double&); MJD& operator-= (const double&);
  MJD& operator+= (const MJD&); MJD& operator-= (const MJD&);
};
const MJD operator+ (const MJD&, const MJD&); const MJD operator- (const MJD&, const MJD&);
const MJD operator+ (const MJD&, double); const MJD operator- (const MJD&, double);
int operator< (const MJD&, const MJD&); int operator> (const MJD&, const MJD&);
int operator<= (const MJD&, const MJD&); int operator>= (const MJD&, const MJD&);
int operator== (const MJD&, const MJD&); int operator!= (const MJD&, const MJD&);
#endif
