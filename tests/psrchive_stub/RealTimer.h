// STUB (see Reference.h)
#ifndef STUB_RealTimer_h
#define STUB_RealTimer_h
class RealTimer { public: RealTimer (); void start (); void stop (); double get_elapsed () const; double get_total () const; };
#endif
