// STUB (see Reference.h)
#ifndef STUB_sky_coord_h
#define STUB_sky_coord_h
class sky_coord { public: sky_coord (); };
#endif
