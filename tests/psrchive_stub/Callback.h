// STUB (see Reference.h)
#ifndef STUB_Callback_h
#define STUB_Callback_h
#include "Reference.h"
template <class Type> class Callback : public Reference::Able {
public:
  void send (const Type&);
  template <class Class, class Method> void connect (Class*, Method);
  template <class Class, class Method> void disconnect (Class*, Method);
};
#endif
