// STUB (see Reference.h)
#ifndef STUB_FTransformAgent_h
#define STUB_FTransformAgent_h
#include "Reference.h"
namespace FTransform {
  enum type { frc = 0, fcc = 1, bcc = 2, bcr = 3 };
  class Plan : public Reference::Able {
  public:
    virtual void frc1d (size_t, float*, const float*);
    virtual void fcc1d (size_t, float*, const float*);
    virtual void bcc1d (size_t, float*, const float*);
    virtual void bcr1d (size_t, float*, const float*);
  };
  class Plan2 : public Reference::Able { };
  class Bench : public Reference::Able { };
  class Agent : public Reference::Able {
  public:
    static Reference::To<Agent, false> current;
    virtual Plan* get_plan (size_t, type);
  };
}
#endif
