// FUNCTIONAL MINIATURE (see ../Error.h): dsp::TScrunch::Engine (Signal/General/dsp/TScrunch.h:61-69)
#pragma once
#include "dsp/Memory.h"
namespace dsp {
  class TScrunch : public Reference::Able {
  public:
    class Engine;
  };
  class TScrunch::Engine : public Reference::Able {
  public:
    virtual void fpt_tscrunch (const TimeSeries* in, TimeSeries* out, unsigned sfactor) = 0;
  };
}
