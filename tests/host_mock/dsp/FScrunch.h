// FUNCTIONAL MINIATURE (see ../Error.h): dsp::FScrunch::Engine (Signal/General/dsp/FScrunch.h:56-64)
#pragma once
#include "dsp/Memory.h"
namespace dsp {
  class FScrunch : public Reference::Able {
  public:
    class Engine;
  };
  class FScrunch::Engine : public Reference::Able {
  public:
    virtual void fpt_fscrunch (const TimeSeries* in, TimeSeries* out, unsigned sfactor) = 0;
  };
}
