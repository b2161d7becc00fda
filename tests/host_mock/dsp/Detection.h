// FUNCTIONAL MINIATURE (see ../Error.h): dsp::Detection::Engine (Signal/General/dsp/Detection.h:98-106)
#pragma once
#include "dsp/Memory.h"
namespace dsp {
  class Detection : public Reference::Able {
  public:
    class Engine;
  };
  class Detection::Engine : public Reference::Able {
  public:
    virtual void polarimetry (unsigned ndim, const TimeSeries* in, TimeSeries* out) = 0;
    virtual void square_law (const TimeSeries* in, TimeSeries* out) = 0;
  };
}
