// STUB (see Reference.h): PSRCHIVE's Signal namespace, the enumerations the dsp headers name
#ifndef STUB_Types_h
#define STUB_Types_h
#include <string>
namespace Signal {
  enum Dimension { Phase, Frequency, Polarization };
  enum Source { Unknown, Pulsar, PolnCal, FluxCalOn, FluxCalOff, Calibrator };
  enum State { Nyquist, Analytic, Intensity, NthPower, PPQQ, Coherence, Stokes, PseudoStokes, Invariant, Other, PP_State, QQ_State };
  enum Basis { Circular = 0, Linear = 1, Elliptical = 2 };
  enum Scale { EMF, Volts, Energy, Joules, FluxDensity, ReferenceFluxDensity, Jansky };
  enum Component { S0, S1, S2, S3, Inv, PP, QQ, RL, LR, PQ, QP, None };
  const char* state_string (State);
  const std::string State2string (State);
}
#endif
