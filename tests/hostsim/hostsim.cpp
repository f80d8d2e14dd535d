// hostsim.cpp -- CPU instantiation of the engine's chain state machine (walnuts_amd/csrc/
// wn_chain.h) with the one-chain-per-lane layout, one chain at a time.
//
// TEST TOOL ONLY: it lets the state-machine logic be checked against the oracle in a container
// without a GPU.  It is built by tests/ into tests/hostsim/ and is never loaded by the
// walnuts_amd package (the product path has no CPU fallback).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "wn_policy_lane.h"
#include "wn_targets.h"

using namespace wn;

template <int DMAX, class Tgt>
static void run_all(const Params& prm) {
  for (int c = 0; c < prm.C; ++c) {
    LanePolicy<DMAX, Tgt> p(prm, c);
    const bool warm = prm.warmupIter > 0;
    if (prm.integrator == I_ADAPT_FLOWD) {
      if (warm) { Engine<LanePolicy<DMAX, Tgt>, true, true, true> e(p, prm); e.run(); }
      else { Engine<LanePolicy<DMAX, Tgt>, true, false, true> e(p, prm); e.run(); }
    } else if (prm.integrator == I_FIXED) {
      if (warm) { Engine<LanePolicy<DMAX, Tgt>, false, true> e(p, prm); e.run(); }
      else { Engine<LanePolicy<DMAX, Tgt>, false, false> e(p, prm); e.run(); }
    } else {
      if (warm) { Engine<LanePolicy<DMAX, Tgt>, true, true> e(p, prm); e.run(); }
      else { Engine<LanePolicy<DMAX, Tgt>, true, false> e(p, prm); e.run(); }
    }
  }
}

extern "C" int hostsim_run(Params* prm, int target) {
  const int D = prm->D;
  size_t nw = 0;
  std::vector<double> work;
#define RUNIT(DMAX, TGT)                                              \
  {                                                                   \
    nw = LanePolicy<DMAX, TGT>::work_doubles(prm->C, prm->M);         \
    work.assign(nw, 0.0);                                             \
    prm->work = work.data();                                          \
    run_all<DMAX, TGT>(*prm);                                         \
    return 0;                                                         \
  }
  switch (target) {
    case T_STDGAUSS:
    case T_DIAGGAUSS:
      if (D <= 4) RUNIT(4, TgtDiag<4>)
      if (D <= 24) RUNIT(24, TgtDiag<24>)
      if (D <= 128) RUNIT(128, TgtDiag<128>)
      return -2;
    case T_DENSEGAUSS:
      if (D <= 8) RUNIT(8, TgtDense<8>)
      if (D <= 128) RUNIT(128, TgtDense<128>)
      return -2;
    case T_FUNNEL10: RUNIT(11, TgtFunnel10)
    case T_CORRGAUSS: RUNIT(2, TgtCorrGauss)
    case T_SMILE: RUNIT(2, TgtSmile)
    case T_FUNNEL1: RUNIT(2, TgtFunnel1)
    case T_MODFUNNEL: RUNIT(2, TgtModFunnel)
  }
  return -1;
}
extern "C" int hostsim_params_size() { return (int)sizeof(Params); }
