// wn_launch.h -- internal launcher interface between the C-ABI (wn_capi.hip) and the kernel
// translation units (wn_k_lane.hip, wn_k_coop.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/walnuts_amd.h"  // WN_ROUTE_*
#include "wn_chain.h"

namespace wn {

struct KernelChoice {
  int layout;  // 0 = lane, 1 = coop, 2 = wave (persistent), 3 = wave split (gradient outside the chain kernel)
  int p0;      // lane: DMAX ; coop: k-steps ; wave: NE
  int p1;      // lane: target kind ; coop: CK_DENSE / CK_DIAG ; wave: target kind ; split: EXT_*
  int W;       // coop: waves per group
  int G;       // wave layout: workgroups per chain (cluster), 1 otherwise
  unsigned route;  // WN_ROUTE_* bits of the engine's configuration (include/walnuts_amd.h)
  char name[96];  // owned by the engine that holds this choice (no shared static storage)
};

// returns 0 and fills `kc` if a kernel exists for (target, D)
int choose_kernel(int target, int D, int waves_per_group, int integrator, int n_chains, unsigned route, KernelChoice* kc);
size_t kernel_work_doubles(const KernelChoice& kc, int C, int M);
hipError_t launch_lane(const KernelChoice& kc, int adapt /* 0 fixed, 1 adaptive, 2 adaptLeapFrogFlowD */, bool warm, const Params& prm, hipStream_t st);
hipError_t launch_coop(const KernelChoice& kc, int adapt /* 0 fixed, 1 adaptive, 2 adaptLeapFrogFlowD */, bool warm, const Params& prm, hipStream_t st);
hipError_t launch_wave(const KernelChoice& kc, int adapt /* 0 fixed, 1 adaptive, 2 adaptLeapFrogFlowD */, bool warm, const Params& prm, hipStream_t st);
hipError_t launch_split_begin(const KernelChoice& kc, int adapt /* 0 fixed, 1 adaptive, 2 adaptLeapFrogFlowD */, bool warm, const Params& prm, hipStream_t st);
hipError_t launch_split_tick(const KernelChoice& kc, int adapt /* 0 fixed, 1 adaptive, 2 adaptLeapFrogFlowD */, bool warm, const Params& prm, hipStream_t st);
hipError_t launch_dense_grad(const double* A, const double* xq, double* xg, int DP, int D, int Cpad, const int* done, int C,
                             hipStream_t st, int nj = 0);
// role-split form of the fixedLeapFrog coop path (wn_k_fast.hip)
bool fast_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm);
hipError_t launch_fast(const KernelChoice& kc, const Params& prm, int* abort_flag, hipStream_t st);
// wave-pair form of the same path for dense targets with D <= 100 (wn_k_fast4.hip): plain and traced sampling runs
bool fast4_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm);
hipError_t launch_fast4(const KernelChoice& kc, const Params& prm, int* abort_flag, hipStream_t st);
int fast4_pairs(int C);
// trajectory-first / batched-bookkeeping kernel of the one-chain-per-wave layout for <= 16-D targets (wn_k_orbit.hip)
bool orbit_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm);
hipError_t launch_orbit(const KernelChoice& kc, const Params& prm, hipStream_t st);
int orbit_wide_ne(const KernelChoice& kc);                      // registers per lane and vector of that mode
bool orbit_wide_geom(const KernelChoice& kc);                   // wn_k_orbit.hip's wide mode (StockWatson: vectors of 12 registers per lane)
size_t orbit_wide_work_doubles(int NE, int C, int M);
bool orbit_wide_run(const KernelChoice& kc, int integrator);   // ... and which integrators' plain sampling runs take it
bool orbit2_eligible(const KernelChoice& kc, int adapt, bool warm, const Params& prm);  // wn_k_orbit2.hip: that mode on two waves per chain (the three leapfrog integrators)
hipError_t launch_orbit2(const KernelChoice& kc, const Params& prm, int* kabort, hipStream_t st);
bool orbit2_run(const KernelChoice& kc, int integrator, int n_chains);  // (only while the chains leave SIMDs idle)
size_t orbit2_work_doubles(int NE, int C, int M);
size_t fast4_work_doubles(int C, int M);
size_t wave_work_doubles(int NE, int C, int M, int G);
bool wavelds_geom(const KernelChoice& kc);  // wn_k_wavelds.hip: dense Gaussian 65 <= D <= 128, adaptive integrators, one chain per wave, the matrix in LDS
hipError_t launch_wavelds(const KernelChoice& kc, int adapt, bool warm, const Params& prm, hipStream_t st);
size_t split_work_doubles(int NE, int C, int M);

}  // namespace wn
