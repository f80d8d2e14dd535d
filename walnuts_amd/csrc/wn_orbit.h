// wn_orbit.h -- helpers shared by the trajectory-first kernels (wn_k_orbit.hip, wn_k_orbit2.hip): row scans and reductions on DPP for the
// lane-parallel bookkeeping (lane k = leaf k), scalar-branch helpers, the wide mode's workspace map.
#pragma once
#include "wn_policy_wave.h"

namespace wn {

namespace orbit {
typedef __attribute__((address_space(3))) double lds_d;
typedef __attribute__((address_space(3))) int lds_i;
// RS: doubles between consecutive ring entries -- 17, not 16: the R phase's check lanes read element d of DIFFERENT entries at once (lane c:
// entry r(c)), and at a stride of 16 doubles = 128 bytes all of them hit the same LDS banks (a 16-way conflict on every read)
enum { RS = 17, RQ = 0, RV = 16 * RS, DUMP = 32 * RS, COLDQ = DUMP + 64, COLDV = COLDQ + 16, COLDG = COLDV + 16, PROP = COLDG + 16, SLOT = PROP + 32 };
__host__ __device__ constexpr int lds_doubles(int M) { return SLOT + 32 * (M + 1); }
// wide mode: the chain's workspace, in vectors of NE * 64 doubles
enum { W_RQ = 0, W_RV = 16, W_COLDQ = 32, W_COLDV = 33, W_COLDG = 34, W_PROP = 35, W_SLOT = 37 };
__host__ __device__ constexpr size_t wide_doubles(int NE, int M) { return (size_t)(W_SLOT + 2 * (M + 1)) * NE * 64; }
template <class T_, class = void> struct tgt_mapped { static constexpr bool value = false; };
template <class T_> struct tgt_mapped<T_, typename std::enable_if<T_::MAPPED>::type> { static constexpr bool value = true; };

// row_shr:1 with zero fill at the first lane of each row of 16
__device__ __forceinline__ double shr1(double x) { return wn_dpp0<0x111>(x); }

template <int CTRL>
__device__ __forceinline__ int dppi(int x, int oob) { return __builtin_amdgcn_update_dpp(oob, x, CTRL, 0xF, 0xF, false); }
// reductions over each row of 16 lanes (every lane of a row gets its row's result)
__device__ __forceinline__ int rowsum_i(int x) {
  x += dppi<0xB1>(x, 0); x += dppi<0x4E>(x, 0); x += dppi<0x141>(x, 0); x += dppi<0x140>(x, 0);
  return x;
}
__device__ __forceinline__ int rowmin_i(int x) {
  x = min(x, dppi<0xB1>(x, x)); x = min(x, dppi<0x4E>(x, x)); x = min(x, dppi<0x141>(x, x)); x = min(x, dppi<0x140>(x, x));
  return x;
}
__device__ __forceinline__ double rowmin_d(double x) {  // fmin: a NaN operand is ignored (Hn < Hmin ? Hn : Hmin)
  x = fmin(x, wn_dpp<0xB1>(x)); x = fmin(x, wn_dpp<0x4E>(x)); x = fmin(x, wn_dpp<0x141>(x)); x = fmin(x, wn_dpp<0x140>(x));
  return x;
}
// serial prefix sum over a row, in order: after t steps lane k holds x_k + (prefix of lane k-1) for every k <= t; further steps change
// nothing.  Straight code (a loop pays a branch per step): 3 steps cover batches of up to 4 leaves, 15 every batch.
__device__ __forceinline__ double scan_steps(double s, double xs, int nbat) {
#pragma unroll
  for (int t = 0; t < 3; ++t) s = xs + shr1(s);
  if (nbat > 4) {
#pragma unroll
    for (int t = 0; t < 12; ++t) s = xs + shr1(s);
  }
  return s;
}
__device__ __forceinline__ int rl_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ int uni_i(int x) { return __builtin_amdgcn_readfirstlane(x); }
// a condition that is the same in every lane, as a SCALAR branch condition (v_cmp -> vcc -> s_cbranch_vccnz costs a lone wave 40-55 cycles,
// a scalar compare and branch 26-35: tools/branch_cost.hip)
__device__ __forceinline__ bool uni_b(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
}  // namespace orbit

}  // namespace wn
