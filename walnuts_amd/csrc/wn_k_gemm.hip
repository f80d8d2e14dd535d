// wn_k_gemm.hip -- the batched dense-Gaussian gradient of split mode (wn_k_split.hip):
//   xg[c][d] = -sum_k A[d][k] xq[c][k]      (A symmetric, zero-padded to DP x DP; xq, xg [Cpad][DP])
// as v_mfma_f64_16x16x4_f64 with M = output index d, N = chain, K = k.  Built with
// -mllvm -amdgpu-mfma-vgpr-form (Makefile): with accumulators homed in AGPRs the compiler moved all
// 128 of them through VGPRs on every trip of the k-loop.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "wn_launch.h"

namespace wn {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));

// Workgroup = 4 waves = one 64 x 64 output tile.  Every wave accumulates the WHOLE tile (16 MFMA
// tiles, 64 accumulator doubles per lane) over its own quarter of k, so each operand element is
// loaded once per workgroup and one 16-byte fragment load feeds eight MFMAs (with a 32 x 32 wave
// tile the vector-memory path, 64 B/clk/CU, was as busy as the matrix pipe).  Both operands are read
// "along k": lane l loads row l%16, k0 + 2*(l/16) .. +1, which serve two k-steps (the sum over k does
// not care which k sits in which MFMA slot as long as A and B agree).  Two register sets alternate,
// the set not being multiplied is in flight.  The four partial tiles meet in LDS at the end, each
// wave finishing and storing a quarter.
#define WN_GEMM_LOAD(SET, KOFF)                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) fa##SET[i_] = *(const d2_t*)(ap + i_ * rs + (KOFF));   \
  _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) fb##SET[j_] = *(const d2_t*)(bp + j_ * rs + (KOFF));
#define WN_GEMM_MUL(SET)                                                           \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                  \
  _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                 \
    acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa##SET[i_][s_], fb##SET[j_][s_], acc[i_][j_], 0, 0, 0);

// NJ = 16-chain blocks per tile: 4 (64 x 64 tiles), or 2 (64 x 32 tiles) when 64 x 64 tiles would leave
// CUs without a workgroup (measured, 1024 chains: D = 1000 44 vs 55 us -- the wider tile's register reuse
// wins once every CU has one; D = 300 15.6 vs 11.3 us -- 128 wide tiles fill half the chip).
// KS = 2: two workgroups share a tile, each with half of k, and ADD their halves into xg (which the chain kernel has zeroed:
// Params::xzero): a sub-batch of half the chains then still has a workgroup for every CU (wn_capi.hip, run_split_dense).  The sum of
// two numbers does not depend on which arrives first, so the result is as reproducible as the one-workgroup form's.
#ifndef WN_GEMM_XD
#define WN_GEMM_XD 2  // row-panel groups over the 8 XCDs (1 = the whole row range per XCD: round 1's contiguous ranges; 8 = the whole chain range per XCD).  Measured in the run, D = 1000 x 1024 chains, us per tick: 1: 51.61, 2: 51.29, 4: 51.20, 8: 51.04 (profiles/r06_gemm_xd_ab.txt): equal within the launch-to-launch spread; 2 fetches the fewest bytes (6 MB per XCD and launch against 9 MB for 1 and for 8: corrected FETCH_SIZE 48 MB per launch against 71 MB, profiles/r06_traffic_side.json)
#endif
template <int NJ, int KS = 1>
__global__ __launch_bounds__(256) void wn_dense_grad_kernel(const double* __restrict__ A, const double* __restrict__ xq,
                                                            double* __restrict__ xg, int DP, int Kr, int tiles_d,
                                                            int tiles_c, const int* done, int C) {
  if (*done >= C) return;  // every chain has finished: nothing reads the result
  __shared__ double red[4 * 3 * 4 * NJ * 64];  // [owner wave][partial accumulator of a contributor][lane]
  const int T = tiles_d * tiles_c * KS, per = (T + 7) / 8;
  // consecutive blocks go to different XCDs (blockIdx.x & 7): give each XCD a RECTANGLE of tiles, XD row-panel groups x 8/XD chain-panel
  // groups, so that its L2 (4 MB) holds what its 32 workgroups share.  D = 1000, 1024 chains (16 x 16 tiles): XD = 2 -> 8 row panels of A
  // (4 MB, each used by 4 workgroups) + 4 chain panels of xq (2 MB, each used by 8) per XCD = 6 MB fetched per XCD and launch; the
  // contiguous range of round 1 (2 chain panels x all 16 row panels = XD 1) streamed all 8 MB of A through every L2 and used each panel
  // twice: 9 MB; XD = 8: 2 row panels of A (1 MB, each used by 16) + all 16 chain panels (8 MB).  Measured (round 6): the tick moves by 1 %
  // between the best and the worst dealing -- the k loop is not waiting for L2 misses.  In bytes it needs (64 + 64) x 8 B per k for 64 clocks of
  // the four SIMDs' MFMA time = 16 B/clk per CU; in cache-LINE accesses (a 16-byte fragment load touches 16 lines and uses half of each) it is
  // within 2x of what the path delivers: without the k split (a 32 x 32 quadrant per wave over the whole k, twice the loads per MFMA, no
  // LDS reduction) the kernel takes 64-73 us alone against 44 (tools/gemm_lab.hip, round 6).  The 35 us of a launch in the run against 27 us
  // of MFMA time are its un-overlapped ends with one workgroup per CU.
  int ct, dt, ks = 0;
  const int xd = WN_GEMM_XD;
  if (KS == 1 && tiles_d % xd == 0 && tiles_c % (8 / xd) == 0) {
    const int nd = tiles_d / xd, nc = tiles_c / (8 / xd), x = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (j >= nd * nc) return;
    ct = (x / xd) * nc + j / nd;
    dt = (x % xd) * nd + j % nd;
  } else {
    const int tv = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tv >= T) return;
    const int t = tv / KS;
    ks = tv % KS;
    ct = t / tiles_d; dt = t % tiles_d;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d0 = dt * 64, c0 = ct * 16 * NJ;
  const int r = lane & 15;
  // this wave's k-range [wave*Kq, (wave+1)*Kq): a multiple of 16 long; beyond Kr both operands are zero
  const int Kq = ((Kr + 4 * KS - 1) / (4 * KS) + 15) / 16 * 16;
  const double* ap = A + (size_t)(d0 + r) * DP + (ks * 4 + wave) * Kq + 2 * (lane >> 4);   // + i * 16 rows
  const double* bp = xq + (size_t)(c0 + r) * DP + (ks * 4 + wave) * Kq + 2 * (lane >> 4);  // + j * 16 rows
  const size_t rs = (size_t)16 * DP;
  d4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (d4_t){0, 0, 0, 0};
  d2_t fa0[4], fb0[NJ], fa1[4], fb1[NJ];
  WN_GEMM_LOAD(0, 0);
  for (int k = 0; k < Kq; k += 16) {
    WN_GEMM_LOAD(1, k + 8);
    WN_GEMM_MUL(0);
    WN_GEMM_LOAD(0, (k + 16 < Kq ? k + 16 : k));
    WN_GEMM_MUL(1);
  }
  // wave w finishes the tiles with d-block i == w: the other waves hand it their partials
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i != wave) {
      const int slot = wave < i ? wave : wave - 1;  // 0..2 among the three contributors of owner i
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[((i * 3 + slot) * 4 * NJ + j * 4 + q) * 64 + lane] = acc[i][j][q];
    }
  }
  __syncthreads();
  // (round 5, tools/gemm_epi.hip: of the GEMM's 42.3 us alone, back to back, the k loop is 36.6 and this epilogue 5.7; storing whole 128-byte
  //  lines after a transposition through LDS makes it 40.0 alone -- and the RUN 2.2 % slower (1.916e7 -> 1.874e7 steps/s, one more barrier per
  //  tile): not taken.  The k loop's 36.6 us against 27.8 us of MFMA time is L2 bandwidth: a 64 x 64 tile loads 1 MB for 8.3 MFLOP, 7 TB/s chip-wide.)
  // C/D layout of v_mfma_f64_16x16x4_f64: register q of lane l holds row 4*q + l/16 (output index),
  // column l%16 (chain)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i == wave) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        double* o = xg + (size_t)(c0 + 16 * j + r) * DP + d0 + 16 * i + (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double v = acc[i][j][q];
#pragma unroll
          for (int sl = 0; sl < 3; ++sl) v += red[((i * 3 + sl) * 4 * NJ + j * 4 + q) * 64 + lane];
          if (KS == 1) o[4 * q] = -v;
          else unsafeAtomicAdd(o + 4 * q, -v);
        }
      }
    }
  }
}

hipError_t launch_dense_grad(const double* A, const double* xq, double* xg, int DP, int D, int Cpad, const int* done, int C,
                             hipStream_t st, int nj) {
  const int Kr = (D + 15) / 16 * 16;
  const int tiles_d = DP / 64;
#ifdef WN_DIAG
  static const int narrow_env = getenv("WN_GEMM_NJ") ? atoi(getenv("WN_GEMM_NJ")) : 0;  // diagnostic override
#else
  const int narrow_env = 0;
#endif
  // nj = 4: the caller runs several sub-batches side by side and wants one workgroup per CU from each (wn_capi.hip, run_split_dense)
  const bool narrow = (nj == 2 || nj == 4) ? nj == 2 : narrow_env ? narrow_env == 2 : (tiles_d * (Cpad / 64) < 256);
  if (nj == 42) {  // 64 x 64 tiles, k split over two workgroups
    const int tiles_c = Cpad / 64, T = tiles_d * tiles_c * 2, per = (T + 7) / 8;
    hipLaunchKernelGGL((wn_dense_grad_kernel<4, 2>), dim3(8 * per), dim3(256), 0, st, A, xq, xg, DP, Kr, tiles_d, tiles_c, done, C);
  } else if (narrow) {
    const int tiles_c = Cpad / 32, T = tiles_d * tiles_c, per = (T + 7) / 8;
    hipLaunchKernelGGL(wn_dense_grad_kernel<2>, dim3(8 * per), dim3(256), 0, st, A, xq, xg, DP, Kr, tiles_d, tiles_c, done, C);
  } else {
    const int tiles_c = Cpad / 64, T = tiles_d * tiles_c, per = (T + 7) / 8;
    hipLaunchKernelGGL(wn_dense_grad_kernel<4>, dim3(8 * per), dim3(256), 0, st, A, xq, xg, DP, Kr, tiles_d, tiles_c, done, C);
  }
  return hipGetLastError();
}

}  // namespace wn
