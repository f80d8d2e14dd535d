// What the split-mode gradient GEMM's epilogue costs (wn_k_gemm.hip: the four k-quarters of a 64 x 64 tile meet in LDS, then 8-byte stores
// with 16 rows per instruction): the same kernel with parts of the epilogue cut out (results then wrong: timing only), and with the tile
// stored in whole lines after a transposition through LDS.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/gemm_epi.hip -o tools/gemm_epi.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));

// MODE 0: as shipped; 1: no LDS reduction (each wave stores its own partial of its quarter: wrong sums); 2: no stores at all (a dummy one);
// 3: reduction + coalesced stores (the owner wave transposes its 16 x 64 block through LDS and writes 128-byte runs)
template <int MODE>
__global__ __launch_bounds__(256) void gemm_k(const double* __restrict__ A, const double* __restrict__ xq, double* __restrict__ xg,
                                              int DP, int Kr, int tiles_d, int tiles_c) {
  constexpr int NJ = 4;
  __shared__ double red[4 * 3 * 4 * NJ * 64];
  const int T = tiles_d * tiles_c, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int ct = t / tiles_d, dt = t % tiles_d;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d0 = dt * 64, c0 = ct * 16 * NJ;
  const int r = lane & 15;
  const int Kq = ((Kr + 3) / 4 + 15) / 16 * 16;
  const double* ap = A + (size_t)(d0 + r) * DP + wave * Kq + 2 * (lane >> 4);
  const double* bp = xq + (size_t)(c0 + r) * DP + wave * Kq + 2 * (lane >> 4);
  const size_t rs = (size_t)16 * DP;
  d4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (d4_t){0, 0, 0, 0};
  d2_t fa0[4], fb0[NJ], fa1[4], fb1[NJ];
#define LOAD(SET, KOFF)                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) fa##SET[i_] = *(const d2_t*)(ap + i_ * rs + (KOFF)); \
  _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) fb##SET[j_] = *(const d2_t*)(bp + j_ * rs + (KOFF));
#define MUL(SET)                                                                                         \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)      \
  _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                                      \
    acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa##SET[i_][s_], fb##SET[j_][s_], acc[i_][j_], 0, 0, 0);
  LOAD(0, 0);
  for (int k = 0; k < Kq; k += 16) {
    LOAD(1, k + 8);
    MUL(0);
    LOAD(0, (k + 16 < Kq ? k + 16 : k));
    MUL(1);
  }
  if (MODE == 2) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678) xg[0] = s;
    return;
  }
  if (MODE != 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i != wave) {
        const int slot = wave < i ? wave : wave - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) red[((i * 3 + slot) * 4 * NJ + j * 4 + q) * 64 + lane] = acc[i][j][q];
      }
    }
    __syncthreads();
  }
  // C/D layout: register q of lane l holds row 4*q + l/16 (output index within the 16-row block), column l%16 (chain)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i == wave) {
      double v[NJ][4];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double x = acc[i][j][q];
          if (MODE != 1) {
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) x += red[((i * 3 + sl) * 4 * NJ + j * 4 + q) * 64 + lane];
          }
          v[j][q] = -x;
        }
      if (MODE == 3) {
        // transpose the wave's 16 (d) x 64 (chains) block through LDS: write [chain][d] with d fastest, read rows of 16 d = 128 bytes
        __syncthreads();  // (everybody has read `red`)
        double* tb = red + wave * (64 * 17);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) tb[(16 * j + r) * 17 + 4 * q + (lane >> 4)] = v[j][q];
        // (same wave: LDS operations in order) lane l -> chain l, 16 consecutive d: four 32-byte pieces... write as 8 x 16 B per lane
        const int ch = lane;
        double* o = xg + (size_t)(c0 + ch) * DP + d0 + 16 * i;
#pragma unroll
        for (int d = 0; d < 16; d += 2) { d2_t w; w[0] = tb[ch * 17 + d]; w[1] = tb[ch * 17 + d + 1]; *(d2_t*)(o + d) = w; }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          double* o = xg + (size_t)(c0 + 16 * j + r) * DP + d0 + 16 * i + (lane >> 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) o[4 * q] = v[j][q];
        }
      }
    }
  }
}

template <int MODE> static double run(int D, int C, const std::vector<double>& h, double* chk) {
  const int DP = 1024, Cpad = C, reps = 200;
  double *A, *xq, *xg;
  hipMalloc(&A, sizeof(double) * DP * DP); hipMalloc(&xq, sizeof(double) * Cpad * DP); hipMalloc(&xg, sizeof(double) * Cpad * DP);
  hipMemset(A, 0, sizeof(double) * DP * DP); hipMemset(xq, 0, sizeof(double) * Cpad * DP);
  hipMemcpy2D(A, sizeof(double) * DP, h.data(), sizeof(double) * DP, sizeof(double) * D, D, hipMemcpyHostToDevice);
  hipMemcpy2D(xq, sizeof(double) * DP, h.data() + 12345, sizeof(double) * DP, sizeof(double) * D, Cpad, hipMemcpyHostToDevice);
  const int Kr = (D + 15) / 16 * 16, tiles_d = DP / 64, tiles_c = Cpad / 64, T = tiles_d * tiles_c, per = (T + 7) / 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gemm_k<MODE>, dim3(8 * per), dim3(256), 0, 0, A, xq, xg, DP, Kr, tiles_d, tiles_c);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_k<MODE>, dim3(8 * per), dim3(256), 0, 0, A, xq, xg, DP, Kr, tiles_d, tiles_c);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> o((size_t)Cpad * DP);
  hipMemcpy(o.data(), xg, sizeof(double) * o.size(), hipMemcpyDeviceToHost);
  double s = 0; for (int c = 0; c < Cpad; ++c) for (int d = 0; d < D; ++d) s += o[(size_t)c * DP + d] * (1 + ((c * 31 + d) % 7));
  *chk = s;
  hipFree(A); hipFree(xq); hipFree(xg);
  return ms * 1e3 / reps;
}
int main() {
  std::vector<double> h((size_t)1024 * 1024 + 20000);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  double c0, c1, c2, c3;
  const double t0 = run<0>(1000, 1024, h, &c0), t1 = run<1>(1000, 1024, h, &c1), t2 = run<2>(1000, 1024, h, &c2), t3 = run<3>(1000, 1024, h, &c3);
  printf("as shipped %.2f us (checksum %.10g) | no LDS reduction %.2f us | no epilogue %.2f us | reduction + whole-line stores %.2f us (checksum %.10g)\n", t0, c0, t1, t2, t3, c3);
  return 0;
}
