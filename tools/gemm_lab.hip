// Variants of the split-mode gradient GEMM (wn_k_gemm.hip) side by side: row stride (leading dimension) and prefetch depth.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/gemm_lab.hip -o tools/gemm_lab.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));

// NSET register sets of 8 k each in rotation, NSET-1 in flight
template <int NSET>
__global__ __launch_bounds__(256) void gemm_k(const double* __restrict__ A, const double* __restrict__ xq, double* __restrict__ xg,
                                              int LD, int LDB, int Kr, int tiles_d, int tiles_c) {
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
  const double* ap = A + (size_t)(d0 + r) * LD + wave * Kq + 2 * (lane >> 4);
  const double* bp = xq + (size_t)(c0 + r) * LDB + wave * Kq + 2 * (lane >> 4);
  const size_t rs = (size_t)16 * LD, rsb = (size_t)16 * LDB;
  d4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (d4_t){0, 0, 0, 0};
  d2_t fa[NSET][4], fb[NSET][NJ];
  auto load = [&](int set, int koff) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[set][i] = *(const d2_t*)(ap + i * rs + koff);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[set][j] = *(const d2_t*)(bp + j * rsb + koff);
  };
  auto mul = [&](int set) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i][s], fb[set][j][s], acc[i][j], 0, 0, 0);
  };
  const int nst = Kq / 8;  // sets of this wave
#pragma unroll
  for (int s = 0; s < NSET - 1; ++s) load(s, (s < nst ? s : nst - 1) * 8);
  for (int k = 0; k < nst; k += NSET) {
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
      const int nx = k + s + NSET - 1;
      load((s + NSET - 1) % NSET, (nx < nst ? nx : nst - 1) * 8);
      if (k + s < nst) mul(s);
    }
  }
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
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i == wave) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        double* o = xg + (size_t)(c0 + 16 * j + r) * LDB + d0 + 16 * i + (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double v = acc[i][j][q];
#pragma unroll
          for (int sl = 0; sl < 3; ++sl) v += red[((i * 3 + sl) * 4 * NJ + j * 4 + q) * 64 + lane];
          o[4 * q] = -v;
        }
      }
    }
  }
}


// Round 6 variant: NO split of k inside the workgroup -- each of the four waves owns a 32 x 32 QUADRANT of the 64 x 64 tile over the whole k range
// (acc 2 x 2 MFMA tiles = 16 doubles per lane), NSET register sets of 8 k in rotation (4 x 16-byte loads and 8 MFMAs per set): no reduction of
// k-quarters through LDS, no barrier, no 98 KB of shared memory; every operand element is loaded by TWO waves (32 B/clk per CU instead of 16).
template <int NSET>
__global__ __launch_bounds__(256) void gemm_quad(const double* __restrict__ A, const double* __restrict__ xq, double* __restrict__ xg,
                                                 int LD, int LDB, int Kr, int tiles_d, int tiles_c) {
  const int T = tiles_d * tiles_c, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int ct = t / tiles_d, dt = t % tiles_d;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d0 = dt * 64 + 32 * (wave & 1), c0 = ct * 64 + 32 * (wave >> 1);
  const int r = lane & 15;
  const double* ap = A + (size_t)(d0 + r) * LD + 2 * (lane >> 4);
  const double* bp = xq + (size_t)(c0 + r) * LDB + 2 * (lane >> 4);
  const size_t rs = (size_t)16 * LD, rsb = (size_t)16 * LDB;
  d4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4_t){0, 0, 0, 0};
  d2_t fa[NSET][2], fb[NSET][2];
  auto load = [&](int set, int koff) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[set][i] = *(const d2_t*)(ap + i * rs + koff);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[set][j] = *(const d2_t*)(bp + j * rsb + koff);
  };
  auto mul = [&](int set) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i][s], fb[set][j][s], acc[i][j], 0, 0, 0);
  };
  const int nst = Kr / 8;
#pragma unroll
  for (int s = 0; s < NSET - 1; ++s) load(s, (s < nst ? s : nst - 1) * 8);
  for (int k = 0; k < nst; k += NSET) {
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
      const int nx = k + s + NSET - 1;
      load((s + NSET - 1) % NSET, (nx < nst ? nx : nst - 1) * 8);
      if (k + s < nst) mul(s);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      double* o = xg + (size_t)(c0 + 16 * j + r) * LDB + d0 + 16 * i + (lane >> 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) o[4 * q] = -acc[i][j][q];
    }
}

template <int NSET, bool QUAD = false> static double run(int LD, int LDB, int D, int C, const std::vector<double>& h, double* chk) {
  const int DP = 1024, Cpad = C, reps = 200;
  double *A, *xq, *xg;
  hipMalloc(&A, sizeof(double) * DP * LD); hipMalloc(&xq, sizeof(double) * Cpad * LDB); hipMalloc(&xg, sizeof(double) * Cpad * LDB);
  hipMemset(A, 0, sizeof(double) * DP * LD); hipMemset(xq, 0, sizeof(double) * Cpad * LDB);
  hipMemcpy2D(A, sizeof(double) * LD, h.data(), sizeof(double) * DP, sizeof(double) * D, D, hipMemcpyHostToDevice);
  hipMemcpy2D(xq, sizeof(double) * LDB, h.data() + 12345, sizeof(double) * DP, sizeof(double) * D, Cpad, hipMemcpyHostToDevice);
  const int Kr = (D + 15) / 16 * 16, tiles_d = DP / 64, tiles_c = Cpad / 64, T = tiles_d * tiles_c, per = (T + 7) / 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&]() { if (QUAD) hipLaunchKernelGGL(gemm_quad<NSET>, dim3(8 * per), dim3(256), 0, 0, A, xq, xg, LD, LDB, Kr, tiles_d, tiles_c); else hipLaunchKernelGGL(gemm_k<NSET>, dim3(8 * per), dim3(256), 0, 0, A, xq, xg, LD, LDB, Kr, tiles_d, tiles_c); };
  for (int i = 0; i < 10; ++i) go();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) go();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> o((size_t)Cpad * LDB);
  hipMemcpy(o.data(), xg, sizeof(double) * o.size(), hipMemcpyDeviceToHost);
  double s = 0; for (int c = 0; c < Cpad; ++c) for (int d = 0; d < D; ++d) s += o[(size_t)c * LDB + d] * (1 + ((c * 31 + d) % 7));
  *chk = s;
  hipFree(A); hipFree(xq); hipFree(xg);
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 1000, C = argc > 2 ? atoi(argv[2]) : 1024;
  std::vector<double> h((size_t)1024 * 1024 + 20000);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  {  // round 6: the quadrant variant (no k split inside the workgroup) against the shipped form, the shipped leading dimensions
    double c0, c1, c2, c3, c4;
    const double t0 = run<2>(1024, 1024, D, C, h, &c0);
    const double t1 = run<2, true>(1024, 1024, D, C, h, &c1), t2 = run<4, true>(1024, 1024, D, C, h, &c2), t3 = run<8, true>(1024, 1024, D, C, h, &c3), t4 = run<12, true>(1024, 1024, D, C, h, &c4);
    printf("shipped (k quarters + LDS reduction): %.2f us | quadrants, sets in rotation 2: %.2f  4: %.2f  8: %.2f  12: %.2f us (checksums %.10g %.10g %.10g %.10g %.10g)\n", t0, t1, t2, t3, t4, c0, c1, c2, c3, c4);
    fflush(stdout);
    if (argc > 3) return 0;
  }
  const int lds[][2] = {{1024, 1024}, {1040, 1024}, {1024, 1040}, {1040, 1040}, {1088, 1024}, {1024 + 32, 1024}, {1024+16, 1024+48}};
  for (auto& L : lds) {
    double c2, c3;
    const double t2 = run<2>(L[0], L[1], D, C, h, &c2), t3 = run<3>(L[0], L[1], D, C, h, &c3);
    printf("LDA=%d LDB=%d: sets 2: %.2f us  3: %.2f us (checksums %.10g %.10g)\n", L[0], L[1], t2, t3, c2, c3);
    fflush(stdout);
  }
  return 0;
}
