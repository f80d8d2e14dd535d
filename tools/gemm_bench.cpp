// Micro-benchmark of wn_dense_grad_kernel alone (diagnostic; links against libwalnuts_amd.so):
//   hipcc -O2 --offload-arch=gfx950 tools/gemm_bench.cpp -Lwalnuts_amd -lwalnuts_amd -o gpurun_out/gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
namespace wn {
hipError_t launch_dense_grad(const double* A, const double* xq, double* xg, int DP, int D, int Cpad, const int* done, int C,
                             hipStream_t st);
}
int main(int argc, char** argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 1000, C = argc > 2 ? atoi(argv[2]) : 1024, reps = 200;
  const int DP = (D <= 512) ? 512 : 1024, Cpad = (C + 63) / 64 * 64;
  double *A, *xq, *xg; int* done;
  hipMalloc(&A, sizeof(double) * DP * DP); hipMalloc(&xq, sizeof(double) * Cpad * DP); hipMalloc(&xg, sizeof(double) * Cpad * DP);
  hipMalloc(&done, 64); hipMemset(done, 0, 64);
  std::vector<double> h((size_t)(Cpad > DP ? Cpad : DP) * DP);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, h.data(), sizeof(double) * DP * DP, hipMemcpyHostToDevice);
  hipMemcpy(xq, h.data(), sizeof(double) * (size_t)Cpad * DP, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) wn::launch_dense_grad(A, xq, xg, DP, D, Cpad, done, C, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) wn::launch_dense_grad(A, xq, xg, DP, D, Cpad, done, C, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, fl = 2.0 * DP * (double)Cpad * ((D + 15) / 16 * 16);
  printf("D=%d C=%d: %.2f us per launch, %.1f TFLOP/s (padded flops), %.1f TFLOP/s (2*D*D*C)\n", D, C, us, fl / us * 1e-6,
         2.0 * D * D * C / us * 1e-6);
  return 0;
}
