// mfma_f64_rate.hip -- measured FP64 matrix-core peak of the device: back-to-back v_mfma_f64_16x16x4_f64 with
// independent accumulators, every SIMD of every CU busy.  The denominator of bench.py's MFMA roofline
// (the MI355X micro-architecture guide has no FP64 row).  Also measures the rate with a DEPENDENT accumulator chain
// (the shape of the sampler's gradient: 25 MFMAs into one tile) at one and at two waves per SIMD, and the same loop with
// FP64 vector FMAs between the MFMAs (do the two share the arithmetic units?).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU>
__global__ void k(double* out, int iters) {
  d4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NVALU; ++j) x[j & 7] = x[j & 7] * b + a;
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int NVALU>
static void run(const char* what, int waves_per_simd) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 20000;
  dim3 grid(cus), block(256 * waves_per_simd);
  double* out;
  hipMalloc(&out, sizeof(double) * grid.x * block.x);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, NVALU>), grid, block, 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, NVALU>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)cus * 4 * waves_per_simd * iters * NACC;  // MFMAs issued
  const double tf = nm * 2048.0 / (ms * 1e-3) / 1e12;                  // 16*16*4*2 flop each
  printf("%-62s %d CUs x 4 SIMDs x %d wave(s): %8.3f ms  %7.2f TFLOP/s (matrix)  %.1f ns per MFMA per SIMD", what, cus, waves_per_simd, ms, tf,
         ms * 1e6 / (iters * NACC * waves_per_simd));
  if (NVALU) printf("  + %.2f TFLOP/s FP64 vector", (double)cus * 4 * waves_per_simd * iters * NACC * NVALU * 128.0 / (ms * 1e-3) / 1e12);
  printf("\n");
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  run<8, 0>("independent accumulators (8), MFMA only", 1);
  run<8, 0>("independent accumulators (8), MFMA only", 2);
  run<1, 0>("one dependent accumulator chain, MFMA only", 1);
  run<1, 0>("one dependent accumulator chain, MFMA only", 2);
  run<1, 4>("one dependent chain + 4 FP64 vector FMAs per MFMA", 1);
  run<1, 4>("one dependent chain + 4 FP64 vector FMAs per MFMA", 2);
  run<1, 12>("one dependent chain + 12 FP64 vector FMAs per MFMA", 2);
  run<8, 8>("independent accumulators + 8 FP64 vector FMAs per MFMA", 1);
  return 0;
}
