// Can ONE wave hide the latency of a dependent vector-instruction chain behind its own FP64 MFMAs?  (round 6: the premise of running a tick's
// post-processing in the shadow of the next tick's MFMA phase.)   hipcc -O3 --offload-arch=gfx950 tools/mfma_interleave_probe.hip -o /tmp/probe && /tmp/probe
// Variants per wave, 75 x v_mfma_f64_4x4x4_4b_f64 (independent accumulators, 6 in rotation) and a chain of 150 dependent v_fma_f64:
//   A: the MFMAs, then the chain          B: one chain link behind every MFMA, then the rest          C: two chain links behind every MFMA
//   D: the chain as TWO independent chains of 75, one link of each behind every MFMA        E: MFMAs only        F: chain only
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ double mf(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
#define LINK(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2))
template <int V>
__global__ __launch_bounds__(64) void probe(double* out, unsigned long long* cyc, int reps) {
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
  double x = 1.0, y = 2.0;
  const double k1 = 0.999, k2 = 1e-3;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 75; ++i) {
      if (V != 5) acc[i % 6] = mf(a, b, acc[i % 6]);
      if (V == 1) LINK(x);
      if (V == 2) { LINK(x); LINK(x); }
      if (V == 3) { LINK(x); LINK(y); }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (V == 0 || V == 5) {
#pragma unroll
      for (int i = 0; i < 150; ++i) LINK(x);
    }
    if (V == 1) {
#pragma unroll
      for (int i = 0; i < 75; ++i) LINK(x);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + x + y;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&cyc, 8 * 1024);
  const int reps = 2000;
  const char* names[6] = {"A: 75 MFMA, then 150 dependent FMA", "B: 1 link behind every MFMA + 75 after", "C: 2 links (dependent) behind every MFMA",
                          "D: 2 links of two independent chains behind every MFMA", "E: MFMAs only", "F: chain only"};
  for (int v = 0; v < 6; ++v) {
    for (int warm = 0; warm < 2; ++warm) {
      switch (v) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, out, cyc, reps); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, out, cyc, reps); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, out, cyc, reps); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, out, cyc, reps); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, out, cyc, reps); break;
        case 5: hipLaunchKernelGGL(probe<5>, dim3(1), dim3(64), 0, 0, out, cyc, reps); break;
      }
      hipDeviceSynchronize();
    }
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f cycles per repetition (s_memtime-rate counter x ?: relative numbers matter)\n", names[v], (double)c / reps);
  }
  return 0;
}
