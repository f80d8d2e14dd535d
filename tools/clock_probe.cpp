// Diagnostic: shader clock held under an FP64-MFMA loop vs a light VALU loop
// (s_memtime = shader cycles, s_memrealtime = 100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int mode>
__global__ __launch_bounds__(256) void probe(int iters, unsigned long long* out, double* sink) {
  d4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = sink[threadIdx.x] + threadIdx.x * 1e-3 + 1.0, y = sink[blockIdx.x + 7] + 1.0 + blockIdx.x * 1e-6;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    } else {
      x = x * 1.0000001 + y;
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
  sink[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x;
}
int main() {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 16 * 1024); hipMalloc(&sink, 256 * 256 * 8); hipMemset(sink, 0, 256 * 256 * 8);
  unsigned long long h[512];
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = mode == 0 ? 200000 : 2000000;
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, iters, out, sink);
      else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, iters, out, sink);
      hipDeviceSynchronize();
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      double c = 0, r = 0;
      for (int b = 0; b < 256; ++b) { c += h[2 * b]; r += h[2 * b + 1]; }
      printf("mode %s: %.0f cycles in %.1f us -> %.3f GHz; cycles per iteration %.1f\n", mode == 0 ? "fp64-mfma" : "valu",
             c / 256, r / 256 / 100.0, (c / 256) / (r / 256 * 10.0), c / 256 / iters);
    }
  return 0;
}
