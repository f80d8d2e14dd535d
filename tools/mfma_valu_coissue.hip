// mfma_valu_coissue.hip -- does a vector-instruction stream issue beside FP64 MFMAs on one SIMD of this chip?
// Every SIMD of every CU gets two waves: wave A loops over dependent-free MFMAs (4x4x4_4b or 16x16x4), wave B over
// vector instructions (32-bit integer, DPP moves, or FP64 FMAs).  Times of A alone, B alone and both together.
// Also: ONE wave that alternates 1 MFMA + n vector instructions (what the sampler's T wave does).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int MK, int VK, int NV>  // MK: 0 none, 1 4x4x4, 2 16x16x4 (waves 0..3);  VK: 0 none, 1 int, 2 dpp, 3 fp64 (waves 4..7)
__global__ __launch_bounds__(512) void k2(double* out, int iters) {
  const int wave = threadIdx.x >> 6;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9, s = 0;
  if (wave < 4) {
    if (MK == 1) {
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 8; ++i) s += acc[i];
    } else if (MK == 2) {
      d4_t acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      s = acc[0][0] + acc[1][1];
    }
  } else {
    if (VK == 1) {
      unsigned x[8];
      for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8 * NV; ++i) x[i & 7] = x[i & 7] * 5u + 12345u;   // independent integer ops (v_mad_u32_u24 / v_mul_lo + add)
      for (int i = 0; i < 8; ++i) s += x[i];
    } else if (VK == 2) {
      int x[8];
      for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 265443 + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8 * NV; ++i) x[i & 7] = __builtin_amdgcn_update_dpp(x[i & 7], x[(i + 1) & 7], 0x128, 0xF, 0x5, false);
      for (int i = 0; i < 8; ++i) s += x[i];
    } else if (VK == 3) {
      double x[8];
      for (int i = 0; i < 8; ++i) x[i] = a + i;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8 * NV; ++i) x[i & 7] = x[i & 7] * b + a;
      for (int i = 0; i < 8; ++i) s += x[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave per SIMD: 1 MFMA (4x4x4) followed by NV vector instructions of kind VK
template <int VK, int NV>
__global__ __launch_bounds__(256) void k1(double* out, int iters) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9, s = 0;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int x[8]; double y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 265443 + i; y[i] = a + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (VK == 1) x[(i + j) & 7] = x[(i + j) & 7] * 5 + 12345;
        if (VK == 2) x[(i + j) & 7] = __builtin_amdgcn_update_dpp(x[(i + j) & 7], x[(i + j + 1) & 7], 0x128, 0xF, 0x5, false);
        if (VK == 3) y[(i + j) & 7] = y[(i + j) & 7] * b + a;
      }
    }
  }
  for (int i = 0; i < 8; ++i) s += acc[i] + x[i] + y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> static float timeit(F launch) {
  launch(50); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); launch(5000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * cus * 512);
  printf("%s: per loop trip wave A issues 8 small / 2 big MFMAs, wave B 8*NV vector instructions; 5000 trips\n", p.name);
#define T2(MK, VK, NV, what) { float ms = timeit([&](int it) { hipLaunchKernelGGL((k2<MK, VK, NV>), dim3(cus), dim3(512), 0, 0, out, it); }); \
    printf("  %-64s %7.3f ms = %6.1f ns per trip\n", what, ms, ms * 1e6 / 5000); }
  T2(1, 0, 1, "A: 8 x mfma 4x4x4 alone");
  T2(2, 0, 1, "A: 2 x mfma 16x16x4 alone");
  T2(0, 1, 2, "B: 16 int alone");
  T2(0, 2, 2, "B: 16 dpp alone");
  T2(0, 3, 2, "B: 16 fp64 fma alone");
  T2(1, 1, 2, "A 8 x 4x4x4 + B 16 int");
  T2(1, 2, 2, "A 8 x 4x4x4 + B 16 dpp");
  T2(1, 3, 2, "A 8 x 4x4x4 + B 16 fp64 fma");
  T2(1, 1, 4, "A 8 x 4x4x4 + B 32 int");
  T2(2, 1, 2, "A 2 x 16x16x4 + B 16 int");
  T2(2, 1, 4, "A 2 x 16x16x4 + B 32 int");
  T2(2, 3, 2, "A 2 x 16x16x4 + B 16 fp64 fma");
#define T1(VK, NV, what) { float ms = timeit([&](int it) { hipLaunchKernelGGL((k1<VK, NV>), dim3(cus), dim3(256), 0, 0, out, it); }); \
    printf("  %-64s %7.3f ms = %6.1f ns per trip (8 MFMAs)\n", what, ms, ms * 1e6 / 5000); }
  T1(0, 0, "one wave: 8 x (mfma 4x4x4)");
  T1(1, 1, "one wave: 8 x (mfma 4x4x4 + 1 int)");
  T1(1, 2, "one wave: 8 x (mfma 4x4x4 + 2 int)");
  T1(1, 3, "one wave: 8 x (mfma 4x4x4 + 3 int)");
  T1(2, 2, "one wave: 8 x (mfma 4x4x4 + 2 dpp)");
  T1(3, 1, "one wave: 8 x (mfma 4x4x4 + 1 fp64 fma)");
  T1(3, 2, "one wave: 8 x (mfma 4x4x4 + 2 fp64 fma)");
  return 0;
}
