// Issue cost of FP64 vector instruction FORMS on gfx950, one wave alone on its SIMD (round 6).  hipcc -O3 --offload-arch=gfx950 tools/valu_f64_rate_probe.hip -o /tmp/p && /tmp/p
// 8 independent registers in rotation (throughput) or one register (dependent chain), 512 instructions per repetition.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int V, bool DEP>
__global__ __launch_bounds__(64) void probe(double* out, unsigned long long* cyc, int reps) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0 + i + threadIdx.x * 1e-3;
  const double k1 = 0.999, k2 = 1e-3;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
#define IDX(j) (DEP ? 0 : j)
#define ONE(j)                                                                                                         \
  if (V == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[IDX(j)]) : "v"(k1), "v"(k2));                            \
  if (V == 1) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[IDX(j)]) : "v"(k1), "v"(k2));                           \
  if (V == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[IDX(j)]) : "v"(k2));                                         \
  if (V == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[IDX(j)]) : "v"(k1));                                         \
  if (V == 4) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(((int*)&x[IDX(j)])[0])); \
  if (V == 5) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(((int*)&x[IDX(j)])[0]) : "v"(((int*)&x[IDX(j)])[1]) : "vcc"); \
  if (V == 6) asm volatile("v_mov_b64_e32 %0, %1" : "+v"(x[IDX(j)]) : "v"(k1));                                         \
  if (V == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(((float*)&x[IDX(j)])[0]) : "v"((float)k1), "v"((float)k2));
      REP8(ONE)
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[threadIdx.x] = s;
}
template <int V> static void run(const char* nm, double* out, unsigned long long* cyc) {
  const int reps = 200; unsigned long long a, b;
  for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL((probe<V, false>), dim3(1), dim3(64), 0, 0, out, cyc, reps); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&a, cyc, 8, hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL((probe<V, true>), dim3(1), dim3(64), 0, 0, out, cyc, reps); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&b, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-22s independent %5.2f cycles each, dependent chain %5.2f\n", nm, (double)a / (reps * 512.0), (double)b / (reps * 512.0));
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 64);
  run<0>("v_fma_f64 (VOP3)", out, cyc); run<1>("v_fmac_f64_e32 (VOP2)", out, cyc); run<2>("v_add_f64", out, cyc); run<3>("v_mul_f64", out, cyc);
  run<4>("v_mov_b32_dpp", out, cyc); run<5>("v_cndmask_b32", out, cyc); run<6>("v_mov_b64", out, cyc); run<7>("v_fma_f32", out, cyc);
  return 0;
}
