// tools/branch_cost.hip -- what a branch costs ONE wave that has its SIMD to itself (the funnel / StockWatson kernels are
// exactly that: one chain per wave, the launch as long as its busiest chain).  Each variant runs the same 32 dependent
// v_fma_f64 per loop trip; variants add 8 branches per trip: taken forward branches, not-taken branches, exec-mask
// (s_and_saveexec + s_cbranch_execz) regions.  Build: hipcc -O3 --offload-arch=gfx950 tools/branch_cost.hip -o /tmp/branch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

#define FMA4 "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n"

template <int MODE>
__global__ void k(double* out, long long* cyc, int n, int flag) {
  double x = out[threadIdx.x], y = 0.999;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile(FMA4 : "+v"(x) : "v"(y));
      if (MODE == 1) {  // taken forward branch
        asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n s_nop 0\n1:\n" ::"s"(flag) : "scc");
      } else if (MODE == 2) {  // not taken
        asm volatile("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n s_nop 0\n1:\n" ::"s"(flag) : "scc");
      } else if (MODE == 3) {  // exec-mask region, all lanes enter
        asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n s_nop 0\n1:\n s_or_b64 exec, exec, s[20:21]\n" ::: "s20", "s21", "scc");
      } else if (MODE == 4) {  // VALU compare -> vcc -> branch (the form of `if (__any(cond))`), not taken
        asm volatile("v_cmp_gt_f64 vcc, %0, %0\n s_cbranch_vccnz 1f\n s_nop 0\n1:\n" ::"v"(x) : "vcc");
      } else if (MODE == 5) {  // the same, taken
        asm volatile("v_cmp_ge_f64 vcc, %0, %0\n s_cbranch_vccnz 1f\n s_nop 0\n1:\n" ::"v"(x) : "vcc");
      } else if (MODE == 6) {  // v_readlane -> SGPR -> VALU use
        asm volatile("v_readlane_b32 s20, %0, 3\n s_nop 3\n v_add_u32 %0, s20, %0\n" : "+v"(((int*)&x)[0])::"s20");
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double* d; long long* c;
  hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
  hipMemset(d, 0, 64 * 8);
  const int n = 20000;
  const char* nm[] = {"32 dependent v_fma_f64", "+ 8 taken s_cbranch_scc1", "+ 8 not-taken s_cbranch_scc1", "+ 8 saveexec/execz regions", "+ 8 v_cmp->vcc->branch not taken",
                      "+ 8 v_cmp->vcc->branch taken", "+ 8 v_readlane->SGPR->VALU"};
  for (int m = 0; m < 7; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (m) {
        case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, c, n, 0); break;
        case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, c, n, 0); break;
        case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, c, n, 0); break;
        case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, c, n, 0); break;
        case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, c, n, 0); break;
        case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, d, c, n, 0); break;
        case 6: hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, d, c, n, 0); break;
      }
      hipDeviceSynchronize();
    }
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-40s %8.1f cycles per trip (s_memtime units), %6.2f per FMA-equivalent slot\n", nm[m], (double)h / n, (double)h / n / 32.0);
    fflush(stdout);
  }
  return 0;
}
