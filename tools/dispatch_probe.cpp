// Diagnostic: how fast does the chip start 1024 small workgroups, depending on what they use?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
template <int THREADS, bool SCRATCH, bool LDS, bool BIGSGPR>
__global__ __launch_bounds__(THREADS) void probe(unsigned long long* out, double* sink, int dyn) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  __shared__ double lds[LDS ? 1232 : 1];
  double acc = 0.0;
  if (SCRATCH) {
    volatile double priv[10];
    for (int i = 0; i < 10; ++i) priv[i] = sink[i + threadIdx.x];
    acc += priv[dyn % 10];
  }
  if (LDS) { lds[threadIdx.x] = sink[threadIdx.x]; __syncthreads(); acc += lds[(threadIdx.x + dyn) % THREADS]; }
  while (__builtin_amdgcn_s_memrealtime() - t0 < 800) __builtin_amdgcn_s_sleep(4);  // 8 us of "work"
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
  sink[4096 + blockIdx.x * THREADS + threadIdx.x] = acc;
}
template <int THREADS, bool SCRATCH, bool LDS, bool BIGSGPR>
void run(const char* name, unsigned long long* out, double* sink) {
  std::vector<unsigned long long> h(2048);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<THREADS, SCRATCH, LDS, BIGSGPR>), dim3(1024), dim3(THREADS), 0, 0, out, sink, 3);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost);
  std::vector<long long> st(1024), en(1024);
  unsigned long long m = ~0ull;
  for (int b = 0; b < 1024; ++b) m = std::min(m, h[2 * b]);
  for (int b = 0; b < 1024; ++b) { st[b] = (long long)(h[2 * b] - m); en[b] = (long long)(h[2 * b + 1] - m); }
  std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
  printf("%-28s start p50 %lld p90 %lld max %lld ; last end %lld (10 ns units)\n", name, st[512], st[921], st[1023], en[1023]);
}
int main() {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 2048 * 8); hipMalloc(&sink, (4096 + 1024 * 256) * 8); hipMemset(sink, 0, (4096 + 1024 * 256) * 8);
  run<64, false, false, false>("64 thr plain", out, sink);
  run<64, true, false, false>("64 thr scratch", out, sink);
  run<64, false, true, false>("64 thr lds", out, sink);
  run<64, true, true, false>("64 thr scratch+lds", out, sink);
  run<128, false, false, false>("128 thr plain", out, sink);
  run<128, true, true, false>("128 thr scratch+lds", out, sink);
  run<256, true, true, false>("256 thr scratch+lds", out, sink);
  return 0;
}
