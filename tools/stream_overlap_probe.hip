// stream_overlap_probe.hip -- do kernels of two streams run concurrently on this box?  Two kernels of 128 workgroups each (half the
// chip's CUs), each spinning for a fixed number of cycles: one stream after the other vs two streams.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long cycles, int* out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main() {
  int* out; hipMalloc(&out, 4096 * sizeof(int));
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg : {64, 128, 256}) {
    for (int mode = 0; mode < 2; ++mode) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s1);
      for (int k = 0; k < 50; ++k) {
        hipLaunchKernelGGL(spin, dim3(wg), dim3(256), 0, s1, 100000ull, out);
        hipLaunchKernelGGL(spin, dim3(wg), dim3(256), 0, mode ? s2 : s1, 100000ull, out + 1024);
      }
      hipStreamSynchronize(s2);
      hipEventRecord(e1, s1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%d workgroups per kernel, 50 x 2 kernels of 100 k cycles, %s: %.2f ms\n", wg, mode ? "two streams" : "one stream ", ms);
    }
  }
  return 0;
}
