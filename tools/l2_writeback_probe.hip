// tools/l2_writeback_probe.hip -- do repeated stores to the same lines reach the fabric once (write-back L2) or every time?
// Each of G workgroups (one per 4-chain group of the headline kernel's geometry) owns a block of BLOCK bytes; per "tick" it rewrites
// HOT bytes of it (the same lines every tick) and, every 16th tick, another COLD chunk.  rocprofv3 --pmc WRITE_SIZE tells what leaves the L2s.
//   hipcc -O3 --offload-arch=gfx950 tools/l2_writeback_probe.hip -o tools/l2_writeback_probe.bin
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- ./tools/l2_writeback_probe.bin <hot_bytes> <block_bytes> <ticks> <spin>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void probe(double* base, size_t block_doubles, int hot_doubles, int ticks, int spin) {
  double* blk = base + (size_t)blockIdx.x * block_doubles;
  const int lane = threadIdx.x;
  for (int t = 0; t < ticks; ++t) {
    for (int k = lane; k < hot_doubles; k += 64) blk[k] = (double)(t + k);
    if ((t & 15) == 15) {  // a cold chunk, visited once
      const size_t off = (size_t)hot_doubles + (size_t)(t >> 4) * 448;
      if (off + 448 <= block_doubles) for (int k = lane; k < 448; k += 64) blk[off + k] = (double)t;
    }
    for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);
  }
}

int main(int argc, char** argv) {
  const int hot = argc > 1 ? atoi(argv[1]) : 28672, block = argc > 2 ? atoi(argv[2]) : 99328, ticks = argc > 3 ? atoi(argv[3]) : 2000, spin = argc > 4 ? atoi(argv[4]) : 10;
  const int G = 1024;
  double* d;
  hipMalloc(&d, (size_t)G * block);
  hipMemset(d, 0, (size_t)G * block);
  hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, d, (size_t)block / 8, hot / 8, ticks, spin);
  hipDeviceSynchronize();
  printf("groups %d hot %d B block %d B ticks %d: stored %.3f GB in the hot region, footprint %.1f MB hot / %.1f MB total\n", G, hot, block, ticks,
         (double)G * hot * ticks / 1e9, (double)G * hot / 1e6, (double)G * block / 1e6);
  return 0;
}
