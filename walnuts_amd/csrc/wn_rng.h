// wn_rng.h -- the engine's counter-based random stream (host + device).
//
// The reference draws from NumPy's global MT19937 in a data-dependent order
// (WALNUTSpy/WALNUTS.py:4,216,236,298,395,426,613; adaptiveIntegrators.py:392 -- SURVEY.md Q1).
// A GPU engine cannot share one sequential stream between thousands of chains, so the engine
// defines its own stream and the parity tests replay it INTO the reference (tape):
//
//   Philox4x32-10, key = (seed_lo, seed_hi), counter = (block, iteration, global chain id, stream)
//   stream 0: uniforms.  The k-th uniform a chain consumes in iteration `it` (in the
//             reference's call order) is word-pair (k&1) of block k>>1:  u = (x >> 11) * 2^-53.
//   stream 1: momentum normals.  Element j of the refresh (WALNUTS.py:236) is Box-Muller output
//             (j&1) of block j>>1:  r = sqrt(-2 ln u1), u1 = ((x0>>12)+0.5) * 2^-52 in (0,1),
//             theta = 2 pi (x1>>11) 2^-53;  z = r cos(theta) (even j) or r sin(theta) (odd j).
//
// Keying by (chain, iteration) makes every chain's draws independent of how chains are
// sharded over GPUs and lets any lane regenerate any draw without state.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WN_HD __host__ __device__ __forceinline__
#else
#define WN_HD inline
#endif

namespace wn {

struct Philox4 {
  uint32_t w[4];
};

WN_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

WN_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                            uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    // one 32x32->64 multiply per product (v_mad_u64_u32 on gfx950) instead of mul_hi + mul_lo
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
  return o;
}

WN_HD double u53(uint32_t lo, uint32_t hi) {
  uint64_t x = ((uint64_t)hi << 32) | lo;
  return (double)(x >> 11) * 0x1.0p-53;
}

// k-th uniform of (chain, iter)
WN_HD double rng_uniform(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t k) {
  Philox4 p = philox4x32_10(k >> 1, iter, chain, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (k & 1u) ? u53(p.w[2], p.w[3]) : u53(p.w[0], p.w[1]);
}

// j-th momentum normal of (chain, iter).  Deliberately NOT inlined on the device: it is called
// once per element per iteration, and inlining log/sincos into the sampler's hot loop inflates its
// register allocation; as a real call only the rare call site pays.
#if defined(__HIPCC__)
#define WN_HD_NOINLINE __host__ __device__ __attribute__((noinline)) inline
#else
#define WN_HD_NOINLINE inline
#endif
WN_HD_NOINLINE double rng_normal(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t j) {
  Philox4 p = philox4x32_10(j >> 1, iter, chain, 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
  uint64_t x0 = ((uint64_t)p.w[1] << 32) | p.w[0];
  double u1 = ((double)(x0 >> 12) + 0.5) * 0x1.0p-52;
  double u2 = u53(p.w[2], p.w[3]);
  double r = sqrt(-2.0 * log(u1));
  double th = 6.283185307179586476925286766559 * u2;
  return (j & 1u) ? r * sin(th) : r * cos(th);
}

}  // namespace wn
