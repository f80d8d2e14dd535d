/* wn_hostrng.c -- the host side of walnuts_batch(): the callers' numpy Generators consumed in C.
 *
 * The reference's walnuts_step (walnuts/walnuts.py:322-359) draws from the caller's Generator in an order that depends on the data
 * (no uniform after a sub-U-turn :343, the categorical draw only after an accept :347-350), so the batched form keeps one exchange
 * per NUTS doubling with the device -- but the draws themselves need not be 4096 x 5 Python calls per doubling (round 4: 1.0e4
 * transitions/s, all of it interpreter time).  numpy ships the C implementations of its Generator methods (numpy/random/lib/
 * libnpyrandom.a, numpy/random/distributions.h) and every Generator exposes its bitgen_t (Generator.bit_generator.ctypes): the calls
 * below are the SAME functions Generator.normal / binomial / integers / uniform / random end in, on the same bit-generator state,
 * in the reference's order -- the streams stay bit-identical to per-chain Python calls (tests: test_walnuts_config1_draws_match_reference,
 * test_hostrng_matches_generator_methods).  Host code only; no GPU, no arithmetic of the sampler.
 *   gcc -O2 -shared -fPIC wn_hostrng.c -I<python include> -I<numpy include> <numpy>/random/lib/libnpyrandom.a -lm
 */
#include <stdint.h>
#include <string.h>

#include <numpy/random/bitgen.h>
#include <numpy/random/distributions.h>

/* rho[b][j] = L_mass[j] * rng_b.normal()  for j < D   (:325: rho = L_mass * rng.normal(size=D)) */
void wn_hr_momentum(void *const *bg, int32_t B, int32_t D, const double *l_mass, double *rho) {
  for (int32_t b = 0; b < B; ++b) {
    bitgen_t *g = (bitgen_t *)bg[b];
    for (int32_t j = 0; j < D; ++j) rho[(size_t)b * D + j] = l_mass[j] * random_normal(g, 0.0, 1.0);
  }
}

/* before an extension of n macro steps, every active chain: going_backward = bool(rng.binomial(1, 0.5)) (:330), then one
 * rng.choice([ell // 2, ell, ell * 2]) per macro step (:194, :256) = Generator.integers(0, 3) */
void wn_hr_extend_draws(void *const *bg, const int32_t *active, int32_t B, int32_t n, int32_t *backward, int32_t *lottery, uint64_t *scratch) {
  for (int32_t b = 0; b < B; ++b) {
    if (!active[b]) continue;
    bitgen_t *g = (bitgen_t *)bg[b];
    binomial_t bs;
    memset(&bs, 0, sizeof bs);
    backward[b] = random_binomial(g, 0.5, 1, &bs) != 0;
    random_bounded_uint64_fill(g, 0, 2, n, 0, scratch);
    for (int32_t m = 0; m < n; ++m) lottery[(size_t)b * n + m] = (int32_t)scratch[m];
  }
}

/* after the extension, every active chain without a sub-U-turn (:343): u[b] = rng.uniform(0, 1) (:346; the caller takes the logarithm
 * with numpy, as the reference does) */
void wn_hr_accept_uniforms(void *const *bg, const int32_t *active, const int32_t *sub, int32_t B, double *u) {
  for (int32_t b = 0; b < B; ++b)
    if (active[b] && !sub[b]) u[b] = random_uniform((bitgen_t *)bg[b], 0.0, 1.0);
}

/* after an accept (:347-350): the uniform of Generator.choice(orbit_ext, p=p) = rng.random() */
void wn_hr_choice_uniforms(void *const *bg, const int32_t *accept, int32_t B, double *u) {
  for (int32_t b = 0; b < B; ++b)
    if (accept[b]) u[b] = random_standard_uniform((bitgen_t *)bg[b]);
}
