// tools/wave_tick_floor.hip -- how fast can ONE wavefront run the bare arithmetic of a funnel10 tick?
// One leapfrog step of the 11-D funnel (targetDistr.py:74-78) with its sums, the energy test, two U-turn checks
// against states in LDS, the multinomial weight (exp), one division and two uniforms per tick -- no state machine,
// no diagnostics, no adaptive passes: the floor under wn_wave_kernel<NE=1,funnel10> (1.9 us per tick).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/wave_tick_floor.hip -o /tmp/wtf && /tmp/wtf
// Measured (MI355X, round 2): 0.68 us per tick with one wave per SIMD (340 vector instructions: one wave alone nearly
// saturates its SIMD, 4.8 cycles per instruction), 1.96 us with four waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rowsum(double x) {  // lanes 0..15 hold data: sum inside the row, then broadcast from lane 0
  x += dpp<0xB1>(x); x += dpp<0x4E>(x); x += dpp<0x141>(x); x += dpp<0x140>(x);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 0), __builtin_amdgcn_readlane(__double2loint(x), 0));
}
__global__ __launch_bounds__(64) void tick_floor(double* out, const double* u, int nticks, double h0, double delta) {
  __shared__ double st[4][16];
  const int lane = threadIdx.x;
  double q = lane < 11 ? 0.1 * (lane + 1) : 0.0, v = lane < 11 ? 0.3 - 0.05 * lane : 0.0, g = 0.0;
  if (lane < 16) { st[0][lane] = q; st[1][lane] = v; st[2][lane] = q; st[3][lane] = v; }
  double H0 = 0.0, Wsum = 1.0, acc = 0.0;
  int picks = 0;
  for (int t = 0; t < nticks; ++t) {
    const double h = h0 * (0.9 + 0.2 * u[(2 * t) & 1023]);
    v += 0.5 * h * g; q += h * v;
    const double q0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(q)), __builtin_amdgcn_readfirstlane(__double2loint(q)));
    const double emq = exp(-q0);
    const double ss = rowsum((lane >= 1 && lane < 11) ? q * q : 0.0);
    g = lane == 0 ? (-5.0 - q0 / 9.0 + 0.5 * emq * ss) : (lane < 11 ? -q * emq : 0.0);
    const double lp = -(q0 * q0) / 18.0 - 0.5 * emq * ss - 5.0 * q0;
    v += 0.5 * h * g;
    const double ke = rowsum(v * v);
    const double H = -lp + 0.5 * ke;
    if (t == 0) H0 = H;
    if (!(fabs(H - H0) < delta)) { acc += 1.0; }  // the energy test (one pass at c = 0)
    // two U-turn checks against stored states
    const double qa = lane < 16 ? st[0][lane] : 0.0, va = lane < 16 ? st[1][lane] : 0.0;
    const double qb = lane < 16 ? st[2][lane] : 0.0, vb = lane < 16 ? st[3][lane] : 0.0;
    const double d1 = rowsum(v * (q - qa)), d2 = rowsum(va * (q - qa)), d3 = rowsum(v * (q - qb)), d4 = rowsum(vb * (q - qb));
    const bool fire = (d1 < 0.0) || (d2 < 0.0) || (d3 < 0.0) || (d4 < 0.0);
    const double W = exp(H0 - H);
    Wsum += W;
    if (u[(2 * t + 1) & 1023] < W / Wsum) {  // multinomial pick: copy the state
      picks++;
      if (lane < 16) { st[2][lane] = q; st[3][lane] = v; }
    }
    if (fire) { Wsum = 1.0; if (lane < 16) { st[0][lane] = q; st[1][lane] = v; } v = -v; }
  }
  if (lane == 0) { out[blockIdx.x * 2] = acc + Wsum + q + v; out[blockIdx.x * 2 + 1] = picks; }
}
int main() {
  double *out, *u; (void)hipMalloc(&out, 4096 * 16); (void)hipMalloc(&u, 1024 * 8);
  double hu[1024]; for (int i = 0; i < 1024; ++i) hu[i] = (i * 2654435761u % 1000) / 1000.0;
  (void)hipMemcpy(u, hu, sizeof hu, hipMemcpyHostToDevice);
  for (int blocks : {1, 1024, 4096}) {
    const int n = 200000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    tick_floor<<<blocks, 64>>>(out, u, 1000, 0.05, 0.3);
    (void)hipEventRecord(a); tick_floor<<<blocks, 64>>>(out, u, n, 0.05, 0.3); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%d waves: %.3f us per tick\n", blocks, ms * 1e3 / n);
  }
  return 0;
}
