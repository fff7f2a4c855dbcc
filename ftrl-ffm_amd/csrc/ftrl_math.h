// ftrl_math.h -- scalar arithmetic of the FTRL hot path for gfx950, operation for operation in
// the reference's order.  Compiled with -ffp-contract=off and hipcc's default correctly rounded
// fp32 divide/sqrt, so every expression rounds exactly where the reference's x86-64 (-O3, no FMA)
// build rounds; f32 subnormals are kept (hipcc default), as on the host.
#pragma once
#include <hip/hip_runtime.h>

namespace ftrl_dev {

struct Hyper {
  float alpha, beta, l1, l2;
  float inv_alpha;  // RN(1/alpha)
  int fast_div;     // 1 when div_alpha's short sequence was proven exact for this alpha
  int learn;        // FFM_FLAG_LEARN: opt-in variant (keep w until n > 0; g2*g2 at ffm.cpp:118)
  int fast_w;       // 1 when fast_div and beta in [2^-40, 2^40]: W(n, z) needs ONE range test, on n
};

// utils::sgn, reference src/include/utils/utils.h:15-18: x > 0 ? 1 : -1 (sgn(0) = -1)
__device__ __forceinline__ float sgn_ref(float x) { return x > 0.0f ? 1.0f : -1.0f; }

// ---- correctly rounded fp32 sqrt and x/alpha in fewer instructions ---------------------------
// These kernels are VALU-bound (DESIGN.md "Rooflines"): hipcc's IEEE sqrtf/divide expand to ~15
// instructions each.  Both replacements below return the SAME bits as sqrtf(x) and x / alpha.

// sqrtf(x) in five instructions, ONE of them transcendental (a quarter-rate instruction): r =
// v_rsq_f32(x) (1 ulp), s = x * r, the exact residual e = x - s*s by FMA, one correction step
// s + e * (0.5 * r) -- Markstein's final iteration, which rounds correctly once s and the reciprocal
// root are this accurate.  Not taken on trust: every float in [2^-96, 2^96] was compared with sqrtf
// on gfx950 (tools/sqrt_probe.hip: 1 610 612 737 inputs, 0 mismatches -- candidate c; rounds 3-4 took
// s from v_sqrt_f32, a second transcendental, equally exact; with 0.5 * v_rcp_f32(s) for the
// correction 96 inputs differ), and ffm_engine_create repeats that comparison on the device it runs
// on and refuses to start if any input differs.  Used when every lane of the wave holds +0 or a
// value in that range; anything else (tiny, huge, negative, -0, NaN) takes sqrtf.
__device__ __forceinline__ bool sqrt_fast_ok(float x) {
  return __float_as_uint(x) == 0u || __builtin_amdgcn_fmed3f(x, 0x1p-96f, 0x1p96f) == x;
}
__device__ __forceinline__ float sqrt_fast(float x) {  // requires x in [2^-96, 2^96]
  const float r = __builtin_amdgcn_rsqf(x);
  const float s = x * r;
  const float h = 0.5f * r;
  const float e = fmaf(-s, s, x);
  return fmaf(e, h, s);
}
// ... and +0: v_rsq_f32(0) is +inf; clamped (inert for x >= 2^-96, where it is at most 2^48), s and
// the correction are 0 * 2^100 = 0.
__device__ __forceinline__ float sqrt_fast0(float x) {  // requires sqrt_fast_ok(x)
  const float r = fminf(__builtin_amdgcn_rsqf(x), 0x1p100f);
  const float s = x * r;
  const float h = 0.5f * r;
  const float e = fmaf(-s, s, x);
  return fmaf(e, h, s);
}
__device__ __forceinline__ float sqrt_cr(float x) {
  if (__all(sqrt_fast_ok(x))) return sqrt_fast0(x);
  return sqrtf(x);
}

// x / alpha for the engine's constant alpha: q0 = x*r with r = RN(1/alpha), exact residual by
// FMA, one correction (Markstein).  ffm_engine_create checks this sequence against the IEEE
// divide for every one of the 2^24 float significands (the rounding of a quotient depends only
// on significands while nothing under/overflows) and clears fast_div if any differs; at run time
// it is used when every lane's dividend is +0 or comfortably normal.
__device__ __forceinline__ bool div_fast_ok(float x) {
  const float ax = fabsf(x);
  return (ax >= 0x1p-60f && ax <= 0x1p60f) || __float_as_uint(x) == 0u;
}
__device__ __forceinline__ float div_alpha_fast(const Hyper &h, float x) {
  const float q0 = x * h.inv_alpha;
  const float rem = fmaf(-q0, h.alpha, x);
  return fmaf(rem, h.inv_alpha, q0);
}
__device__ __forceinline__ float div_alpha(const Hyper &h, float x) {
  if (h.fast_div && __all(div_fast_ok(x))) return div_alpha_fast(h, x);
  return x / h.alpha;
}

// One test for a sqrt operand of the accumulator chains: x in [2^-70, 2^96] (false for 0, negative,
// NaN, inf).  Inside it sqrt_fast(x) is exact, and the difference of two such square roots is +0
// or has magnitude in [2^-58, 2^48] -- inside div_alpha_fast's proven range.
__device__ __forceinline__ bool chain_operand_ok(float x) {
  return __builtin_amdgcn_fmed3f(x, 0x1p-70f, 0x1p96f) == x;
}

// N values at once with ONE wave vote (a vote is a scheduling barrier: hoisting it lets the
// compiler interleave the N independent sequences).
template <int N>
__device__ __forceinline__ void sqrt_cr_n(const float (&x)[N], float (&r)[N]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; i++) ok = ok && sqrt_fast_ok(x[i]);
  if (__all(ok)) {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = sqrt_fast0(x[i]);
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = sqrtf(x[i]);
  }
}
template <int N>
__device__ __forceinline__ void div_alpha_n(const Hyper &h, const float (&x)[N], float (&r)[N]) {
  bool ok = h.fast_div != 0;
#pragma unroll
  for (int i = 0; i < N; i++) ok = ok && div_fast_ok(x[i]);
  if (__all(ok)) {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = div_alpha_fast(h, x[i]);
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = x[i] / h.alpha;
  }
}

// FtrlModel::maybe_zero_weight, src/include/model/ftrl_model.h:28-33.  The reference promotes the
// divide to double and narrows the result; with 24-bit operands that double rounding is
// innocuous, so the fp32 divide below is bit-identical (tests/golden/g1).
__device__ __forceinline__ float ftrl_weight(const Hyper &h, float n, float z) {
  if (fabsf(z) <= h.l1) return 0.0f;
  const float num = z - sgn_ref(z) * h.l1;
  const float den = h.l2 + div_alpha(h, h.beta + sqrt_cr(n));
  return (-num) / den;
}

// N weights at once (float4s of slots): same arithmetic as ftrl_weight per component, the square
// roots and alpha divides each behind one wave vote and interleaved across the N values.
template <int N>
__device__ __forceinline__ void ftrl_weight_n(const Hyper &h, const float (&nn)[N],
                                              const float (&zz)[N], float (&w)[N]) {
  float dv[N];
  // fast_w: beta is comfortably normal, so beta + sqrt(n) lies inside div_alpha_fast's proven range
  // whenever n lies inside sqrt_fast's -- one range test per value, one vote for all
  bool ok = h.fast_w != 0;
#pragma unroll
  for (int i = 0; i < N; i++) ok = ok && sqrt_fast_ok(nn[i]);
  if (__all(ok)) {
#pragma unroll
    for (int i = 0; i < N; i++) dv[i] = div_alpha_fast(h, h.beta + sqrt_fast0(nn[i]));
  } else {
    float sq[N], t[N];
    sqrt_cr_n<N>(nn, sq);
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = h.beta + sq[i];
    div_alpha_n<N>(h, t, dv);
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    // z - sgn(z)*l1: sgn*l1 is +-l1 exactly, and a - b == a + (-b)
    const float num = zz[i] + (zz[i] > 0.0f ? -h.l1 : h.l1);
    const float den = h.l2 + dv[i];
    w[i] = fabsf(zz[i]) <= h.l1 ? 0.0f : (-num) / den;
  }
}
// Latent refresh rule: W(n, z), except that the learning variant keeps the stored weight of a slot
// that has not seen a gradient yet (n == 0).
__device__ __forceinline__ float latent_weight(const Hyper &h, float n, float z, float w_old) {
  const float w = ftrl_weight(h, n, z);
  return (h.learn && !(n > 0.0f)) ? w_old : w;
}
__device__ __forceinline__ float4 ftrl_weight4(const Hyper &h, float4 n, float4 z) {
  const float nn[4] = {n.x, n.y, n.z, n.w}, zz[4] = {z.x, z.y, z.z, z.w};
  float w[4];
  ftrl_weight_n<4>(h, nn, zz, w);
  return make_float4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ float4 latent_weight4(const Hyper &h, float4 n, float4 z, float4 w_old) {
  float4 w = ftrl_weight4(h, n, z);
  if (h.learn) {
    w.x = n.x > 0.0f ? w.x : w_old.x;
    w.y = n.y > 0.0f ? w.y : w_old.y;
    w.z = n.z > 0.0f ? w.z : w_old.z;
    w.w = n.w > 0.0f ? w.w : w_old.w;
  }
  return w;
}

// Linear / bias accumulator step, src/model/ftrl_model.cpp:69-74 and :81-84:
//   s = (sqrtf(n + g*g) - sqrtf(n)) / alpha;  z += g - s*w;  n += g*g
__device__ __forceinline__ void nz_step_linear(const Hyper &h, float w, float g, float &n,
                                               float &z) {
  const float s = div_alpha(h, sqrt_cr(n + g * g) - sqrt_cr(n));
  z = z + (g - s * w);
  n = n + g * g;
}

// Latent accumulator step, FFM i-side src/model/ffm.cpp:112-115 and FM src/model/fm.cpp:90-94:
//   s = (sqrtf(n + g*g) - sqrtf(n)) / alpha;  z' = z + g - s*w;  n' = n + g*g
__device__ __forceinline__ void nz_step_latent(const Hyper &h, float w, float g, float &n,
                                               float &z) {
  const float s = div_alpha(h, sqrt_cr(n + g * g) - sqrt_cr(n));
  z = (z + g) - s * w;
  n = n + g * g;
}

// The same two steps for a chain of touches of one accumulator: sqrt(n + g*g) of a touch is
// sqrt(n) of the next, so the caller carries it (sqn = sqrt_cr(n) on entry; both square roots are
// correctly rounded, so the bits are those of the two-root form).
__device__ __forceinline__ void nz_step_latent_carry(const Hyper &h, float w, float g, float &n,
                                                     float &z, float &sqn) {
  const float na = n + g * g;
  const float sa = sqrt_cr(na);
  const float s = div_alpha(h, sa - sqn);
  z = (z + g) - s * w;
  n = na;
  sqn = sa;
}
__device__ __forceinline__ void nz_step_linear_carry(const Hyper &h, float w, float g, float &n,
                                                     float &z, float &sqn) {
  const float na = n + g * g;
  const float sa = sqrt_cr(na);
  const float s = div_alpha(h, sa - sqn);
  z = z + (g - s * w);
  n = na;
  sqn = sa;
}

// FFM j-side, src/model/ffm.cpp:117-120 -- INCLUDING the :118 quirk: the square root takes
// n + g2*g1 (product of the two different gradients), which is NaN when that is negative.
__device__ __forceinline__ void nz_step_latent_jside(const Hyper &h, float w, float g2, float g1,
                                                     float &n, float &z) {
  const float s = div_alpha(h, sqrt_cr(n + (h.learn ? g2 * g2 : g2 * g1)) - sqrt_cr(n));
  z = (z + g2) - s * w;
  n = n + g2 * g2;
}

// std::exp(float) as the reference's C library evaluates it.  glibc >= 2.27 expf (the algorithm of
// ARM optimized-routines: x*N/ln2 = k + r, 2^(k/N) from a 32-entry table, cubic in r, all in
// double, one rounding to float) is NOT correctly rounded in ~8e-5 of inputs, so a "more accurate"
// device exp would differ from the reference in those.  This is that algorithm, with the fused
// multiply-adds x86-64 glibc's FMA build performs; checked on the device against the host C
// library's 1/(1+expf(-x)) on 300 k inputs including the tails
// (tests/test_gpu_parity.py::test_device_sigmoid_is_the_c_library_sigmoid).
// 2^(i/32) for i = 0..31, bits of the double with the exponent's low bits folded in (glibc's
// __exp2f_data.tab).  Callers on a latency-critical path stage it in LDS and pass that copy.
static __device__ const uint64_t kExpTab[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

template <typename TabPtr>
__device__ __forceinline__ float expf_glibc_tab(float x, TabPtr T) {
  const uint32_t ix = __float_as_uint(x);
  const uint32_t abstop = (ix >> 20) & 0x7ff;
  if (abstop >= 0x42b) {  // |x| >= 88 or NaN
    if (ix == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);  // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;                         // underflow
  }
  const double InvLn2N = 0x1.71547652b82fep+0 * 32;
  const double Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32,
               C2 = 0x1.62e42ff0c52d6p-1 / 32;
  const double xd = static_cast<double>(x);
  const double z = InvLn2N * xd;
  double kd = z + Shift;
  const uint64_t ki = static_cast<uint64_t>(__double_as_longlong(kd));
  kd -= Shift;
  const double r = fma(InvLn2N, xd, -kd);
  const uint64_t t = T[ki & 31] + (ki << 47);
  const double s = __longlong_as_double(static_cast<long long>(t));
  const double p = fma(C0, r, C1);
  const double r2 = r * r;
  double y = fma(C2, r, 1.0);
  y = fma(p, r2, y);
  y = y * s;
  return static_cast<float>(y);
}

__device__ __forceinline__ float expf_glibc(float x) { return expf_glibc_tab(x, kExpTab); }

// utils::sigmoid<float>, utils.h:20-23: 1 / (1 + std::exp(-x)) with std::exp(float) = expf.
__device__ __forceinline__ float sigmoid_ref(float x) {
  return 1.0f / (1.0f + expf_glibc(-x));
}
template <typename TabPtr>
__device__ __forceinline__ float sigmoid_ref_tab(float x, TabPtr T) {
  return 1.0f / (1.0f + expf_glibc_tab(-x, T));
}

// loss(int y, double logit), src/include/eval/loss.h:8-12 (inf/NaN at saturation preserved)
__device__ __forceinline__ double logloss_ref(int y, float logit) {
  const double s = 1.0 / (1.0 + exp(-static_cast<double>(logit)));
  return static_cast<double>(-y) * log(s) - static_cast<double>(1 - y) * log(1.0 - s);
}

}  // namespace ftrl_dev
