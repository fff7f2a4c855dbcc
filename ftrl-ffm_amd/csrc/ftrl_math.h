// ftrl_math.h -- scalar arithmetic of the FTRL hot path for gfx950, operation for operation in
// the reference's order.  Compiled with -ffp-contract=off and hipcc's default correctly rounded
// fp32 divide/sqrt, so every expression rounds exactly where the reference's x86-64 (-O3, no FMA)
// build rounds; f32 subnormals are kept (hipcc default), as on the host.
#pragma once
#include <hip/hip_runtime.h>

namespace ftrl_dev {

struct Hyper {
  float alpha, beta, l1, l2;
};

// utils::sgn, reference src/include/utils/utils.h:15-18: x > 0 ? 1 : -1 (sgn(0) = -1)
__device__ __forceinline__ float sgn_ref(float x) { return x > 0.0f ? 1.0f : -1.0f; }

// FtrlModel::maybe_zero_weight, src/include/model/ftrl_model.h:28-33.  The reference promotes the
// divide to double and narrows the result; with 24-bit operands that double rounding is
// innocuous, so the fp32 divide below is bit-identical (tests/golden/g1).
__device__ __forceinline__ float ftrl_weight(const Hyper &h, float n, float z) {
  if (fabsf(z) <= h.l1) return 0.0f;
  const float num = z - sgn_ref(z) * h.l1;
  const float den = h.l2 + (h.beta + sqrtf(n)) / h.alpha;
  return (-num) / den;
}

// Linear / bias accumulator step, src/model/ftrl_model.cpp:69-74 and :81-84:
//   s = (sqrtf(n + g*g) - sqrtf(n)) / alpha;  z += g - s*w;  n += g*g
__device__ __forceinline__ void nz_step_linear(const Hyper &h, float w, float g, float &n,
                                               float &z) {
  const float s = (sqrtf(n + g * g) - sqrtf(n)) / h.alpha;
  z = z + (g - s * w);
  n = n + g * g;
}

// Latent accumulator step, FFM i-side src/model/ffm.cpp:112-115 and FM src/model/fm.cpp:90-94:
//   s = (sqrtf(n + g*g) - sqrtf(n)) / alpha;  z' = z + g - s*w;  n' = n + g*g
__device__ __forceinline__ void nz_step_latent(const Hyper &h, float w, float g, float &n,
                                               float &z) {
  const float s = (sqrtf(n + g * g) - sqrtf(n)) / h.alpha;
  z = (z + g) - s * w;
  n = n + g * g;
}

// FFM j-side, src/model/ffm.cpp:117-120 -- INCLUDING the :118 quirk: the square root takes
// n + g2*g1 (product of the two different gradients), which is NaN when that is negative.
__device__ __forceinline__ void nz_step_latent_jside(const Hyper &h, float w, float g2, float g1,
                                                     float &n, float &z) {
  const float s = (sqrtf(n + g2 * g1) - sqrtf(n)) / h.alpha;
  z = (z + g2) - s * w;
  n = n + g2 * g2;
}

// utils::sigmoid<float>, utils.h:20-23: 1 / (1 + std::exp(-x)) with std::exp(float) = expf.
// glibc's expf evaluates in double and rounds once (<= 0.502 ULP); exp() in double rounded to
// float reproduces that result except within ~1e-3 ULP of a rounding boundary.
__device__ __forceinline__ float sigmoid_ref(float x) {
  const float e = static_cast<float>(exp(static_cast<double>(-x)));
  return 1.0f / (1.0f + e);
}

// loss(int y, double logit), src/include/eval/loss.h:8-12 (inf/NaN at saturation preserved)
__device__ __forceinline__ double logloss_ref(int y, float logit) {
  const double s = 1.0 / (1.0 + exp(-static_cast<double>(logit)));
  return static_cast<double>(-y) * log(s) - static_cast<double>(1 - y) * log(1.0 - s);
}

// Counter-based N(0,1): one draw per (seed, stream, index), identical wherever it is evaluated.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float normal01(uint64_t seed, uint64_t stream, uint64_t index) {
  const uint64_t h = mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) + index);
  const uint32_t a = static_cast<uint32_t>(h >> 32), b = static_cast<uint32_t>(h);
  const float u1 = (static_cast<float>(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
  const float u2 = static_cast<float>(b >> 8) * (1.0f / 16777216.0f);           // [0,1)
  return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

}  // namespace ftrl_dev
