// init_rng.h -- the engine's seeded N(mean, stddev) weight initialiser, one draw per (seed, stream,
// index), IDENTICAL BITS on the device and on the host.
//
// Replaces utils::gaussian / utils::init_weights (reference src/include/utils/utils.h:30-61: a
// fresh std::random_device + std::mt19937 + std::normal_distribution per weight, unseeded and
// ~32 us per weight).  Here a draw is a pure function of its coordinates: a 64-bit integer hash
// gives two uniforms, Box-Muller turns them into a normal.  The transcendental parts (ln, cos) are
// spelled out in double precision with +, -, *, one /, one sqrt and explicit operation order only
// -- every one of those is correctly rounded IEEE-754 on gfx950 and on x86-64 alike, and both
// builds forbid contraction (-ffp-contract=off) -- so the host function below returns exactly what
// init_weights_kernel stores.  ffm_engine_init_weights_host exposes it; tests compare bit for bit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FTRL_HD __host__ __device__ __forceinline__
#else
#define FTRL_HD inline
#endif

namespace ftrl_rng {

FTRL_HD uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

FTRL_HD uint64_t draw_bits(uint64_t seed, uint64_t stream, uint64_t index) {
  return mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) + index);
}

FTRL_HD double bits_to_double(uint64_t b) {
  union { uint64_t u; double d; } v;
  v.u = b;
  return v.d;
}
FTRL_HD uint64_t double_to_bits(double d) {
  union { uint64_t u; double d; } v;
  v.d = d;
  return v.u;
}

// ln(u) for a normal double u in (0, 1]: u = m * 2^e with m in [1/sqrt2, sqrt2),
// ln m = 2 atanh(t), t = (m-1)/(m+1), |t| <= 0.1716: odd series to t^17 (error < 1e-14).
FTRL_HD double ln_unit(double u) {
  const uint64_t b = double_to_bits(u);
  int e = static_cast<int>((b >> 52) & 0x7ff) - 1023;
  double m = bits_to_double((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);  // [1, 2)
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  const double t = (m - 1.0) / (m + 1.0);
  const double t2 = t * t;
  double p = 1.0 / 17.0;
  p = p * t2 + 1.0 / 15.0;
  p = p * t2 + 1.0 / 13.0;
  p = p * t2 + 1.0 / 11.0;
  p = p * t2 + 1.0 / 9.0;
  p = p * t2 + 1.0 / 7.0;
  p = p * t2 + 1.0 / 5.0;
  p = p * t2 + 1.0 / 3.0;
  p = p * t2 + 1.0;
  return static_cast<double>(e) * 0.6931471805599453 + (2.0 * t) * p;
}

// cos(2 pi u) for u in [0, 1): quadrant from 4u, then the Taylor series of sin / cos on [0, pi/2].
FTRL_HD double cos_2pi(double u) {
  const double x = u * 4.0;
  const int q = static_cast<int>(x);          // 0..3
  const double a = (x - static_cast<double>(q)) * 1.5707963267948966;  // [0, pi/2)
  const double a2 = a * a;
  // cos a = sum (-1)^n a^(2n) / (2n)!, to a^24
  double c = 1.0 / 620448401733239439360000.0;
  c = -1.0 / 1124000727777607680000.0 + c * a2;
  c = 1.0 / 2432902008176640000.0 + c * a2;
  c = -1.0 / 6402373705728000.0 + c * a2;
  c = 1.0 / 20922789888000.0 + c * a2;
  c = -1.0 / 87178291200.0 + c * a2;
  c = 1.0 / 479001600.0 + c * a2;
  c = -1.0 / 3628800.0 + c * a2;
  c = 1.0 / 40320.0 + c * a2;
  c = -1.0 / 720.0 + c * a2;
  c = 1.0 / 24.0 + c * a2;
  c = -0.5 + c * a2;
  c = 1.0 + c * a2;
  // sin a = a * sum (-1)^n a^(2n) / (2n+1)!, to a^25
  double s = 1.0 / 15511210043330985984000000.0;
  s = -1.0 / 25852016738884976640000.0 + s * a2;
  s = 1.0 / 51090942171709440000.0 + s * a2;
  s = -1.0 / 121645100408832000.0 + s * a2;
  s = 1.0 / 355687428096000.0 + s * a2;
  s = -1.0 / 1307674368000.0 + s * a2;
  s = 1.0 / 6227020800.0 + s * a2;
  s = -1.0 / 39916800.0 + s * a2;
  s = 1.0 / 362880.0 + s * a2;
  s = -1.0 / 5040.0 + s * a2;
  s = 1.0 / 120.0 + s * a2;
  s = -1.0 / 6.0 + s * a2;
  s = (1.0 + s * a2) * a;
  return q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

#if defined(__HIPCC__)
FTRL_HD double sqrt_ieee(double x) { return __builtin_sqrt(x); }  // v_sqrt_f64 + fixup: correctly rounded
#else
FTRL_HD double sqrt_ieee(double x) { return __builtin_sqrt(x); }
#endif

// One N(0,1) draw: u1 in (0,1), u2 in [0,1) from the two halves of the hash; |z| <= 6.66.
FTRL_HD float normal01(uint64_t seed, uint64_t stream, uint64_t index) {
  const uint64_t h = draw_bits(seed, stream, index);
  const double u1 = (static_cast<double>(static_cast<uint32_t>(h >> 32)) + 0.5) * (1.0 / 4294967296.0);
  const double u2 = static_cast<double>(static_cast<uint32_t>(h)) * (1.0 / 4294967296.0);
  const double r = sqrt_ieee(-2.0 * ln_unit(u1));
  return static_cast<float>(r * cos_2pi(u2));
}

// uniform in [0, 1) with 24 bits
FTRL_HD float uniform01(uint64_t seed, uint64_t stream, uint64_t index) {
  return static_cast<float>(draw_bits(seed, stream, index) >> 40) * (1.0f / 16777216.0f);
}

// The weight the engine gives to linear feature i (stream 0) / latent element idx (stream 1).
FTRL_HD float init_weight(uint64_t seed, int latent, uint64_t index, float mean, float stddev) {
  return mean + stddev * normal01(seed, latent ? 1 : 0, index);
}

}  // namespace ftrl_rng
