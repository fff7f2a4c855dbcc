// Exhaustive check of a shorter correctly rounded sqrt candidate against sqrtf on gfx950:
//   s = v_sqrt_f32(x); h = 0.5f * v_rsq_f32(x); e = fma(-s, s, x); r = fma(e, h, s)
// over every float in [2^-96, 2^96] (and 0).  Prints the number of mismatches and the first few.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/sqrt_probe.hip -o /tmp/sqrt_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float cand_a(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float e = fmaf(-s, s, x);
  return fmaf(e, h, s);
}
__device__ __forceinline__ float cand_b(float x) {  // reciprocal of s instead of rsq(x)
  const float s = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rcpf(s);
  const float e = fmaf(-s, s, x);
  return fmaf(e, h, s);
}
__device__ __forceinline__ float cand_c(float x) {  // ONE transcendental: s = x * rsq(x)
  const float r = __builtin_amdgcn_rsqf(x);
  const float s = x * r;
  const float h = 0.5f * r;
  const float e = fmaf(-s, s, x);
  return fmaf(e, h, s);
}
__device__ __forceinline__ float cand_d(float x) {  // ... with a second correction
  const float r = __builtin_amdgcn_rsqf(x);
  const float s = x * r;
  const float h = 0.5f * r;
  const float e = fmaf(-s, s, x);
  const float s1 = fmaf(e, h, s);
  const float e1 = fmaf(-s1, s1, x);
  return fmaf(e1, h, s1);
}
__global__ void probe(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t b = lo + blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; b <= hi; b += stride) {
    const float x = __uint_as_float(static_cast<uint32_t>(b));
    const float want = sqrtf(x);
    const float a = cand_a(x), c = cand_b(x);
    if (__float_as_uint(a) != __float_as_uint(want)) {
      const unsigned long long k = atomicAdd(&bad[0], 1ull);
      if (k < 8) first[k] = static_cast<uint32_t>(b);
    }
    if (__float_as_uint(c) != __float_as_uint(want)) {
      const unsigned long long k = atomicAdd(&bad[1], 1ull);
      if (k < 8) first[8 + k] = static_cast<uint32_t>(b);
    }
    if (__float_as_uint(cand_c(x)) != __float_as_uint(want)) atomicAdd(&bad[2], 1ull);
    if (__float_as_uint(cand_d(x)) != __float_as_uint(want)) atomicAdd(&bad[3], 1ull);
  }
}
int main() {
  unsigned long long *bad; uint32_t *first;
  hipMalloc(&bad, 32); hipMalloc(&first, 64); hipMemset(bad, 0, 32); hipMemset(first, 0, 64);
  const float flo = 0x1p-96f, fhi = 0x1p96f;
  uint32_t lo, hi; memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
  hipLaunchKernelGGL(probe, dim3(8192), dim3(256), 0, 0, lo, hi, bad, first);
  unsigned long long hb[4]; uint32_t hf[16];
  hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 64, hipMemcpyDeviceToHost);
  printf("inputs %llu\n", (unsigned long long)hi - lo + 1);
  printf("candidate a (0.5*rsq(x)): %llu mismatches", hb[0]);
  for (int i = 0; i < 8 && i < (int)hb[0]; i++) printf(" %08x", hf[i]);
  printf("\ncandidate b (0.5*rcp(s)): %llu mismatches", hb[1]);
  for (int i = 0; i < 8 && i < (int)hb[1]; i++) printf(" %08x", hf[8 + i]);
  printf("\ncandidate c (s = x*rsq(x), one correction): %llu mismatches", hb[2]);
  printf("\ncandidate d (s = x*rsq(x), two corrections): %llu mismatches", hb[3]);
  printf("\n");
  return 0;
}
