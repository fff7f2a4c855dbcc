// dpp_probe.hip -- does a strictly sequential fp32 running sum over the 16 lanes of a DPP row,
// built from in-place `v_add_f32_dpp row_shr:1` steps, give the bits of a one-lane loop, and what
// does one step cost?  (Design probe for the very-hot-feature chain kernel; not part of the product.)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dpp_probe.hip -o tools/dpp_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// lanes 1..15 of each row: S += left neighbour's S ... expressed as S = S_left + a with the add
// fused into the DPP move; lane 0 of a row has no source (bound_ctrl off): it keeps its S.
__device__ __forceinline__ float row_chain15(float s, float a) {
#pragma unroll
  for (int r = 1; r < 16; r++)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(s)
                 : "v"(a));
  return s;
}

// z chain: Z = (Z_left + g) - m
__device__ __forceinline__ float row_chain15_z(float z, float g, float m) {
#pragma unroll
  for (int r = 1; r < 16; r++)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_sub_f32 %0, %0, %2"
                 : "+v"(z)
                 : "v"(g), "v"(m));
  return z;
}

// one wave: 4 rows = 4 independent chains of `steps` x 16 touches
__global__ void probe(const float *a, const float *g, const float *m, float *out_n, float *out_z,
                      int steps, long long *cycles) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & 15, row = lane >> 4;
  float nc = 0.25f + row, zc = -0.5f * row;
  const long long t0 = clock64();
  for (int st = 0; st < steps; st++) {
    const int t = st * 16 + tl;
    const float av = a[(row * steps * 16) + t], gv = g[(row * steps * 16) + t], mv = m[(row * steps * 16) + t];
    // lane 0 applies its touch to the carried value; lanes 1..15 chain from the left
    float S = tl == 0 ? nc + av : 0.0f;
    S = row_chain15(S, av);
    float Z = tl == 0 ? (zc + gv) - mv : 0.0f;
    Z = row_chain15_z(Z, gv, tl == 0 ? 0.0f : mv);  // lane 0 keeps its value: the sub must be - (+0)
    // carry = lane 15 of the row
    nc = __shfl(S, (lane & ~15) | 15, 64);
    zc = __shfl(Z, (lane & ~15) | 15, 64);
    out_n[(row * steps * 16) + t] = S;
    out_z[(row * steps * 16) + t] = Z;
  }
  if (lane == 0) *cycles = clock64() - t0;
}

int main() {
  const int steps = 512, n = 4 * steps * 16;
  std::vector<float> a(n), g(n), m(n), rn(n), rz(n), on(n), oz(n);
  srand(1);
  for (int i = 0; i < n; i++) {
    a[i] = (rand() % 1000) * 1e-4f + 1e-7f * (rand() % 100);
    g[i] = ((rand() % 2000) - 1000) * 1e-3f;
    m[i] = ((rand() % 2000) - 1000) * 3e-4f;
  }
  for (int row = 0; row < 4; row++) {
    float nc = 0.25f + row, zc = -0.5f * row;
    for (int t = 0; t < steps * 16; t++) {
      const int i = row * steps * 16 + t;
      nc = nc + a[i];
      zc = (zc + g[i]) - m[i];
      rn[i] = nc;
      rz[i] = zc;
    }
  }
  float *da, *dg, *dm, *dn, *dz;
  long long *dc, cyc = 0;
  hipMalloc(&da, n * 4); hipMalloc(&dg, n * 4); hipMalloc(&dm, n * 4);
  hipMalloc(&dn, n * 4); hipMalloc(&dz, n * 4); hipMalloc(&dc, 8);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dg, g.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dm, m.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dg, dm, dn, dz, steps, dc);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dg, dm, dn, dz, steps, dc);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("one wave, %d steps of 16 touches: %.1f us -> %.1f ns per step, %.2f ns per touch\n", steps, ms * 1e3,
         ms * 1e6 / steps, ms * 1e6 / steps / 16);
  hipMemcpy(on.data(), dn, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(oz.data(), dz, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  int bad_n = 0, bad_z = 0;
  for (int i = 0; i < n; i++) {
    bad_n += memcmp(&on[i], &rn[i], 4) != 0;
    bad_z += memcmp(&oz[i], &rz[i], 4) != 0;
  }
  printf("dpp row_shr chain: n mismatches %d / %d, z mismatches %d / %d; %.1f clock64 ticks per 16-touch step "
         "(n chain + z chain + loads), %.2f per touch\n", bad_n, n, bad_z, n, (double)cyc / steps, (double)cyc / steps / 16);
  return (bad_n || bad_z) ? 1 : 0;
}
