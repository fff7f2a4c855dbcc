// kernels_touch.h -- one FTRL (n, z) touch of an FFM slot, shared by the feature-major update
// kernels (kernels_update.h) and the row kernel's in-row update of once-only features
// (kernels_row.h).  FFM::update_vector_nz, src/model/ffm.cpp:104-120 incl. the :118 quirk.
#pragma once
#include "engine_types.h"

namespace ftrl_dev {

// One touch of slot (own feature, partner field fp) by the pair {own entry, other entry}.
__device__ __forceinline__ void ffm_touch(const Hyper &h, bool own_first, float tg, float x_own,
                                          float x_other, float vp, float w, float &n, float &z) {
  if (own_first) {
    // own entry is the pair's first: slot (i, field2), ffm.cpp:112-115
    const float x = x_own * x_other;
    const float g1 = tg * vp * x;
    nz_step_latent(h, w, g1, n, z);
  } else {
    // own entry is the pair's second: slot (j, field1), ffm.cpp:117-120 with the :118 quirk
    const float x = x_other * x_own;
    const float g2 = tg * vp * x;  // tmp_grad * vif1 * x
    const float g1 = tg * w * x;   // tmp_grad * vif2 * x (the first entry's gradient)
    nz_step_latent_jside(h, w, g2, g1, n, z);
  }
}

// The same touch on N factors of one slot held by one lane, votes hoisted (sqrt_cr_n /
// div_alpha_n): one wave vote per stage instead of one per factor.
template <int N>
__device__ __forceinline__ void ffm_touch_n(const Hyper &h, bool own_first, float tg, float x_own,
                                            float x_other, const float (&vp)[N],
                                            const float (&w)[N], float (&n)[N], float (&z)[N]) {
  float g[N], arg[2 * N], sq[2 * N], d[N], sg[N];
  const float x = own_first ? x_own * x_other : x_other * x_own;
#pragma unroll
  for (int i = 0; i < N; i++) {
    g[i] = tg * vp[i] * x;              // own slot's gradient
    const float g1 = tg * w[i] * x;     // second-entry case: the first entry's gradient
    arg[i] = n[i] + ((own_first || h.learn) ? g[i] * g[i] : g[i] * g1);  // ffm.cpp:113 / :118
    arg[N + i] = n[i];
  }
  // both square roots and the alpha divide in their short exact forms when every operand of the
  // wave is comfortably normal (one vote; ftrl_math.h: chain_operand_ok), else IEEE
  bool ok = h.fast_div != 0;
#pragma unroll
  for (int i = 0; i < 2 * N; i++) ok = ok && chain_operand_ok(arg[i]);
  if (__all(ok)) {
#pragma unroll
    for (int i = 0; i < 2 * N; i++) sq[i] = sqrt_fast(arg[i]);
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = sq[i] - sq[N + i];
#pragma unroll
    for (int i = 0; i < N; i++) sg[i] = div_alpha_fast(h, d[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 2 * N; i++) sq[i] = sqrtf(arg[i]);
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = sq[i] - sq[N + i];
#pragma unroll
    for (int i = 0; i < N; i++) sg[i] = d[i] / h.alpha;
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    z[i] = (z[i] + g[i]) - sg[i] * w[i];
    n[i] = n[i] + g[i] * g[i];
  }
}

__device__ __forceinline__ void ffm_touch4(const Hyper &h, bool own_first, float tg, float x_own,
                                           float x_other, float4 vp4, float4 w4, float4 &n4,
                                           float4 &z4) {
  const float vp[4] = {vp4.x, vp4.y, vp4.z, vp4.w}, w[4] = {w4.x, w4.y, w4.z, w4.w};
  float n[4] = {n4.x, n4.y, n4.z, n4.w}, z[4] = {z4.x, z4.y, z4.z, z4.w};
  ffm_touch_n<4>(h, own_first, tg, x_own, x_other, vp, w, n, z);
  n4 = make_float4(n[0], n[1], n[2], n[3]);
  z4 = make_float4(z[0], z[1], z[2], z[3]);
}

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Strictly left-to-right running sum over the 64 lanes: returns S_j = ((carry + a_0) + a_1) ... + a_j
// in lane j, every addition rounded exactly as a one-lane sequential loop would round it.  Step t
// finalises lane t (S_t = S_{t-1} + a_t through a whole-wave shift-right-by-one DPP move); lanes
// already final recompute the same value, so no predication is needed.  Idle lanes pass -0.0f
// (x + -0.0f == x bit for bit, for every x including both zeros).
// Each of the 63 steps is ONE in-place `v_add_f32_dpp s, s, a wave_shr:1`: lane t adds its a to
// its left neighbour's running value; lane 0 has no source lane, so the hardware leaves it alone
// (bound_ctrl off) and it keeps carry + a_0.  (As a DPP move followed by an add this chain was two
// dependent instructions per touch: the bias chain of a 65536-row block took 1.65 ms that way.)
__device__ __forceinline__ float wave_sequential_prefix(float carry, float a) {
  const int lane = threadIdx.x & 63;
  float s = lane == 0 ? carry + a : a;
#define FTRL_WSHR "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define FTRL_REP7(x) x x x x x x x
  asm volatile(FTRL_REP7(FTRL_REP7(FTRL_WSHR)) FTRL_REP7(FTRL_WSHR) FTRL_REP7(FTRL_WSHR)
               : "+v"(s)
               : "v"(a));
#undef FTRL_REP7
#undef FTRL_WSHR
  return s;
}

}  // namespace ftrl_dev
