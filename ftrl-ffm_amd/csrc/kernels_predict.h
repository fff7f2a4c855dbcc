// kernels_predict.h -- FFM evaluation rows, one WAVE per row (src/eval/evaluate.cpp:23-33 over
// FFM::predict, src/model/ffm.cpp:24-70 with the stored w: no refresh, no update).
//
// A predict row has nothing to hide its serial sections behind when a whole workgroup waits on
// them (ffm_row_kernel<false, ..>: entries staged by one wave, the linear logit by one lane, the
// 741 terms of a 39-field row added in the reference's order by one wave, three barriers): six to
// eight rows per CU in flight, each ~20 us long.  Here a row is ONE wave and needs no barrier, so
// all 8192 rows of a block are resident at once (eight waves per SIMD) and a row's latency hides
// behind 31 others on its CU.
//
// Lanes: LPP consecutive lanes share a pair, each loads VPL 16-byte vectors of both slots -- with
// LPP = k / 4 one load instruction fetches whole 64-byte slots (a lane per pair reads 16 of the 64
// bytes of a line per instruction).  The k-long dot runs lane after lane in factor order (DPP
// row_shr:1 hands the running value on), so it is the reference's sequential sum; the term is
// (dot*x1)*x2; the terms go through the wave's LDS into the strictly ordered sum
// (wave_sequential_prefix: one DPP add per term).  U steps of loads are in flight together.
// Sharded / compact engines, LR, and k outside {4, 8, 16, 32, 64} keep ffm_row_kernel.
// A wave's LDS holds kPredLdsCap entries at most: when the caller's rows may be longer (a device-
// resident block is not scanned on the host: its bound is the engine's max_row_nnz), rows beyond
// that are left to ffm_row_kernel in a second launch, which returns at once for the others.
#pragma once
#include "engine_types.h"
#include "kernels_touch.h"
#include "kernels_update.h"

namespace ftrl_dev {

constexpr int kPredRows = 4;     // rows (= waves) per workgroup
constexpr int kPredTerms = 256;  // terms staged per wave between two ordered sums (a multiple of 64)
constexpr int kPredLdsCap = 128; // entries of a row a wave stages at most (12 KB per workgroup: eight waves per SIMD)

__host__ __device__ inline size_t pred_lds_bytes(int lds_cap) {
  return static_cast<size_t>(kPredRows) * (static_cast<size_t>(lds_cap) * 16 + kPredTerms * 4);
}

// the value of the lane below (row_shr:1 inside the 16-lane DPP row; the first lane of a pair's
// group never uses it)
__device__ __forceinline__ float pred_lane_below(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
}

// Six waves per SIMD, not the seven its registers would allow: with the rows streaming from host
// memory the upload kernel of the next block needs room beside this one (its 24 workgroups otherwise
// wait for a whole round of rows to retire): 51.9 -> 54.3 M rows/s with the H2D, resident unchanged.
#ifndef FFM_PRED_WAVES
#define FFM_PRED_WAVES 6
#endif
#define FFM_PRED_OCC __attribute__((amdgpu_waves_per_eu(FFM_PRED_WAVES, FFM_PRED_WAVES)))
template <int LPP, int VPL, int U>
__global__ __launch_bounds__(64 * kPredRows) FFM_PRED_OCC void ffm_predict_wave_kernel(ModelDev m, Rows rows, Scratch s,
                                                                         int max_row_nnz, int lds_cap, float *out,
                                                                         int output_prob) {
  static_assert(LPP >= 1 && LPP <= 16 && (LPP & (LPP - 1)) == 0, "a pair's lanes sit inside one DPP row");
  static_assert(kPredTerms % 64 == 0, "whole prefix chunks");
  constexpr int PPS = 64 / LPP;  // pairs per step
  static_assert(kPredTerms % (PPS * U) == 0, "a batch of terms is whole unrolled steps");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = wave_uniform(blockIdx.x * kPredRows + wv);
  if (r >= rows.n_rows) return;
  int4 *E = reinterpret_cast<int4 *>(smem + static_cast<size_t>(wv) * (static_cast<size_t>(lds_cap) * 16 + kPredTerms * 4));
  float *terms = reinterpret_cast<float *>(E + lds_cap);
  const int b = wave_uniform(rows.row_ptr[r]);
  const int nnz = wave_uniform(rows.row_ptr[r + 1]) - b;
  if (nnz > max_row_nnz) {  // beyond the validated capacity: flagged, NaN outputs
    if (lane == 0) {
      atomicOr(s.err, ERR_ROW_TOO_LONG);
      const float nan = __int_as_float(0x7fc00000);
      s.loss[r] = static_cast<double>(nan);
      if (out) out[r] = nan;
    }
    return;
  }
  if (nnz > lds_cap) return;  // (a longer row: ffm_row_kernel's, in the launch behind this one)
  const int k = m.n_factors, RL = m.row_len;

  // ---- entries: remove_out_range (ftrl_model.cpp:36-42, ffm.cpp:30-36), the survivors in row
  // order into LDS as {feature, field, value}; the linear logit bias + sum lin_w * x in row order
  // (compute_linear_logit, ftrl_model.cpp:44-50) as a strictly sequential prefix
  float result = m.bias3[0];
  int nv = 0;
  for (int base = 0; base < nnz; base += 64) {
    const int p = base + lane;
    int i = 0, f = 0;
    float x = 0.0f, lw = 0.0f;
    bool valid = false;
    if (p < nnz) {
      i = rows.feat[b + p];
      f = rows.field[b + p];
      x = rows.val[b + p];
      valid = i >= 0 && i < m.n_feats && f >= 0 && f < m.n_fields;
    }
    if (valid) lw = m.lin_w[i];
    const unsigned long long mask = __ballot(valid);
    if (valid) E[nv + __popcll(mask & ((1ull << lane) - 1ull))] = make_int4(i, f, __float_as_int(x), 0);
    nv += __popcll(mask);
    // x + -0.0f == x bit for bit: idle lanes and erased entries add nothing
    const float run = wave_sequential_prefix(result, valid ? lw * x : -0.0f);
    result = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(run), 63));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

  // ---- pairs in the reference's order (a outer, b inner: ffm.cpp:52-66)
  const int n_pairs = nv * (nv - 1) / 2;
  const int g = lane / LPP, c = lane - g * LPP;
  // this lane's pair of the coming step: number q = step * PPS + g
  int pa = 0, pb = 1 + g;
  if (g < n_pairs)
    while (pb >= nv) { pb = pb - nv + pa + 2; pa++; }
  for (int q0 = 0; q0 < n_pairs; q0 += kPredTerms) {
    const int cnt = min(kPredTerms, n_pairs - q0);
    for (int st = 0; st * PPS < cnt; st += U) {
      float4 x[U][VPL], y[U][VPL];
      float xa[U], xb[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int q = q0 + (st + u) * PPS + g;
        ok[u] = q < n_pairs;
        const int a = ok[u] ? pa : 0, bb = ok[u] ? pb : (nv > 1 ? 1 : 0);
        const int4 ea = E[a], eb = E[bb];
        xa[u] = __int_as_float(ea.z);
        xb[u] = __int_as_float(eb.z);
        const float4 *va = reinterpret_cast<const float4 *>(m.lat + static_cast<int64_t>(ea.x) * 3 * RL + LAT_W * RL + eb.y * k) + c * VPL;
        const float4 *vb = reinterpret_cast<const float4 *>(m.lat + static_cast<int64_t>(eb.x) * 3 * RL + LAT_W * RL + ea.y * k) + c * VPL;
#pragma unroll
        for (int v = 0; v < VPL; v++) { x[u][v] = va[v]; y[u][v] = vb[v]; }
        if (q + PPS < n_pairs) {  // the pair PPS further on
          pb += PPS;
          while (pb >= nv) { pb = pb - nv + pa + 2; pa++; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        float pr[VPL][4];
#pragma unroll
        for (int v = 0; v < VPL; v++) {
          pr[v][0] = x[u][v].x * y[u][v].x;
          pr[v][1] = x[u][v].y * y[u][v].y;
          pr[v][2] = x[u][v].z * y[u][v].z;
          pr[v][3] = x[u][v].w * y[u][v].w;
        }
        // stage t finalises the lanes c == t: dot = ((0 + w0*v0) + w1*v1) + ... in factor order
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < LPP; t++) {
          float in = LPP > 1 ? pred_lane_below(dot) : 0.0f;
          in = c == 0 ? 0.0f : in;
#pragma unroll
          for (int v = 0; v < VPL; v++) {
            in = in + pr[v][0];
            in = in + pr[v][1];
            in = in + pr[v][2];
            in = in + pr[v][3];
          }
          dot = in;
        }
        const float term = dot * xa[u] * xb[u];
        if (c == LPP - 1 && ok[u]) terms[(st + u) * PPS + g] = term;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int t0 = 0; t0 < cnt; t0 += 64) {
      const float t = t0 + lane < cnt ? terms[t0 + lane] : -0.0f;
      const float run = wave_sequential_prefix(result, t);
      result = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(run), 63));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }

  if (lane == 0) {
    out[r] = output_prob ? sigmoid_ref(result) : result;
    if (rows.label) s.loss[r] = logloss_ref(rows.label[r], result);
  }
}

}  // namespace ftrl_dev
