// kernels_fused.h -- the whole per-row half of one FFM training step in ONE pass over the row's
// slots: one workgroup per row, one THREAD per feature pair.  For the pair (a, b) of the row
// (a < b in row order) the thread owns the two slots the pair touches, A = (feature a, field of b)
// and B = (feature b, field of a), and does for them everything the reference does in train():
//   FFM::update_vector_w       src/model/ffm.cpp:72-88   w = W(n, z), stored            (read n,z; write w)
//   FFM::compute_ffm_logit     src/model/ffm.cpp:57-70   dot(wA, wB) * xa * xb           (registers)
//   FFM::update_vector_nz      src/model/ffm.cpp:90-136  (n, z) += ...  incl. the :118 quirk  (write n,z)
// with (n, z, w) of both slots held in registers from the first load to the last store, so a slot
// of a feature that occurs ONCE in the block moves exactly the bytes the algorithm needs: 8 in,
// 12 out per factor.  Features that occur several times in the block cannot be finished here
// (their touches must be applied in row order across rows): for them the thread stores w and, when
// the feature is hot, appends the partner's weights to the occurrence-ordered stream s.pstream
// that the feature-major update kernels (kernels_update.h) then read instead of gathering.
//
// The logit is the same bit pattern as the reference's: each pair's dot product is the k-long
// sequential fp32 chain of std::inner_product, the term is (dot*xa)*xb, and wave 0 adds the terms
// to the linear part in the reference's pair order, one dependent add after another.
//
// Eligibility (otherwise the block takes the general kernels in kernels_row.h): FFM, one shard,
// n_factors in {4, 8, 16}, every row with at most kFusedMaxNv surviving entries and at most one
// per field (row_shape_kernel decides per block).
#pragma once
#include "kernels_row.h"
#include "kernels_update.h"

namespace ftrl_dev {

constexpr int kFusedThreads = 768;                // >= kFusedMaxNv*(kFusedMaxNv-1)/2 pairs
constexpr int kFusedMaxNv = 39;
constexpr int kFusedTermsPerLane = kFusedThreads / 64;

// The slot-wide operations in chunks of kFusedChunk factors: enough independent work per wave vote
// to fill the pipeline, few enough temporaries to stay inside the register budget.
constexpr int kFusedChunk = 4;

template <int K>
__device__ __forceinline__ void refresh_chunks(const Hyper &h, const float (&n)[K],
                                               const float (&z)[K], float (&w)[K]) {
  constexpr int C = K < kFusedChunk ? K : kFusedChunk;
#pragma unroll
  for (int c0 = 0; c0 < K; c0 += C) {
    float nn[C], zz[C], ww[C];
#pragma unroll
    for (int i = 0; i < C; i++) { nn[i] = n[c0 + i]; zz[i] = z[c0 + i]; }
    ftrl_weight_n<C>(h, nn, zz, ww);
#pragma unroll
    for (int i = 0; i < C; i++) w[c0 + i] = ww[i];
  }
}

template <int K>
__device__ __forceinline__ void touch_chunks(const Hyper &h, bool own_first, float tg, float x_own,
                                             float x_other, const float (&vp)[K],
                                             const float (&w)[K], float (&n)[K], float (&z)[K]) {
  constexpr int C = K < kFusedChunk ? K : kFusedChunk;
#pragma unroll
  for (int c0 = 0; c0 < K; c0 += C) {
    float vv[C], ww[C], nn[C], zz[C];
#pragma unroll
    for (int i = 0; i < C; i++) { vv[i] = vp[c0 + i]; ww[i] = w[c0 + i]; nn[i] = n[c0 + i]; zz[i] = z[c0 + i]; }
    ffm_touch_n<C>(h, own_first, tg, x_own, x_other, vv, ww, nn, zz);
#pragma unroll
    for (int i = 0; i < C; i++) { n[c0 + i] = nn[i]; z[c0 + i] = zz[i]; }
  }
}

template <int K4>
__global__ __launch_bounds__(kFusedThreads) void ffm_fused_row_kernel(ModelDev m, Rows rows,
                                                                      Scratch s, int max_row_nnz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_nv;
  __shared__ float s_tg;
  __shared__ uint64_t s_tab[32];
  if (s.counters[CNT_NOFUSE]) return;  // this block takes the general path
  constexpr int K = 4 * K4;
  const int F = m.n_fields, RL = m.row_len, RL4 = RL >> 2;
  RowLds lds = carve_row_lds(smem, max_row_nnz, F);
  const int tid = threadIdx.x;
  const int r = blockIdx.x;
  const int b = rows.row_ptr[r];
  int nnz = rows.row_ptr[r + 1] - b;
  if (nnz > max_row_nnz) {
    if (tid == 0) atomicOr(&s.counters[CNT_ERROR], ERR_ROW_TOO_LONG);
    nnz = max_row_nnz;
  }
  const int label = tid < 64 ? rows.label[r] : 0;
  for (int f = tid; f < F; f += blockDim.x) { lds.fcnt[f] = 0; lds.ffirst[f] = -1; }
  if (tid < 32) s_tab[tid] = kExpTab[tid];
  stage_row(m, rows, b, nnz, lds, &s_nv);
  __syncthreads();
  const int nv = s_nv;
  const int n_pairs = nv * (nv - 1) / 2;
  const bool has_pairs = n_pairs > 0;      // workgroup-uniform
  const bool act = tid < n_pairs;          // idle threads shadow pair 0 and never store

  // ---- both slots' accumulators: in flight while the row's tables are published ----
  int a = 0, bb = 1;
  if (has_pairs) unrank_pair(act ? tid : 0, nv, a, bb);
  int fa = 0, fb = 0;
  float4 *recA = nullptr, *recB = nullptr;  // 16-byte view of slot A / B inside the n row
  float nA[K], zA[K], nB[K], zB[K];
  if (has_pairs) {
    fa = lds.field[a];
    fb = lds.field[bb];
    recA = reinterpret_cast<float4 *>(lat_row(m, lds.feat[a]) + fb * K);
    recB = reinterpret_cast<float4 *>(lat_row(m, lds.feat[bb]) + fa * K);
#pragma unroll
    for (int i = 0; i < K4; i++) {
      const float4 n1 = recA[LAT_N * RL4 + i], z1 = recA[LAT_Z * RL4 + i];
      const float4 n2 = recB[LAT_N * RL4 + i], z2 = recB[LAT_Z * RL4 + i];
      nA[4 * i] = n1.x; nA[4 * i + 1] = n1.y; nA[4 * i + 2] = n1.z; nA[4 * i + 3] = n1.w;
      zA[4 * i] = z1.x; zA[4 * i + 1] = z1.y; zA[4 * i + 2] = z1.z; zA[4 * i + 3] = z1.w;
      nB[4 * i] = n2.x; nB[4 * i + 1] = n2.y; nB[4 * i + 2] = n2.z; nB[4 * i + 3] = n2.w;
      zB[4 * i] = z2.x; zB[4 * i + 1] = z2.y; zB[4 * i + 2] = z2.z; zB[4 * i + 3] = z2.w;
    }
  }

  for (int t = tid; t < nv; t += blockDim.x) {
    atomicAdd(&lds.fcnt[lds.field[t]], 1);
    atomicMin(reinterpret_cast<unsigned *>(&lds.ffirst[lds.field[t]]), static_cast<unsigned>(t));
  }
  __syncthreads();
  publish_row_tables(m, rows, s, lds, r, b, nv, F);

  // linear weights of the surviving entries (update_linear_w, ftrl_model.cpp:52-59)
  for (int t = tid; t < nv; t += blockDim.x) {
    const int i = lds.feat[t];
    const float lw = ftrl_weight(m.h, m.lin_n[i], m.lin_z[i]);
    m.lin_w[i] = lw;
    lds.linw[t] = lw;
  }

  // ---- refresh both slots (ffm.cpp:72-88), store w, feed the hot features' streams, the term ----
  float wA[K], wB[K];
  float xa = 0.0f, xb = 0.0f;
  int opa = OCC_FEW, opb = OCC_FEW;
  if (has_pairs) {
    refresh_chunks<K>(m.h, nA, zA, wA);
    refresh_chunks<K>(m.h, nB, zB, wB);
    xa = lds.val[a];
    xb = lds.val[bb];
    opa = lds.opos[a];   // published above, behind a barrier
    opb = lds.opos[bb];
    if (act) {
#pragma unroll
      for (int i = 0; i < K4; i++) {
        recA[LAT_W * RL4 + i] = make_float4(wA[4 * i], wA[4 * i + 1], wA[4 * i + 2], wA[4 * i + 3]);
        recB[LAT_W * RL4 + i] = make_float4(wB[4 * i], wB[4 * i + 1], wB[4 * i + 2], wB[4 * i + 3]);
      }
      if (opa >= 0) {  // feature a is hot: its touch by this pair needs the partner's weights wB
        float4 *dst = reinterpret_cast<float4 *>(s.pstream + static_cast<int64_t>(opa) * RL + fb * K);
#pragma unroll
        for (int i = 0; i < K4; i++)
          dst[i] = make_float4(wB[4 * i], wB[4 * i + 1], wB[4 * i + 2], wB[4 * i + 3]);
      }
      if (opb >= 0) {
        float4 *dst = reinterpret_cast<float4 *>(s.pstream + static_cast<int64_t>(opb) * RL + fa * K);
#pragma unroll
        for (int i = 0; i < K4; i++)
          dst[i] = make_float4(wA[4 * i], wA[4 * i + 1], wA[4 * i + 2], wA[4 * i + 3]);
      }
      float dot = 0.0f;
#pragma unroll
      for (int i = 0; i < K; i++) dot = dot + wA[i] * wB[i];
      lds.terms[tid] = dot * xa * xb;
    }
  }
  __syncthreads();

  // ---- logit: bias + linear terms in row order + pair terms in pair order, strictly sequential
  //      (ftrl_model.cpp:44-50, ffm.cpp:57-70); tmp_grad (ffm.cpp:44).  Wave 0, all lanes alike. ----
  if (tid < 64) {
    float tl[kFusedTermsPerLane];
#pragma unroll
    for (int j = 0; j < kFusedTermsPerLane; j++) {
      const int q = tid * kFusedTermsPerLane + j;
      tl[j] = q < n_pairs ? lds.terms[q] : -0.0f;  // x + -0.0f == x bit for bit
    }
    const float bias = ftrl_weight(m.h, m.bias3[1], m.bias3[2]);  // update_bias, ftrl_model.cpp:61-64
    if (r == 0 && tid == 0) m.bias3[0] = bias;
    float result = bias;
    const float lp = tid < nv ? lds.linw[tid] * lds.val[tid] : 0.0f;  // nv <= kFusedMaxNv < 64
    for (int t = 0; t < nv; t++)
      result = result + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lp), t));
    const int lanes_used = (n_pairs + kFusedTermsPerLane - 1) / kFusedTermsPerLane;
    for (int l = 0; l < lanes_used; l++) {
#pragma unroll
      for (int j = 0; j < kFusedTermsPerLane; j++)
        result = result + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tl[j]), l));
    }
    const float tg = sigmoid_ref_tab(result, s_tab) - static_cast<float>(label);
    if (tid == 0) {
      s.logit[r] = result;
      s_tg = tg;
    }
  }
  __syncthreads();

  // ---- accumulator update of the slots whose feature occurs once in the block (ffm.cpp:90-136).
  //      Computed by every lane (the stages vote wave-wide), stored by the lanes it applies to. ----
  if (has_pairs) {
    const float tg = s_tg;
    touch_chunks<K>(m.h, true, tg, xa, xb, wB, wA, nA, zA);    // a is the pair's first entry
    touch_chunks<K>(m.h, false, tg, xb, xa, wA, wB, nB, zB);   // b its second (:117-120)
    if (act && opa == OCC_ONCE) {
#pragma unroll
      for (int i = 0; i < K4; i++) {
        recA[LAT_N * RL4 + i] = make_float4(nA[4 * i], nA[4 * i + 1], nA[4 * i + 2], nA[4 * i + 3]);
        recA[LAT_Z * RL4 + i] = make_float4(zA[4 * i], zA[4 * i + 1], zA[4 * i + 2], zA[4 * i + 3]);
      }
    }
    if (act && opb == OCC_ONCE) {
#pragma unroll
      for (int i = 0; i < K4; i++) {
        recB[LAT_N * RL4 + i] = make_float4(nB[4 * i], nB[4 * i + 1], nB[4 * i + 2], nB[4 * i + 3]);
        recB[LAT_Z * RL4 + i] = make_float4(zB[4 * i], zB[4 * i + 1], zB[4 * i + 2], zB[4 * i + 3]);
      }
    }
  }
}

}  // namespace ftrl_dev
