// kernels_row.h -- row-major half of the path: lazy weight refresh + forward logit, one workgroup
// per row.  Replaces, for a whole block of rows at once,
//   FtrlModel::update_linear_w / update_bias      src/model/ftrl_model.cpp:52-64
//   FFM::update_vector_w / compute_ffm_logit      src/model/ffm.cpp:72-88 / :57-70
//   FM::update_vector_w / compute_fm_logit        src/model/fm.cpp:69-78 / :40-67
//   FtrlModel::compute_linear_logit               src/model/ftrl_model.cpp:44-50
// and, with TRAIN = false, the predict() bodies (ffm.cpp:51-55, fm.cpp:34-38, lr.cpp:20-24).
//
// Numerics: each pair's dot product is a k-long sequential fp32 chain (std::inner_product, init
// 0.0f), the term is (dot*x1)*x2, and the terms are added to the linear logit in the reference's
// pair order by one lane, so the logit is the bit pattern the reference computes.
#pragma once
#include "engine_types.h"

namespace ftrl_dev {

constexpr int kRowThreads = 256;
constexpr int kTermsCap = 2048;  // pair terms staged in LDS per pass

// Dynamic LDS carve of the row kernels (16-byte aligned base, guide G17).
struct RowLds {
  int *idx;      // [max_row_nnz] positions (relative to the row start) of the surviving entries
  float *linw;   // [max_row_nnz] their linear weights
  float *terms;  // [kTermsCap]
  int *fcnt;     // [n_fields] surviving entries per field
};
__host__ __device__ inline size_t row_lds_bytes(int max_row_nnz, int n_fields) {
  return sizeof(int) * (size_t)max_row_nnz + sizeof(float) * (size_t)max_row_nnz +
         sizeof(float) * kTermsCap + sizeof(int) * (size_t)(n_fields > 0 ? n_fields : 1) + 64;
}
__device__ __forceinline__ RowLds carve_row_lds(char *base, int max_row_nnz) {
  RowLds l;
  l.idx = reinterpret_cast<int *>(base);
  l.linw = reinterpret_cast<float *>(l.idx + max_row_nnz);
  l.terms = l.linw + max_row_nnz;
  l.fcnt = reinterpret_cast<int *>(l.terms + kTermsCap);
  return l;
}

// Compacts the surviving entries of the row (remove_out_range) into lds.idx, in row order.
// Executed by wave 0; returns the count through *nv_out (LDS).
__device__ __forceinline__ void compact_row(const ModelDev &m, const Rows &rows, int b, int nnz,
                                            int *idx, int *nv_out) {
  if (threadIdx.x < 64) {
    int nv = 0;
    for (int base = 0; base < nnz; base += 64) {
      const int p = base + threadIdx.x;
      bool valid = false;
      if (p < nnz) {
        const int i = rows.feat[b + p];
        valid = i >= 0 && i < m.n_feats;
        if (m.type == 2) {
          const int f = rows.field[b + p];
          valid = valid && f >= 0 && f < m.n_fields;
        }
      }
      const unsigned long long mask = __ballot(valid);
      if (valid) idx[nv + __popcll(mask & ((1ull << threadIdx.x) - 1ull))] = p;
      nv += __popcll(mask);
    }
    if (threadIdx.x == 0) *nv_out = nv;
  }
}

// pair number q (reference order: a outer, b inner, a < b) -> (a, b) over nv entries
__device__ __forceinline__ void unrank_pair(int q, int nv, int &a, int &b) {
  // rows of the strict upper triangle have nv-1, nv-2, ... entries
  const float fn = 2.0f * nv - 1.0f;
  int aa = static_cast<int>((fn - sqrtf(fn * fn - 8.0f * q)) * 0.5f);
  if (aa < 0) aa = 0;
  if (aa > nv - 2) aa = nv - 2;
  // first pair of row aa: aa*nv - aa*(aa+1)/2
  while (aa > 0 && aa * nv - aa * (aa + 1) / 2 > q) aa--;
  while ((aa + 1) * nv - (aa + 1) * (aa + 2) / 2 <= q) aa++;
  a = aa;
  b = q - (aa * nv - aa * (aa + 1) / 2) + aa + 1;
}

// ------------------------------------------------------------------------------------------
// FFM (and LR when row_len == 0): one workgroup per row.
// TRAIN: refresh linear/bias/latent weights of everything the row touches, build the row's
// per-field entry chains for the update kernel, write the (partial) logit.
// !TRAIN: logit from the stored weights; out = logit or sigmoid(logit); per-row loss if labelled.
// ------------------------------------------------------------------------------------------
template <bool TRAIN>
__global__ __launch_bounds__(kRowThreads) void ffm_row_kernel(ModelDev m, Rows rows, Scratch s,
                                                              int max_row_nnz, float *out,
                                                              int output_prob) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_nv;
  RowLds lds = carve_row_lds(smem, max_row_nnz);
  const int r = blockIdx.x;
  const int b = rows.row_ptr[r];
  int nnz = rows.row_ptr[r + 1] - b;
  if (nnz > max_row_nnz) {
    if (threadIdx.x == 0) atomicOr(&s.counters[CNT_ERROR], ERR_ROW_TOO_LONG);
    nnz = max_row_nnz;
  }
  const int F = m.n_fields, k = m.n_factors, RL = m.row_len;
  const bool is_ffm = m.type == 2;
  const bool lin_owner = m.shard_rank == 0;

  for (int f = threadIdx.x; f < F; f += blockDim.x) lds.fcnt[f] = 0;
  compact_row(m, rows, b, nnz, lds.idx, &s_nv);
  __syncthreads();
  const int nv = s_nv;

  if (is_ffm) {
    for (int a = threadIdx.x; a < nv; a += blockDim.x)
      atomicAdd(&lds.fcnt[rows.field[b + lds.idx[a]]], 1);
    __syncthreads();
  }

  if (TRAIN && is_ffm) {
    // per-field chains of this row's surviving entries, ascending position
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
      int head = -1, prev = -1;
      if (lds.fcnt[f] > 0)
        for (int a = 0; a < nv; a++) {
          const int p = b + lds.idx[a];
          if (rows.field[p] == f) {
            if (prev < 0) head = p; else s.next[prev] = p;
            prev = p;
          }
        }
      if (prev >= 0) s.next[prev] = -1;
      s.head[static_cast<int64_t>(r) * F + f] = head;
    }
    // lazy refresh of every slot (feature a, partner field fp) that a pair of this row touches:
    // FFM::update_vector_w, ffm.cpp:72-88
    const int total = nv * RL;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      const int a = t / RL, e = t - a * RL;
      const int fp = e / k;
      const int p = b + lds.idx[a];
      const int fa = rows.field[p];
      const bool touched = (lds.fcnt[fp] - (fa == fp ? 1 : 0)) > 0 && owns_pair(m, fa, fp);
      if (touched) {
        float *row = lat_row(m, rows.feat[p]);
        row[LAT_W * RL + e] = ftrl_weight(m.h, row[LAT_N * RL + e], row[LAT_Z * RL + e]);
      }
    }
  }

  // linear weights of the surviving entries (update_linear_w, ftrl_model.cpp:52-59)
  for (int a = threadIdx.x; a < nv; a += blockDim.x) {
    const int i = rows.feat[b + lds.idx[a]];
    float lw;
    if (TRAIN) {
      lw = ftrl_weight(m.h, m.lin_n[i], m.lin_z[i]);
      if (lin_owner) m.lin_w[i] = lw;
    } else {
      lw = m.lin_w[i];
    }
    lds.linw[a] = lw;
  }
  __syncthreads();  // this row's refreshed weights are now readable by the whole workgroup

  // linear logit, sequential in row order (compute_linear_logit, ftrl_model.cpp:44-50)
  float result = 0.0f;
  if (threadIdx.x == 0) {
    if (lin_owner) {
      float bias;
      if (TRAIN) {
        bias = ftrl_weight(m.h, m.bias3[1], m.bias3[2]);  // update_bias, ftrl_model.cpp:61-64
        if (r == 0) m.bias3[0] = bias;
      } else {
        bias = m.bias3[0];
      }
      result = bias;
      for (int a = 0; a < nv; a++) result = result + lds.linw[a] * rows.val[b + lds.idx[a]];
    }
  }

  if (is_ffm && nv > 1) {
    const int n_pairs = nv * (nv - 1) / 2;
    for (int q0 = 0; q0 < n_pairs; q0 += kTermsCap) {
      const int q1 = min(q0 + kTermsCap, n_pairs);
      for (int q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
        int a, bb;
        unrank_pair(q, nv, a, bb);
        const int pa = b + lds.idx[a], pb = b + lds.idx[bb];
        const int fa = rows.field[pa], fb = rows.field[pb];
        float term = 0.0f;
        bool own = owns_pair(m, fa, fb);
        if (own) {
          const float *va = lat_row(m, rows.feat[pa]) + LAT_W * RL + fb * k;
          const float *vb = lat_row(m, rows.feat[pb]) + LAT_W * RL + fa * k;
          float dot = 0.0f;
          for (int f = 0; f < k; f++) dot = dot + va[f] * vb[f];
          term = dot * rows.val[pa] * rows.val[pb];
        }
        // non-owned pairs contribute nothing on this shard; mark them so the sum skips them
        lds.terms[q - q0] = own ? term : __int_as_float(0x7fc00001);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        if (m.n_shards <= 1) {
          for (int j = 0; j < q1 - q0; j++) result += lds.terms[j];
        } else {
          for (int j = 0; j < q1 - q0; j++) {
            const float t = lds.terms[j];
            if (__float_as_int(t) != 0x7fc00001) result += t;
          }
        }
      }
      __syncthreads();
    }
  }

  if (threadIdx.x == 0) {
    if (TRAIN) {
      s.logit[r] = result;
    } else {
      out[r] = output_prob ? sigmoid_ref(result) : result;
      if (rows.label) s.loss[r] = logloss_ref(rows.label[r], result);
    }
  }
}

// ------------------------------------------------------------------------------------------
// FM: one workgroup per row.  fm.cpp:40-67: factor-outer, feature-inner, all sequential fp32.
// ------------------------------------------------------------------------------------------
template <bool TRAIN>
__global__ __launch_bounds__(kRowThreads) void fm_row_kernel(ModelDev m, Rows rows, Scratch s,
                                                             int max_row_nnz, float *out,
                                                             int output_prob) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_nv;
  RowLds lds = carve_row_lds(smem, max_row_nnz);
  const int r = blockIdx.x;
  const int b = rows.row_ptr[r];
  int nnz = rows.row_ptr[r + 1] - b;
  if (nnz > max_row_nnz) {
    if (threadIdx.x == 0) atomicOr(&s.counters[CNT_ERROR], ERR_ROW_TOO_LONG);
    nnz = max_row_nnz;
  }
  const int k = m.n_factors;
  compact_row(m, rows, b, nnz, lds.idx, &s_nv);
  __syncthreads();
  const int nv = s_nv;

  if (TRAIN) {  // FM::update_vector_w, fm.cpp:69-78
    const int total = nv * k;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      const int a = t / k, e = t - a * k;
      float *row = lat_row(m, rows.feat[b + lds.idx[a]]);
      row[LAT_W * k + e] = ftrl_weight(m.h, row[LAT_N * k + e], row[LAT_Z * k + e]);
    }
  }
  for (int a = threadIdx.x; a < nv; a += blockDim.x) {
    const int i = rows.feat[b + lds.idx[a]];
    float lw;
    if (TRAIN) {
      lw = ftrl_weight(m.h, m.lin_n[i], m.lin_z[i]);
      m.lin_w[i] = lw;
    } else {
      lw = m.lin_w[i];
    }
    lds.linw[a] = lw;
  }
  __syncthreads();

  // per factor: s = sum v*x, q = sum (v*x)^2, term = 0.5*(s*s - q)   (k <= kTermsCap)
  for (int f = threadIdx.x; f < k; f += blockDim.x) {
    float s_vx = 0.0f, sum_sqr = 0.0f;
    for (int a = 0; a < nv; a++) {
      const int p = b + lds.idx[a];
      const float vx = lat_row(m, rows.feat[p])[LAT_W * k + f] * rows.val[p];
      s_vx += vx;
      sum_sqr += vx * vx;
    }
    if (TRAIN) s.svx[static_cast<int64_t>(r) * k + f] = s_vx;
    lds.terms[f] = 0.5f * (s_vx * s_vx - sum_sqr);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float bias;
    if (TRAIN) {
      bias = ftrl_weight(m.h, m.bias3[1], m.bias3[2]);
      if (r == 0) m.bias3[0] = bias;
    } else {
      bias = m.bias3[0];
    }
    float result = bias;
    for (int a = 0; a < nv; a++) result = result + lds.linw[a] * rows.val[b + lds.idx[a]];
    for (int f = 0; f < k; f++) result += lds.terms[f];
    if (TRAIN) {
      s.logit[r] = result;
    } else {
      out[r] = output_prob ? sigmoid_ref(result) : result;
      if (rows.label) s.loss[r] = logloss_ref(rows.label[r], result);
    }
  }
}

// tmp_grad = sigmoid(logit) - y (ffm.cpp:44, fm.cpp:27, lr.cpp:13) and the row's logloss
// (ftrl_offline.cpp:80) from the full logit (after the cross-shard sum when sharded).
__global__ void tmp_grad_kernel(int n_rows, const float *logit, const int *label, float *tg,
                                double *loss, float *logit_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const float lg = logit[r];
  const int y = label[r];
  tg[r] = sigmoid_ref(lg) - static_cast<float>(y);
  loss[r] = logloss_ref(y, lg);
  if (logit_out) logit_out[r] = lg;
}

// Deterministic sum of the per-row losses (fixed order: 256 strided partials, then a tree).
__global__ __launch_bounds__(256) void loss_sum_kernel(int n_rows, const double *loss,
                                                       double *out) {
  __shared__ double part[256];
  double acc = 0.0;
  for (int r = threadIdx.x; r < n_rows; r += 256) acc += loss[r];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = part[0];
}

}  // namespace ftrl_dev
