// kernels_update.h -- feature-major half of the path: the FTRL (n,z) accumulator update.
// Replaces, for a whole block of rows at once,
//   FtrlModel::update_linear_nz / update_bias_nz   src/model/ftrl_model.cpp:66-85
//   FFM::update_vector_nz                          src/model/ffm.cpp:90-136  (incl. :118)
//   FM::update_vector_nz                           src/model/fm.cpp:80-101
//
// Every distinct feature of the block has exactly one owner (one wave per 64 latent elements of
// its record).  The owner keeps (n, z, w) in registers and applies the block's touches to them in
// row order -- the same sequence of fp32 operations the reference's one-thread loop performs
// with w and tmp_grad frozen at the block start -- then writes (n, z) back once.  No float
// atomics, no locks, bit-reproducible.
#pragma once
#include "engine_types.h"

namespace ftrl_dev {

constexpr int kUpdThreads = 256;
constexpr int kUpdWaves = kUpdThreads / 64;

// FFM latent update.  Work item = (distinct feature u, chunk of 64 elements of its row_len).
__global__ __launch_bounds__(kUpdThreads) void ffm_update_kernel(ModelDev m, Rows rows,
                                                                 Scratch s) {
  const int RL = m.row_len, k = m.n_factors, F = m.n_fields;
  const int chunks = (RL + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kUpdWaves + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * kUpdWaves;
  const int64_t n_items = static_cast<int64_t>(s.counters[CNT_NUNIQ]) * chunks;
  for (int64_t item = wave; item < n_items; item += n_waves) {
    const int u = static_cast<int>(item / chunks);
    const int e = static_cast<int>(item - static_cast<int64_t>(u) * chunks) * 64 + lane;
    if (e >= RL) continue;
    const int fp = e / k, kk = e - fp * k;  // this lane's slot: partner field fp, factor kk
    const int i = s.uniq[u];
    const int start = s.ustart[u], c = s.ucount[u];
    float *rec = lat_row(m, i);
    float n = rec[LAT_N * RL + e], z = rec[LAT_Z * RL + e];
    const float w = rec[LAT_W * RL + e];
    bool touched = false;
    for (int t = 0; t < c; t++) {
      const int p = s.occ[start + t];  // an entry holding feature i
      const int r = s.row_of[p];
      const int fm = rows.field[p];
      if (!owns_pair(m, fm, fp)) continue;
      const float tg = s.tg[r];
      const float xm = rows.val[p];
      // every other surviving entry q of row r whose field is fp pairs with p through this slot
      for (int q = s.head[static_cast<int64_t>(r) * F + fp]; q >= 0; q = s.next[q]) {
        if (q == p) continue;
        const float vp = lat_row(m, rows.feat[q])[LAT_W * RL + fm * k + kk];  // partner slot w
        touched = true;
        if (p < q) {
          // p is the pair's first entry: slot (i, field2), ffm.cpp:112-115
          const float x = xm * rows.val[q];
          const float g1 = tg * vp * x;
          nz_step_latent(m.h, w, g1, n, z);
        } else {
          // p is the pair's second entry: slot (j, field1), ffm.cpp:117-120 with the :118 quirk
          const float x = rows.val[q] * xm;
          const float g2 = tg * vp * x;  // tmp_grad * vif1 * x
          const float g1 = tg * w * x;   // tmp_grad * vif2 * x (the first entry's gradient)
          nz_step_latent_jside(m.h, w, g2, g1, n, z);
        }
      }
    }
    if (touched) {
      rec[LAT_N * RL + e] = n;
      rec[LAT_Z * RL + e] = z;
    }
  }
}

// FM latent update.  Work item = (distinct feature u, chunk of 64 factors).
__global__ __launch_bounds__(kUpdThreads) void fm_update_kernel(ModelDev m, Rows rows, Scratch s) {
  const int k = m.n_factors;
  const int chunks = (k + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kUpdWaves + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * kUpdWaves;
  const int64_t n_items = static_cast<int64_t>(s.counters[CNT_NUNIQ]) * chunks;
  for (int64_t item = wave; item < n_items; item += n_waves) {
    const int u = static_cast<int>(item / chunks);
    const int e = static_cast<int>(item - static_cast<int64_t>(u) * chunks) * 64 + lane;
    if (e >= k) continue;
    const int i = s.uniq[u];
    const int start = s.ustart[u], c = s.ucount[u];
    float *rec = lat_row(m, i);
    float n = rec[LAT_N * k + e], z = rec[LAT_Z * k + e];
    const float w = rec[LAT_W * k + e];
    for (int t = 0; t < c; t++) {  // fm.cpp:84-95
      const int p = s.occ[start + t];
      const int r = s.row_of[p];
      const float x = rows.val[p];
      const float s_vx = s.svx[static_cast<int64_t>(r) * k + e];
      const float g = s.tg[r] * (x * s_vx - w * x * x);
      nz_step_latent(m.h, w, g, n, z);
    }
    rec[LAT_N * k + e] = n;
    rec[LAT_Z * k + e] = z;
  }
}

// Linear + bias update.  Thread per distinct feature (update_linear_nz, ftrl_model.cpp:66-77);
// the very first thread also walks all rows for the bias (update_bias_nz, :79-85).
__global__ __launch_bounds__(kUpdThreads) void linear_update_kernel(ModelDev m, Rows rows,
                                                                    Scratch s) {
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_uniq = s.counters[CNT_NUNIQ];
  for (int u = gtid; u < n_uniq; u += gridDim.x * blockDim.x) {
    const int i = s.uniq[u];
    const int start = s.ustart[u], c = s.ucount[u];
    float n = m.lin_n[i], z = m.lin_z[i];
    const float w = m.lin_w[i];
    for (int t = 0; t < c; t++) {
      const int p = s.occ[start + t];
      nz_step_linear(m.h, w, s.tg[s.row_of[p]] * rows.val[p], n, z);
    }
    m.lin_n[i] = n;
    m.lin_z[i] = z;
  }
}

__global__ void bias_update_kernel(ModelDev m, int n_rows, Scratch s) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float n = m.bias3[1], z = m.bias3[2];
  const float w = m.bias3[0];
  for (int r = 0; r < n_rows; r++) nz_step_linear(m.h, w, s.tg[r], n, z);
  m.bias3[1] = n;
  m.bias3[2] = z;
}

}  // namespace ftrl_dev
