// kernels_fm.h -- FM rows, one WAVE per row, lane = factor (n_factors <= 64).
// Replaces, for a whole block of rows at once,
//   FM::update_vector_w / compute_fm_logit     src/model/fm.cpp:69-78 / :40-67
//   FtrlModel::update_linear_w / update_bias / compute_linear_logit   src/model/ftrl_model.cpp:44-64
//   tmp_grad = sigmoid(logit) - y              src/model/fm.cpp:27
// and, for the features that occur ONCE in the block (most of them under a Zipf law over 10 M ids),
//   FM::update_vector_nz                       src/model/fm.cpp:80-101
//   FtrlModel::update_linear_nz                src/model/ftrl_model.cpp:66-77
// and, with TRAIN = false, FM::predict (fm.cpp:34-38).
//
// Why: fm_row_kernel (kernels_row.h) spends a 256-thread workgroup per row,
// of which k threads walk the row's features for the factor sums with 4-byte loads and one thread
// adds the terms: 144 us per 8192-row block at k = 64 (0.09 of the HBM roof for config 4).  An FM
// record is [n | z | w] x k floats, so with lane = factor a wave reads a record's row as ONE
// coalesced 4k-byte load, the per-factor sums s_f = sum_a v_af x_a (fm.cpp:47-55: factor-outer,
// feature-inner, sequential) are one dependent chain PER LANE -- exactly the reference's order --
// and the ordered sums over the row's entries / over the factors are DPP prefix chains
// (wave_sequential_prefix).  The whole forward of a row needs no LDS and no barrier; 8192 rows are
// 8192 waves, all resident at once.
//
// The once-only features' (n, z) step needs tmp_grad, s_f and the row's value -- all in this wave's
// registers once the logit is known -- so it is applied here (one more trip to the record, which was
// read microseconds ago), as ffm_row_kernel does for FFM; fm_update_kernel / linear_update_kernel
// then skip the features with one occurrence.  Same touches, same order: bit-identical.
#pragma once
#include <type_traits>

#include "engine_types.h"
#include "kernels_update.h"

namespace ftrl_dev {

constexpr int kFmRowsPerBlock = 4;  // waves (= rows) per workgroup
constexpr int kFmChunk = 8;         // entries whose records are in flight together
// With lane = factor a record costs THREE registers per lane (n, z, w), so the training kernel can
// keep the first kFmPark entries' records from the refresh to the (n, z) step instead of reading them
// again.  Measured on config 4 (8192 x 39 rows, k = 64; tools/ab_c4.sh): parking 40 / 32 / 24 / 8
// entries (169 / 145 / 121 / 73 VGPRs, 3 / 3 / 4 / 6 waves per SIMD) gives 194 / 187 / 182 / 163 us per
// block -- the bytes saved (655 -> 418 MB) are worth less than the waves lost, so little is parked;
// the other entries are read again out of the L2 / Infinity Cache.  Four entries parked fit 61 VGPRs
// = 8 waves per SIMD, one wave per row of an 8192-row block resident at once: 176 -> 153 us (eight
// parked at 7 / 8 waves: 170 us / 178 us with 70 spilled registers).
#ifndef FFM_FM_PARK
#define FFM_FM_PARK 4
#endif
constexpr int kFmPark = FFM_FM_PARK;
constexpr int kFmParkStep = kFmPark < kFmChunk ? kFmPark : kFmChunk;  // parked entries walked at a time

#ifndef FFM_FM_WAVES
#define FFM_FM_WAVES 8
#endif
template <bool TRAIN>
__global__ __launch_bounds__(64 * kFmRowsPerBlock) __attribute__((amdgpu_waves_per_eu(TRAIN ? FFM_FM_WAVES : 8, 8)))
void fm_row_wave_kernel(ModelDev m, Rows rows, Scratch s,
                                                                           int max_row_nnz, float *out,
                                                                           int output_prob, int own_tg) {
  __shared__ uint64_t s_tab[32];  // expf's table (sigmoid_ref_tab)
  if (threadIdx.x < 32) s_tab[threadIdx.x] = kExpTab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r = wave_uniform(blockIdx.x * kFmRowsPerBlock + (threadIdx.x >> 6));
  if (r >= rows.n_rows) return;
  const int b = wave_uniform(rows.row_ptr[r]);
  const int nnz = wave_uniform(rows.row_ptr[r + 1]) - b;
  // a row beyond the validated capacity: the grouping has flagged the whole training block (no
  // kernel touches the model); a predict call flags it here.  Its outputs are NaN.
  if ((TRAIN && s.counters[CNT_ERROR]) || nnz > max_row_nnz) {
    if (lane == 0) {
      if (nnz > max_row_nnz) atomicOr(s.err, ERR_ROW_TOO_LONG);
      const float nan = __int_as_float(0x7fc00000);
      if (TRAIN) { s.logit[r] = nan; s.tg[r] = 0.0f; }
      s.loss[r] = static_cast<double>(nan);
      if (out) out[r] = nan;
    }
    return;
  }
  const int k = m.n_factors;
  const bool fl = lane < k;     // this lane carries a factor
  const int f = fl ? lane : 0;  // (idle lanes shadow factor 0, never store)
  const int64_t rec_stride = 3ll * k;

  // ---- linear part, lane = entry (64 per pass): update_linear_w, then bias + sum lin_w * x in row
  // order (compute_linear_logit, ftrl_model.cpp:44-50) as a strictly sequential DPP prefix.
  // remove_out_range (ftrl_model.cpp:36-42): an entry outside [0, n_feats) contributes nothing.
  float result;
  {
    float bias;
    if (TRAIN) {
      bias = ftrl_weight(m.h, m.bias3[1], m.bias3[2]);  // update_bias, ftrl_model.cpp:61-64
      if (r == 0 && lane == 0) m.bias3[0] = bias;
    } else {
      bias = m.bias3[0];
    }
    result = bias;
  }
  // the first pass's entry stays in registers for the once-only linear step below
  int e_i = -1, e_once = 0;
  float e_x = 0.0f, e_n = 0.0f, e_z = 0.0f, e_w = 0.0f;
  for (int a0 = 0; a0 < nnz; a0 += 64) {
    const int a = a0 + lane;
    int i = -1;
    float x = 0.0f, ln = 0.0f, lz = 0.0f, lw = 0.0f;
    if (a < nnz) {
      i = rows.feat[b + a];
      x = rows.val[b + a];
      if (i < 0 || i >= m.n_feats) i = -1;
    }
    if (i >= 0) {
      if (TRAIN) {
        ln = m.lin_n[i];
        lz = m.lin_z[i];
        lw = ftrl_weight(m.h, ln, lz);  // update_linear_w, ftrl_model.cpp:52-59
        m.lin_w[i] = lw;
      } else {
        lw = m.lin_w[i];
      }
    }
    // x + -0.0f == x bit for bit: idle lanes and erased entries add nothing
    const float run = wave_sequential_prefix(result, i >= 0 ? lw * x : -0.0f);
    result = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(run), 63));
    if (a0 == 0) {
      e_i = i; e_x = x; e_n = ln; e_z = lz; e_w = lw;
      if (TRAIN && own_tg && i >= 0) e_once = s.occpos[b + a] == OCC_ONCE;
    }
  }

  // ---- factor part, lane = factor: w = W(n, z) of every entry's record (FM::update_vector_w,
  // fm.cpp:69-78), s_f = sum v*x, q_f = sum (v*x)^2 in entry order (fm.cpp:47-55)
  float s_vx = 0.0f, sum_sqr = 0.0f;
  // The parked entries are walked in steps of kFmParkStep (a chunk, or all of them when fewer are
  // parked than a chunk holds): every step reads and writes exactly the registers it is handed.
  static_assert(kFmPark >= 1 && kFmPark % kFmParkStep == 0, "parked entries are whole steps");
  constexpr int kPark = TRAIN ? kFmPark : 1;  // (predict parks nothing)
  float pn[kPark], pz[kPark], pw[kPark];      // records of entries 0 .. kFmPark-1 (TRAIN)
  // one chunk of N entries: ids / values (wave-uniform scalar loads), the records' loads in flight
  // together, then W(n, z) and the running sums; (n, z, w) stay in kn / kz / kw[0 .. N)
  auto chunk = [&](int a0, float *kn, float *kz, float *kw, auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    int ids[N];
    float xs[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int a = a0 + j;
      int i = -1;
      float x = 0.0f;
      if (a < nnz) {
        i = rows.feat[b + a];
        x = rows.val[b + a];
        if (i < 0 || i >= m.n_feats) i = -1;
      }
      ids[j] = wave_uniform(i);
      xs[j] = __int_as_float(wave_uniform(__float_as_int(x)));
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (ids[j] < 0) continue;
      const float *rec = m.lat + ids[j] * rec_stride;
      if (TRAIN) {
        kn[j] = rec[LAT_N * k + f];
        kz[j] = rec[LAT_Z * k + f];
        kw[j] = m.h.learn ? rec[LAT_W * k + f] : 0.0f;
      } else {
        kw[j] = rec[LAT_W * k + f];
      }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (ids[j] < 0) continue;
      if (TRAIN) {
        kw[j] = latent_weight(m.h, kn[j], kz[j], kw[j]);
        if (fl) m.lat[ids[j] * rec_stride + LAT_W * k + f] = kw[j];
      }
      const float vx = kw[j] * xs[j];
      s_vx += vx;
      sum_sqr += vx * vx;
    }
  };
  if (TRAIN) {
#pragma unroll
    for (int c0 = 0; c0 < kFmPark; c0 += kFmParkStep)
      if (c0 < nnz) chunk(c0, pn + c0, pz + c0, pw + c0, std::integral_constant<int, kFmParkStep>{});
  }
  for (int a0 = TRAIN ? kFmPark : 0; a0 < nnz; a0 += kFmChunk) {
    float tn[kFmChunk], tz[kFmChunk], tw[kFmChunk];
    chunk(a0, tn, tz, tw, std::integral_constant<int, kFmChunk>{});
  }
  if (TRAIN && fl) s.svx[static_cast<int64_t>(r) * k + f] = s_vx;  // sum_vx (fm.h:24) for the update kernels
  // logit += 0.5 * (s_f^2 - q_f), factor after factor (fm.cpp:56-64)
  {
    const float term = fl ? 0.5f * (s_vx * s_vx - sum_sqr) : -0.0f;
    const float run = wave_sequential_prefix(result, term);
    result = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(run), 63));
  }

  if (!TRAIN) {
    if (lane == 0) {
      out[r] = output_prob ? sigmoid_ref(result) : result;
      if (rows.label) s.loss[r] = logloss_ref(rows.label[r], result);
    }
    return;
  }
  if (lane == 0) s.logit[r] = result;
  if (!own_tg) return;  // (a caller that sums logits elsewhere: tmp_grad_kernel and the update kernels follow)

  // ---- tmp_grad = sigmoid(logit) - y (fm.cpp:27), the row's logloss (ftrl_offline.cpp:80)
  const int y = rows.label[r];
  const float tg = sigmoid_ref_tab(result, s_tab) - static_cast<float>(y);
  if (lane == 0) {
    s.tg[r] = tg;
    s.loss[r] = logloss_ref(y, result);
    if (out) out[r] = result;
  }

  // ---- the features that occur nowhere else in the block: their one touch, right here.
  // linear (update_linear_nz, ftrl_model.cpp:66-77), lane = entry; the first 64 entries' (n, z, w)
  // are still in registers
  for (int a0 = 0; a0 < nnz; a0 += 64) {
    int i = e_i, once = e_once;
    float x = e_x, ln = e_n, lz = e_z, lw = e_w;
    if (a0 > 0) {
      const int a = a0 + lane;
      i = -1;
      once = 0;
      if (a < nnz) {
        i = rows.feat[b + a];
        x = rows.val[b + a];
        if (i < 0 || i >= m.n_feats) i = -1;
      }
      if (i >= 0) once = s.occpos[b + a] == OCC_ONCE;
      if (i >= 0 && once) { ln = m.lin_n[i]; lz = m.lin_z[i]; lw = m.lin_w[i]; }
    }
    if (i >= 0 && once) {
      nz_step_linear(m.h, lw, tg * x, ln, lz);
      m.lin_n[i] = ln;
      m.lin_z[i] = lz;
    }
  }
  // latent (FM::update_vector_nz, fm.cpp:80-101), lane = factor: g = tmp_grad * (x * s_f - v * x^2)
  auto update_chunk = [&](int a0, float *kn, float *kz, float *kw, bool parked, auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    int ids[N];
    float xs[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int a = a0 + j;
      int i = -1;
      float x = 0.0f;
      if (a < nnz) {
        i = rows.feat[b + a];
        x = rows.val[b + a];
        if (i < 0 || i >= m.n_feats || s.occpos[b + a] != OCC_ONCE) i = -1;
      }
      ids[j] = wave_uniform(i);
      xs[j] = __int_as_float(wave_uniform(__float_as_int(x)));
    }
    if (!parked) {
#pragma unroll
      for (int j = 0; j < N; j++) {
        if (ids[j] < 0) continue;
        const float *rec = m.lat + ids[j] * rec_stride;
        kn[j] = rec[LAT_N * k + f];
        kz[j] = rec[LAT_Z * k + f];
        kw[j] = rec[LAT_W * k + f];
      }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (ids[j] < 0) continue;
      const float x = xs[j];
      const float g = tg * (x * s_vx - kw[j] * x * x);  // fm.cpp:84-95
      nz_step_latent(m.h, kw[j], g, kn[j], kz[j]);
      if (fl) {
        float *rec = m.lat + ids[j] * rec_stride;
        rec[LAT_N * k + f] = kn[j];
        rec[LAT_Z * k + f] = kz[j];
      }
    }
  };
#pragma unroll
  for (int c0 = 0; c0 < kFmPark; c0 += kFmParkStep)
    if (c0 < nnz) update_chunk(c0, pn + c0, pz + c0, pw + c0, true, std::integral_constant<int, kFmParkStep>{});
  for (int a0 = kFmPark; a0 < nnz; a0 += kFmChunk) {
    float tn[kFmChunk], tz[kFmChunk], tw[kFmChunk];
    update_chunk(a0, tn, tz, tw, false, std::integral_constant<int, kFmChunk>{});
  }
}

}  // namespace ftrl_dev
