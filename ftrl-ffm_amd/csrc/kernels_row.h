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
//
// Memory: a feature's record is [n row | z row | w row], each row_len floats, so the refresh of
// one row's nnz features is nnz pairs of contiguous 16-byte-vector streams in and one out; the
// pair phase re-reads the freshly written w slots (64 B each at k = 16) out of L2.
#pragma once
#include "engine_types.h"
#include "kernels_touch.h"


namespace ftrl_dev {

// The once-only features' (n, z) stream through the row kernel -- read by the refresh, read again and
// written back by the in-row update, touched by no other row of the block -- while the w rows of
// the features many rows share are what the 4 MB L2 of an XCD should keep: the (n, z) traffic
// carries the non-temporal hint (global_load / store ... nt).  Bits of FFM_ROW_NT: 1 the update's
// stores, 2 the refresh's loads, 4 the update's loads, 8 the update's w load.  Measured on C5
// (profiles/archive/r03_fused_row_experiment.txt): 7 -> row kernel 550 -> 540 us; 1, 2, 3, 4 alone: noise.
#ifndef FFM_ROW_NT
#define FFM_ROW_NT 7
#endif
// Experiment builds of the FFM row kernel (tools/ab_env.sh; profiles/r06_experiments.md).  Bit 1 is
// exact (the in-row update recomputes its own w from the (n, z) it holds instead of reading it back);
// every other bit drops a class of memory accesses and gives WRONG results -- timing only:
//   2 pair phase loads no weights | 4 update takes no second (n, z) read | 8 update loads no partner w
//   16 update stores nothing | 32 refresh stores no w | 64 no in-row update at all
#ifndef FFM_ROW_EXP
#define FFM_ROW_EXP 0
#endif
constexpr int kRowThreads = 256;
#ifndef FFM_ROW_MAXT
#define FFM_ROW_MAXT 256
#endif
constexpr int kRowMaxThreads = FFM_ROW_MAXT;  // launch bound of the FFM row kernel (experiments: 512 / 1024)
constexpr int kTermsCap = 2048;  // most pair terms staged in LDS per pass
// Terms buffer a row kernel actually needs (a multiple of 4, at most kTermsCap): all pairs of the
// longest admissible row -- or, on a shard, its entries times the slots of a record -- or the
// FM factor count.  Short rows then cost little LDS and many more of them are resident per CU.
__host__ __device__ inline int row_terms_cap(int max_row_nnz, int sharded_span, int fm_factors) {
  long long need = static_cast<long long>(max_row_nnz) * (max_row_nnz - 1) / 2;
  if (sharded_span > 0) need = static_cast<long long>(max_row_nnz) * sharded_span;
  if (fm_factors > need) need = fm_factors;
  if (need < 64) need = 64;
  if (need > kTermsCap) need = kTermsCap;
  return static_cast<int>((need + 3) & ~3ll);
}

// Dynamic LDS carve of the row kernels (16-byte aligned base, guide G17).  All per-entry arrays
// are indexed by the COMPACT index a of the surviving entries (remove_out_range applied).
struct RowLds {
  float *terms;  // [terms_cap]           (first: keeps 16-byte alignment for b128 reads)
  int *pos;      // [max_row_nnz] position of entry a inside the row
  int *field;    // [max_row_nnz]
  int *feat;     // [max_row_nnz]
  float *val;    // [max_row_nnz]
  float *linw;   // [max_row_nnz] linear weight
  int *opos;     // [max_row_nnz] occurrence position when the entry's feature is hot, else -1
  int *slist;    // [max_row_nnz] the entries this row refreshes itself (compact indices)
  int *sidx;     // [max_row_nnz] position of entry a in slist, -1 when another kernel refreshes it
  int *fcnt;     // [n_fields] surviving entries per field
  int *ffirst;   // [n_fields] compact index of the first entry of the field, -1 if none
};
__host__ __device__ inline size_t row_lds_bytes(int max_row_nnz, int n_fields, int terms_cap) {
  const size_t M = (size_t)((max_row_nnz + 3) & ~3), Fp = (size_t)((n_fields + 3) & ~3);
  return sizeof(float) * terms_cap + 8 * 4 * M + 2 * 4 * (Fp ? Fp : 4);
}
__device__ __forceinline__ RowLds carve_row_lds(char *base, int max_row_nnz, int n_fields, int terms_cap) {
  const int M = (max_row_nnz + 3) & ~3, Fp = ((n_fields + 3) & ~3) ? ((n_fields + 3) & ~3) : 4;
  RowLds l;
  l.terms = reinterpret_cast<float *>(base);
  l.pos = reinterpret_cast<int *>(l.terms + terms_cap);
  l.field = l.pos + M;
  l.feat = l.field + M;
  l.val = reinterpret_cast<float *>(l.feat + M);
  l.linw = l.val + M;
  l.opos = reinterpret_cast<int *>(l.linw + M);
  l.slist = l.opos + M;
  l.sidx = l.slist + M;
  l.fcnt = l.sidx + M;
  l.ffirst = l.fcnt + Fp;
  return l;
}

// Stages the surviving entries of the row (remove_out_range: ftrl_model.cpp:36-42, ffm.cpp:30-36)
// into LDS, in row order.  Executed by wave 0; the count goes to *nv_out.
// occpos (training FFM rows): the entries' occurrence classes ride along, so nothing later in the
// row waits on a load of its own for them.
__device__ __forceinline__ void stage_row(const ModelDev &m, const Rows &rows, int b, int nnz,
                                          RowLds &lds, int *nv_out, const int *occpos = nullptr) {
  if (threadIdx.x < 64) {
    int nv = 0;
    bool bad = false;
    for (int base = 0; base < nnz; base += 64) {
      const int p = base + threadIdx.x;
      bool valid = false;
      int i = 0, f = 0, op = 0;
      float x = 0.0f;
      if (p < nnz) {
        i = rows.feat[b + p];
        x = rows.val[b + p];
        if (occpos) op = occpos[b + p];
        valid = i >= 0 && i < m.n_feats;
        if (m.type == 2) {
          f = rows.field[b + p];
          valid = valid && f >= 0 && f < m.n_fields;
          // compact shard: an id outside its field's id range has no record here (group_keys_kernel
          // voids a TRAINING block for it; predict rows are caught here -- ADVICE r02)
          if (valid && m.field_start && (i < m.field_start[f] || i >= m.field_start[f + 1])) bad = true;
          if (valid) valid = keeps_field(m, f);  // a compact shard drops the columns it owns nothing of
        }
      }
      const unsigned long long mask = __ballot(valid);
      if (valid) {
        const int a = nv + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
        lds.pos[a] = p;
        lds.field[a] = f;
        lds.feat[a] = i;
        lds.val[a] = x;
        if (occpos) lds.opos[a] = op;
      }
      nv += __popcll(mask);
    }
    const bool any_bad = __any(bad);
    if (threadIdx.x == 0) *nv_out = any_bad ? -1 : nv;  // -1: an id outside its field's range
  }
}

// pair number q (reference order: a outer, b inner, a < b) -> (a, b) over nv entries
__device__ __forceinline__ void unrank_pair(int q, int nv, int &a, int &b) {
  const float fn = 2.0f * nv - 1.0f;
  int aa = static_cast<int>((fn - __fsqrt_rn(fn * fn - 8.0f * q)) * 0.5f);
  if (aa < 0) aa = 0;
  if (aa > nv - 2) aa = nv - 2;
  while (aa > 0 && aa * nv - aa * (aa + 1) / 2 > q) aa--;
  while ((aa + 1) * nv - (aa + 1) * (aa + 2) / 2 <= q) aa++;
  a = aa;
  b = q - (aa * nv - aa * (aa + 1) / 2) + aa + 1;
}

// Sequential sum of cnt staged terms into acc, in order (one lane; b128 LDS reads).
__device__ __forceinline__ float add_terms_in_order(const float *terms, int cnt, float acc) {
  int j = 0;
  const float4 *t4 = reinterpret_cast<const float4 *>(terms);
  for (; j + 4 <= cnt; j += 4) {
    const float4 t = t4[j >> 2];
    acc += t.x; acc += t.y; acc += t.z; acc += t.w;
  }
  for (; j < cnt; j++) acc += terms[j];
  return acc;
}

// TRAIN, FFM: loads the entries' occurrence classes and publishes the row's per-field tables for
// the feature-major update kernels.  Called by every thread of the workgroup (has barriers).
__device__ __forceinline__ void publish_row_tables(const ModelDev &m, const Rows &rows,
                                                   const Scratch &s, RowLds &lds, int r, int b,
                                                   int nv, int F) {
  // (lds.opos: staged with the row)
  // Per-field view of this row for the update kernel: rowtab[r][f] = {feat, val bits, entry
  // index, count} of the only entry with field f (entry index -1: none, -2: several -- then
  // head/next chains list them in row order).
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const int cnt = lds.fcnt[f];
    int4 t = make_int4(-1, 0, -1, cnt);
    int head = -1;
    if (cnt >= 1) {
      const int a0 = lds.ffirst[f];
      head = b + lds.pos[a0];
      t = make_int4(lds.feat[a0], __float_as_int(lds.val[a0]), cnt == 1 ? head : -2, cnt);
      if (cnt > 1) {
        int prev = head;
        for (int a = a0 + 1; a < nv; a++)
          if (lds.field[a] == f) {
            s.next[prev] = b + lds.pos[a];
            prev = b + lds.pos[a];
          }
        s.next[prev] = -1;
      } else {
        s.next[head] = -1;
      }
    }
    s.head[static_cast<int64_t>(r) * F + f] = head;
    s.rowtab[static_cast<int64_t>(r) * F + f] = t;
  }
}

// The same sum by one whole wave: lane 0's acc + terms[0] + terms[1] + ... strictly in order, one
// dependent add after another: lane l of a 64-term chunk adds its term to the running value of the
// lane below (wave_sequential_prefix: ONE DPP add per term; rounds 2-4 fed the chain by v_readlane +
// v_add, 24 cycles per term: C5 row kernel 515 -> 507 us, resident step 1.022 -> 0.989 ms).  Terms
// carrying the "not this shard's pair" tag are replaced by -0.0f (x + -0.0f == x bit for bit).
// Every lane returns the sum.
__device__ __forceinline__ float wave_add_terms_in_order(const float *terms, int cnt, float acc0) {
  const int lane = threadIdx.x & 63;
  float acc = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(acc0)));
  for (int base = 0; base < cnt; base += 64) {
    float t = base + lane < cnt ? terms[base + lane] : -0.0f;
    t = __float_as_int(t) == 0x7fc00001 ? -0.0f : t;
    const float run = wave_sequential_prefix(acc, t);
    acc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(run), 63));
  }
  return acc;
}

// ------------------------------------------------------------------------------------------
// FFM (and LR when row_len == 0): one workgroup per row.
// TRAIN: refresh linear/bias/latent weights of everything the row touches, publish the row's
// per-field tables for the update kernel, write the (partial) logit.
// !TRAIN: logit from the stored weights; out = logit or sigmoid(logit); per-row loss if labelled.
// VEC4: n_factors is a multiple of 4, so every slot is a whole number of 16-byte vectors.
// ------------------------------------------------------------------------------------------
#ifdef FFM_ROW_WAVES
#define FFM_ROW_OCC __attribute__((amdgpu_waves_per_eu(FFM_ROW_WAVES, FFM_ROW_WAVES)))
#else
#define FFM_ROW_OCC
#endif
// WHOLE: the kernel has the whole logit (one shard): it also produces tmp_grad / loss and may
// apply the once-only features' update (own_tg, refreshed == 3).  A shard's instantiation leaves all
// of that out -- and the registers it costs: more rows in flight per SIMD.
template <bool TRAIN, bool VEC4, bool WHOLE = TRAIN>
__global__ __launch_bounds__(kRowMaxThreads) FFM_ROW_OCC void ffm_row_kernel(ModelDev m, Rows rows, Scratch s,
                                                              int max_row_nnz, float *out,
                                                              int output_prob, int refreshed,
                                                              int own_tg_arg, int row0, int park_vecs) {
  const int own_tg = WHOLE ? own_tg_arg : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_nv, s_ns;
  __shared__ float s_tg;
  __shared__ uint64_t s_tab[32];  // expf's table, staged so the row's last step waits on no load
  if (TRAIN && WHOLE && own_tg && threadIdx.x < 32) s_tab[threadIdx.x] = kExpTab[threadIdx.x];
  const int F = m.n_fields, k = m.n_factors, RL = m.row_len;
  const int terms_cap = row_terms_cap(max_row_nnz, m.n_shards > 1 ? record_span(m, 1) : 0, 0);
  RowLds lds = carve_row_lds(smem, max_row_nnz, F, terms_cap);
  // the first park_vecs 16-byte vectors of w that this row's refresh computes for its once-only
  // features stay in LDS (behind the row's staging arrays): the pair phase and the in-row update take
  // them from there instead of reading them back through an L2 they have long left
  float4 *park_w = reinterpret_cast<float4 *>(smem + ((row_lds_bytes(max_row_nnz, F, terms_cap) + 15) & ~static_cast<size_t>(15)));
  const int r = blockIdx.x + row0;  // (row0: first row of this launch's row phase)
  const int b = rows.row_ptr[r];
  const int nnz = rows.row_ptr[r + 1] - b;
  // (evaluation rows of at most park_vecs entries were ffm_predict_wave_kernel's: this launch is for
  // the longer ones)
  if (!TRAIN && park_vecs > 0 && nnz <= park_vecs) return;
  // a row beyond the LDS capacity: the grouping has flagged the whole training block (no kernel
  // touches the model); a predict call flags it here.  Its outputs are NaN.
  if ((TRAIN && s.counters[CNT_ERROR]) || nnz > max_row_nnz) {
    if (threadIdx.x == 0) {
      if (nnz > max_row_nnz) atomicOr(s.err, ERR_ROW_TOO_LONG);
      const float nan = __int_as_float(0x7fc00000);
      if (TRAIN) { s.logit[r] = nan; s.tg[r] = 0.0f; }
      s.loss[r] = static_cast<double>(nan);
      if (out) out[r] = nan;
    }
    return;
  }
  const bool is_ffm = m.type == 2;

  for (int f = threadIdx.x; f < F; f += blockDim.x) { lds.fcnt[f] = 0; lds.ffirst[f] = -1; }
  stage_row(m, rows, b, nnz, lds, &s_nv, TRAIN && is_ffm ? s.occpos : nullptr);
  __syncthreads();
  const int nv = s_nv;
  if (nv < 0) {  // compact shard, an id outside its field's range: no record to read -- NaN row
    if (threadIdx.x == 0) {
      atomicOr(s.err, ERR_FIELD_MAP);
      const float nan = __int_as_float(0x7fc00000);
      if (TRAIN) { s.logit[r] = nan; s.tg[r] = 0.0f; }
      s.loss[r] = static_cast<double>(nan);
      if (out) out[r] = nan;
    }
    return;
  }

  // linear weights of the surviving entries (update_linear_w, ftrl_model.cpp:52-59).  The first
  // blockDim.x of them are fetched now into a register each, so that the loads fly while the row's
  // tables are built, and parked in LDS afterwards.
  // (a shard adds -- and refreshes -- the linear terms of the fields it owns them for only)
  auto linear_weight = [&](int a) {
    const int i = lds.feat[a];
    if (!owns_linear(m, lds.field[a])) return 0.0f;
    float lw;
    // refreshed >= 2: the features that occur once in the block are refreshed here, by their row
    if (TRAIN && (refreshed == 0 || (refreshed >= 2 && lds.opos[a] == OCC_ONCE))) {
      lw = ftrl_weight(m.h, m.lin_n[i], m.lin_z[i]);
      m.lin_w[i] = lw;
    } else {
      lw = m.lin_w[i];
    }
    return lw;
  };
  float lw_early = 0.0f;
  if (static_cast<int>(threadIdx.x) < nv) lw_early = linear_weight(threadIdx.x);
  if (is_ffm) {
    for (int a = threadIdx.x; a < nv; a += blockDim.x) {
      atomicAdd(&lds.fcnt[lds.field[a]], 1);
      atomicMin(reinterpret_cast<unsigned *>(&lds.ffirst[lds.field[a]]), static_cast<unsigned>(a));
    }
    __syncthreads();
  }

  if (TRAIN && is_ffm) {
    publish_row_tables(m, rows, s, lds, r, b, nv, F);
    // lazy refresh of every slot (feature a, partner field fp) that a pair of this row touches:
    // FFM::update_vector_w, ffm.cpp:72-88 -- unless ffm_refresh_kernel did it for the whole block
    if (refreshed == 1) {
    } else if (VEC4) {
      // the entries refreshed here (all of them, or with refreshed == 2 those whose feature occurs
      // nowhere else in the block), compacted by wave 0; then kRefreshFly 16-byte vectors of (n, z)
      // per thread in flight at a time
      if (threadIdx.x < 64) {
        int ns = 0;
        for (int base = 0; base < nv; base += 64) {
          const int a = base + threadIdx.x;
          const bool mine = a < nv && (refreshed == 0 || lds.opos[a] == OCC_ONCE);
          const unsigned long long mask = __ballot(mine);
          const int at = ns + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
          if (mine) lds.slist[at] = a;
          if (a < nv) lds.sidx[a] = mine ? at : -1;
          ns += __popcll(mask);
        }
        if (threadIdx.x == 0) s_ns = ns;
      }
      __syncthreads();
#ifndef FFM_REFRESH_FLY
#define FFM_REFRESH_FLY 4
#endif
      // (a shard's instantiation: two, its records are short -- 49 VGPRs instead of 77, a wave more per SIMD)
      constexpr int kRefreshFly = WHOLE ? FFM_REFRESH_FLY : 2;
      const int RL4 = RL >> 2, k4 = k >> 2;
      const int per = record_span(m, k4);  // vectors walked per record
      const float inv_per = 1.0f / static_cast<float>(per), inv_k4 = 1.0f / static_cast<float>(k4);
      const int total = s_ns * per;
      for (int t0 = threadIdx.x; t0 < total; t0 += kRefreshFly * blockDim.x) {
        float4 *wp[kRefreshFly];
        float4 n4[kRefreshFly], z4[kRefreshFly], w4[kRefreshFly];
#pragma unroll
        for (int u = 0; u < kRefreshFly; u++) {
          const int t = t0 + u * blockDim.x;
          wp[u] = nullptr;
          if (t >= total) continue;
          int j = static_cast<int>((t + 0.5f) * inv_per);
          j += (j + 1) * per <= t ? 1 : (j * per > t ? -1 : 0);  // exact for any size
          const int a = lds.slist[j];
          const int c4 = t - j * per;
          int sl = static_cast<int>((c4 + 0.5f) * inv_k4);  // slot of the record
          sl += (sl + 1) * k4 <= c4 ? 1 : (sl * k4 > c4 ? -1 : 0);
          const int fa = lds.field[a];
          const int fp = walk_field(m, fa, sl);
          const bool touched = fp >= 0 && (lds.fcnt[fp] - (fa == fp ? 1 : 0)) > 0 && owns_pair(m, fa, fp);
          if (touched) {
            float4 *row = reinterpret_cast<float4 *>(lat_row(m, lds.feat[a], fa));
            if (FFM_ROW_NT & 2) {
              n4[u] = load_nt(row + LAT_N * RL4 + c4);
              z4[u] = load_nt(row + LAT_Z * RL4 + c4);
            } else {
            n4[u] = row[LAT_N * RL4 + c4];
            z4[u] = row[LAT_Z * RL4 + c4];
            }
            w4[u] = m.h.learn ? row[LAT_W * RL4 + c4] : n4[u];
            wp[u] = row + LAT_W * RL4 + c4;
          }
        }
#pragma unroll
        for (int u = 0; u < kRefreshFly; u++)
          if (wp[u]) {
            const float4 wn = latent_weight4(m.h, n4[u], z4[u], w4[u]);
            if (!(FFM_ROW_EXP & 32) || wn.x == 123.456f) {
              if (FFM_ROW_NT & 16) store_nt(wp[u], wn);
              else *wp[u] = wn;
            }
            const int t = t0 + u * blockDim.x;
            if (WHOLE && t < park_vecs) park_w[t] = wn;
          }
      }
    } else {
      const int total = nv * RL;
      for (int t = threadIdx.x; t < total; t += blockDim.x) {
        const int a = t / RL, e = t - a * RL;
        if (refreshed >= 2 && lds.opos[a] != OCC_ONCE) continue;
        const int fa = lds.field[a];
        const int fp = walk_field(m, fa, e / k);
        const bool touched = fp >= 0 && (lds.fcnt[fp] - (fa == fp ? 1 : 0)) > 0 && owns_pair(m, fa, fp);
        if (touched) {
          float *row = lat_row(m, lds.feat[a], fa);
          row[LAT_W * RL + e] = latent_weight(m.h, row[LAT_N * RL + e], row[LAT_Z * RL + e],
                                             m.h.learn ? row[LAT_W * RL + e] : 0.0f);
        }
      }
    }
  }

  if (static_cast<int>(threadIdx.x) < nv) lds.linw[threadIdx.x] = lw_early;
  for (int a = threadIdx.x + blockDim.x; a < nv; a += blockDim.x) lds.linw[a] = linear_weight(a);
  __syncthreads();  // this row's refreshed weights are now readable by the whole workgroup

  // linear logit, sequential in row order (compute_linear_logit, ftrl_model.cpp:44-50)
  float result = 0.0f;
  if (threadIdx.x == 0) {
    if (m.bias_own) {
      float bias;
      if (TRAIN) {
        bias = ftrl_weight(m.h, m.bias3[1], m.bias3[2]);  // update_bias, ftrl_model.cpp:61-64
        if (r == 0) m.bias3[0] = bias;
      } else {
        bias = m.bias3[0];
      }
      result = bias;
    }
    if (!m.lin_own) {
      for (int a = 0; a < nv; a++) result = result + lds.linw[a] * lds.val[a];
    } else {
      for (int a = 0; a < nv; a++)
        if (m.lin_own[lds.field[a]]) result = result + lds.linw[a] * lds.val[a];
    }
  }

  // A shard owns 1/n_shards of the field pairs: walk (entry a, owned partner field of a's field)
  // instead of testing all nv(nv-1)/2 pairs -- when no field holds two entries in this row.  The
  // cross-shard sum reorders the pairs anyway, so the shard's own terms need no particular order,
  // only a FIXED one: every thread adds the terms of its items (t, t + blockDim.x, ...), the waves
  // reduce by a butterfly, wave after wave -- deterministic, and a dozen instructions where the
  // strictly ordered sum of 200 mostly empty terms cost 400 (a fifth of a shard row's instructions).
  bool shard_walk = false;
  if (is_ffm && m.n_shards > 1 && nv > 1) {
    bool multi = false;
    for (int f = threadIdx.x; f < F; f += blockDim.x) multi = multi || lds.fcnt[f] > 1;
    shard_walk = !__syncthreads_or(multi);
  }
  if (shard_walk) {
    const int om = record_span(m, 1), items = nv * om;
    float part = 0.0f;
    for (int t = threadIdx.x; t < items; t += blockDim.x) {
      const int a = t / om, j = t - a * om;
      const int fa = lds.field[a];
      const int fb = walk_field(m, fa, j);
      if (fb >= 0 && owns_pair(m, fa, fb)) {
        const int bb = lds.ffirst[fb];
        if (lds.fcnt[fb] == 1 && bb > a) {
          const float *va = lat_row(m, lds.feat[a], fa) + LAT_W * RL + slot_of(m, fa, fb) * k;
          const float *vb = lat_row(m, lds.feat[bb], fb) + LAT_W * RL + slot_of(m, fb, fa) * k;
          float dot = 0.0f;
          if (VEC4) {
            const float4 *va4 = reinterpret_cast<const float4 *>(va);
            const float4 *vb4 = reinterpret_cast<const float4 *>(vb);
            for (int f4 = 0; f4 < (k >> 2); f4++) {
              const float4 x = va4[f4], y = vb4[f4];
              dot = dot + x.x * y.x;
              dot = dot + x.y * y.y;
              dot = dot + x.z * y.z;
              dot = dot + x.w * y.w;
            }
          } else {
            for (int f = 0; f < k; f++) dot = dot + va[f] * vb[f];
          }
          part = part + dot * lds.val[a] * lds.val[bb];
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part = part + __shfl_xor(part, off, 64);
    if (blockDim.x > 64) {  // several waves per row: their sums in wave order
      if ((threadIdx.x & 63) == 0) lds.terms[threadIdx.x >> 6] = part;
      __syncthreads();
      if (threadIdx.x == 0) {
        part = 0.0f;
        for (unsigned wv = 0; wv < blockDim.x >> 6; wv++) part = part + lds.terms[wv];
      }
    }
    if (threadIdx.x == 0) result = result + part;
  } else if (is_ffm && nv > 1) {
    const int n_pairs = nv * (nv - 1) / 2;
    for (int q0 = 0; q0 < n_pairs; q0 += terms_cap) {
      const int q1 = min(q0 + terms_cap, n_pairs);
      for (int q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
        int a, bb;
        unrank_pair(q, nv, a, bb);
        const int fa = lds.field[a], fb = lds.field[bb];
        float term = 0.0f;
        const bool own = owns_pair(m, fa, fb);
        if (own) {
          const float *va = lat_row(m, lds.feat[a], fa) + LAT_W * RL + slot_of(m, fa, fb) * k;
          const float *vb = lat_row(m, lds.feat[bb], fb) + LAT_W * RL + slot_of(m, fb, fa) * k;
          float dot = 0.0f;
          if (VEC4) {
            const float4 *va4 = reinterpret_cast<const float4 *>(va);
            const float4 *vb4 = reinterpret_cast<const float4 *>(vb);
            // a slot whose w this row's refresh parked in LDS is read there (ta / tb: its first vector)
            int ta = -1, tb = -1;
            if (TRAIN && WHOLE && park_vecs > 0) {
              const int k4 = k >> 2, per4 = record_span(m, k4);
              const int ja = lds.sidx[a], jb = lds.sidx[bb];
              const int sa = ja * per4 + slot_of(m, fa, fb) * k4, sb = jb * per4 + slot_of(m, fb, fa) * k4;
              ta = ja >= 0 && sa + k4 <= park_vecs ? sa : -1;
              tb = jb >= 0 && sb + k4 <= park_vecs ? sb : -1;
            }
            if (k == 16) {  // the common slot size: both slots' eight vectors in flight together
              float4 x[4], y[4];
              if (FFM_ROW_EXP & 2) {
#pragma unroll
                for (int f4 = 0; f4 < 4; f4++) x[f4] = y[f4] = make_float4(lds.val[a], lds.val[bb], 0.5f, 0.25f);
              } else {
                if (ta >= 0) {
#pragma unroll
                  for (int f4 = 0; f4 < 4; f4++) x[f4] = park_w[ta + f4];
                } else {
#pragma unroll
                  for (int f4 = 0; f4 < 4; f4++) x[f4] = va4[f4];
                }
                if (tb >= 0) {
#pragma unroll
                  for (int f4 = 0; f4 < 4; f4++) y[f4] = park_w[tb + f4];
                } else {
#pragma unroll
                  for (int f4 = 0; f4 < 4; f4++) y[f4] = vb4[f4];
                }
              }
#pragma unroll
              for (int f4 = 0; f4 < 4; f4++) {
                dot = dot + x[f4].x * y[f4].x;
                dot = dot + x[f4].y * y[f4].y;
                dot = dot + x[f4].z * y[f4].z;
                dot = dot + x[f4].w * y[f4].w;
              }
            } else {
              for (int f4 = 0; f4 < (k >> 2); f4++) {
                float4 x, y;
                if (ta >= 0) x = park_w[ta + f4]; else x = va4[f4];
                if (tb >= 0) y = park_w[tb + f4]; else y = vb4[f4];
                dot = dot + x.x * y.x;
                dot = dot + x.y * y.y;
                dot = dot + x.z * y.z;
                dot = dot + x.w * y.w;
              }
            }
          } else {
            for (int f = 0; f < k; f++) dot = dot + va[f] * vb[f];
          }
          term = dot * lds.val[a] * lds.val[bb];
        }
        // pairs another shard owns contribute nothing here; tag them so the sum skips them
        lds.terms[q - q0] = own ? term : __int_as_float(0x7fc00001);
      }
      __syncthreads();
      if (threadIdx.x < 64) result = wave_add_terms_in_order(lds.terms, q1 - q0, result);
      __syncthreads();
    }
  }

  if (threadIdx.x == 0) {
    if (TRAIN) {
      s.logit[r] = result;
      if (WHOLE && own_tg) {
        // the whole logit is here (one shard): tmp_grad = sigmoid(logit) - y (ffm.cpp:44) and the
        // row's logloss (ftrl_offline.cpp:80) without a pass of their own
        const int y = rows.label[r];
        const float tg = sigmoid_ref_tab(result, s_tab) - static_cast<float>(y);
        s.tg[r] = tg;
        s.loss[r] = logloss_ref(y, result);
        if (out) out[r] = result;
        s_tg = tg;
      }
    } else {
      out[r] = output_prob ? sigmoid_ref(result) : result;
      if (rows.label) s.loss[r] = logloss_ref(rows.label[r], result);
    }
  }
  if (TRAIN && WHOLE && own_tg && is_ffm) {
    __syncthreads();
    const float tg = s_tg;
    // refreshed == 3: the (n, z) update of the features that occur nowhere else in the block
    // (FFM::update_vector_nz, ffm.cpp:90-136, for their slots), right here: this row is their only
    // touch, tmp_grad is known, and their records and the partners' weights were read moments ago
    // (refresh and pair phase above), so the second read comes out of the caches instead of HBM.
    // Same touches in the same order as ffm_update_single_kernel, which then has nothing to do.
    if (VEC4 && refreshed == 3 && !(FFM_ROW_EXP & 64)) {
#ifndef FFM_UPD_FLY
#define FFM_UPD_FLY 1
#endif

      constexpr int kUpdFly = FFM_UPD_FLY;
      const int RL4 = RL >> 2, k4 = k >> 2;
      const int per = record_span(m, k4);
      const float inv_per = 1.0f / static_cast<float>(per), inv_k4 = 1.0f / static_cast<float>(k4);
      const int total = s_ns * per;
      for (int t0 = threadIdx.x; t0 < total; t0 += kUpdFly * blockDim.x) {
        float4 *rp[kUpdFly];
        float4 n4[kUpdFly], z4[kUpdFly], w4[kUpdFly], vp[kUpdFly];
        int ia[kUpdFly], ifp[kUpdFly], ikq[kUpdFly];
#pragma unroll
        for (int u = 0; u < kUpdFly; u++) {
          const int t = t0 + u * blockDim.x;
          rp[u] = nullptr;
          if (t >= total) continue;
          int j = static_cast<int>((t + 0.5f) * inv_per);
          j += (j + 1) * per <= t ? 1 : (j * per > t ? -1 : 0);
          const int a = lds.slist[j];
          const int c4 = t - j * per;
          int sl = static_cast<int>((c4 + 0.5f) * inv_k4);
          sl += (sl + 1) * k4 <= c4 ? 1 : (sl * k4 > c4 ? -1 : 0);
          const int fa = lds.field[a];
          const int fp = walk_field(m, fa, sl);
          if (fp < 0 || (lds.fcnt[fp] - (fa == fp ? 1 : 0)) <= 0) continue;
          float4 *row = reinterpret_cast<float4 *>(lat_row(m, lds.feat[a], fa));
          if (FFM_ROW_EXP & 4) {
            n4[u] = z4[u] = make_float4(tg, 1.0f, 2.0f, 3.0f);
          } else if (FFM_ROW_NT & 4) {
            n4[u] = load_nt(row + LAT_N * RL4 + c4);
            z4[u] = load_nt(row + LAT_Z * RL4 + c4);
          } else {
          n4[u] = row[LAT_N * RL4 + c4];
          z4[u] = row[LAT_Z * RL4 + c4];
          }
          // (the refresh stored W(n, z) of these very (n, z) -- nobody else touches a once-only
          // feature's record inside the block -- so recomputing it gives the stored bits; the
          // learning variant's w depends on the old w and is read back)
          if (t < park_vecs) w4[u] = park_w[t];
          else if ((FFM_ROW_EXP & 1) && !m.h.learn) w4[u] = ftrl_weight4(m.h, n4[u], z4[u]);
          else w4[u] = (FFM_ROW_NT & 8) ? load_nt(row + LAT_W * RL4 + c4) : row[LAT_W * RL4 + c4];
          rp[u] = row + c4;
          ia[u] = a;
          ifp[u] = fp;
          ikq[u] = c4 - sl * k4;
          if (FFM_ROW_EXP & 8) vp[u] = make_float4(0.01f, 0.02f, tg, 0.03f);
          else if (lds.fcnt[fp] == 1) {  // (then fa != fp: the only entry of that field is the partner)
            const int bb = lds.ffirst[fp];
            const int jb = lds.sidx[bb];
            const int tp = jb * per + slot_of(m, fp, fa) * k4 + ikq[u];
            if (jb >= 0 && tp < park_vecs) vp[u] = park_w[tp];  // the partner's w is parked, too
            else vp[u] = reinterpret_cast<const float4 *>(lat_row(m, lds.feat[bb], fp))
                [LAT_W * RL4 + slot_of(m, fp, fa) * k4 + ikq[u]];
          }
        }
#pragma unroll
        for (int u = 0; u < kUpdFly; u++) {
          // (one vector's arithmetic after the other: interleaved, their temporaries cost a wave per SIMD)
          if (kUpdFly > 1) __builtin_amdgcn_sched_barrier(0);
          if (!rp[u]) continue;
          const int a = ia[u], fp = ifp[u], fa = lds.field[a];
          if (lds.fcnt[fp] == 1) {
            const int bb = lds.ffirst[fp];
            ffm_touch4(m.h, a < bb, tg, lds.val[a], lds.val[bb], vp[u], w4[u], n4[u], z4[u]);
          } else {  // several entries of that field in the row: one touch each, in row order
            for (int bb = lds.ffirst[fp]; bb < nv; bb++) {
              if (bb == a || lds.field[bb] != fp) continue;
              const float4 vq = reinterpret_cast<const float4 *>(lat_row(m, lds.feat[bb], fp))
                  [LAT_W * RL4 + slot_of(m, fp, fa) * k4 + ikq[u]];
              ffm_touch4(m.h, a < bb, tg, lds.val[a], lds.val[bb], vq, w4[u], n4[u], z4[u]);
            }
          }
          if (FFM_ROW_EXP & 16) {
            if (n4[u].x == 123.456f) rp[u][LAT_N * RL4] = z4[u];  // (keeps the arithmetic alive)
          } else if (FFM_ROW_NT & 1) {
            store_nt(rp[u] + LAT_N * RL4, n4[u]);
            store_nt(rp[u] + LAT_Z * RL4, z4[u]);
          } else {
          rp[u][LAT_N * RL4] = n4[u];
          rp[u][LAT_Z * RL4] = z4[u];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Lazy weight refresh of a whole block, once per DISTINCT feature instead of once per occurrence:
// with (n, z) frozen at the block start every occurrence of a feature computes the same w, so the
// block's refresh (FFM::update_vector_w ffm.cpp:72-88, update_linear_w ftrl_model.cpp:52-59) is
// one elementwise pass over the records of the block's distinct features -- the slots some row of
// the block touches (s.gmask, from the grouping) -- and the row kernel is left with the forward.
// Work item = (distinct feature u, 16-byte vector l of its record); VEC4 as in the row kernel.
// ------------------------------------------------------------------------------------------
// skip_single: the features that occur once in the block are refreshed by their row (the row
// kernel with refreshed == 2: their w then reaches the pair phase through L2 instead of a second
// trip to HBM); this pass walks the few / big / huge / giant lists only.
template <bool VEC4>
__global__ __launch_bounds__(256) void ffm_refresh_kernel(ModelDev m, Scratch s, int skip_single) {
  const int n_few = s.counters[CNT_NFEW], n_big = s.counters[CNT_NBIG];
  const int n_huge = s.counters[CNT_NHUGE], n_giant = s.counters[CNT_NGIANT];
  const int n_uniq = skip_single ? n_few + n_big + n_huge + n_giant : s.counters[CNT_NUNIQ];
  const int k = m.n_factors, RL = m.row_len;
  const int kv = VEC4 ? (k >> 2) : k;          // items per slot
  const int per = record_span(m, kv);          // items walked per record
  const float inv_kv = 1.0f / static_cast<float>(kv);
  const double inv_per = 1.0 / static_cast<double>(per);
  const unsigned total = static_cast<unsigned>(n_uniq) * static_cast<unsigned>(per);  // < 2^31 (engine)
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    // u = t / per through the double reciprocal (exact after one correction for t < 2^31)
    int u = static_cast<int>((static_cast<double>(t) + 0.5) * inv_per);
    u += static_cast<unsigned>(u + 1) * per <= t ? 1 : (static_cast<unsigned>(u) * per > t ? -1 : 0);
    const int lc = static_cast<int>(t - static_cast<unsigned>(u) * per);
    if (skip_single)
      u = u < n_few ? s.few[u]
          : u < n_few + n_big ? s.big[u - n_few]
          : u < n_few + n_big + n_huge ? s.huge[u - n_few - n_big] : s.giant[u - n_few - n_big - n_huge];
    const int4 ud = s.udesc[u];  // {feature, start, count, field}
    const int fa = ud.w;
    if (lc == 0 && owns_linear(m, fa)) {
      const int i0 = ud.x;
      m.lin_w[i0] = ftrl_weight(m.h, m.lin_n[i0], m.lin_z[i0]);
    }
    const int l = lc;  // walk position = element of the stored record
    int sl = static_cast<int>((l + 0.5f) * inv_kv);  // its slot
    sl += (sl + 1) * kv <= l ? 1 : (sl * kv > l ? -1 : 0);
    const int fp = walk_field(m, fa, sl);
    if (fp < 0) continue;
    const int i = ud.x;
    const unsigned long long mask = s.gmask[ud.y];
    if (!((mask >> fp) & 1ull)) continue;
    if (VEC4) {
      const int RL4 = RL >> 2;
      float4 *row = reinterpret_cast<float4 *>(lat_row(m, i, fa));
      const float4 n4 = row[LAT_N * RL4 + l], z4 = row[LAT_Z * RL4 + l];
      const float4 w4 = m.h.learn ? row[LAT_W * RL4 + l] : n4;
      row[LAT_W * RL4 + l] = latent_weight4(m.h, n4, z4, w4);
    } else {
      float *row = lat_row(m, i, fa);
      row[LAT_W * RL + l] = latent_weight(m.h, row[LAT_N * RL + l], row[LAT_Z * RL + l],
                                         m.h.learn ? row[LAT_W * RL + l] : 0.0f);
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
  RowLds lds = carve_row_lds(smem, max_row_nnz, 1, row_terms_cap(2, 0, m.n_factors));
  const int r = blockIdx.x;
  const int b = rows.row_ptr[r];
  const int nnz = rows.row_ptr[r + 1] - b;
  if ((TRAIN && s.counters[CNT_ERROR]) || nnz > max_row_nnz) {  // as in ffm_row_kernel
    if (threadIdx.x == 0) {
      if (nnz > max_row_nnz) atomicOr(s.err, ERR_ROW_TOO_LONG);
      const float nan = __int_as_float(0x7fc00000);
      if (TRAIN) { s.logit[r] = nan; s.tg[r] = 0.0f; }
      s.loss[r] = static_cast<double>(nan);
      if (out) out[r] = nan;
    }
    return;
  }
  const int k = m.n_factors;
  stage_row(m, rows, b, nnz, lds, &s_nv);
  __syncthreads();
  const int nv = s_nv;

  if (TRAIN) {  // FM::update_vector_w, fm.cpp:69-78
    const int total = nv * k;
    const float inv_k = 1.0f / static_cast<float>(k);
    for (int t = threadIdx.x; t < total; t += kRowThreads) {
      int a = static_cast<int>((t + 0.5f) * inv_k);
      a += (a + 1) * k <= t ? 1 : (a * k > t ? -1 : 0);
      const int e = t - a * k;
      float *row = lat_row(m, lds.feat[a], 0);
      row[LAT_W * k + e] = latent_weight(m.h, row[LAT_N * k + e], row[LAT_Z * k + e],
                                        m.h.learn ? row[LAT_W * k + e] : 0.0f);
    }
  }
  for (int a = threadIdx.x; a < nv; a += blockDim.x) {
    const int i = lds.feat[a];
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
      const float vx = lat_row(m, lds.feat[a], 0)[LAT_W * k + f] * lds.val[a];
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
    for (int a = 0; a < nv; a++) result = result + lds.linw[a] * lds.val[a];
    result = add_terms_in_order(lds.terms, k, result);
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

// Second half of predict() for logits that were summed outside (sharded engines): the value
// predict returns (ffm.cpp:51-55) and the row's logloss (eval/loss.h:8-12).
__global__ void predict_finish_kernel(int n_rows, const float *logit, const int *label,
                                      int output_prob, float *out, double *loss) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const float lg = logit[r];
  if (label) loss[r] = logloss_ref(label[r], lg);
  if (out) out[r] = output_prob ? sigmoid_ref(lg) : lg;
}

// Deterministic sum of the per-row losses (fixed order: 256 strided partials, then a tree).
// Several workgroups, each over a contiguous range of rows; the one that finishes last adds the
// partial sums in workgroup order -- a fixed order of additions whatever the timing.  (One
// workgroup took 110-150 us for the 65536 rows of an 8-GPU rank's block, at the end of the main
// stream.)  scratch: [kLossParts] partial sums, then the ticket counter.
constexpr int kLossParts = 64;
// (bidx of gdim: the workgroups of a launch that sum the losses)
__device__ __forceinline__ void loss_sum_body(int n_rows, const double *loss, double *out, double *scratch,
                                              unsigned bidx, unsigned gdim) {
  __shared__ double part[256];
  __shared__ bool last;
  const int per = (n_rows + gdim - 1) / gdim;
  const int r0 = bidx * per, r1 = min(n_rows, r0 + per);
  // (the first 256 threads of the workgroup sum; a larger workgroup's others only keep the barriers)
  if (threadIdx.x < 256) {
    double acc = 0.0;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) acc += loss[r];
    part[threadIdx.x] = acc;
  }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  unsigned *ticket = reinterpret_cast<unsigned *>(scratch + kLossParts);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&scratch[bidx], part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(ticket, 1u) == gdim - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    double total = 0.0;
    for (unsigned b = 0; b < gdim; b++)
      total += __hip_atomic_load(&scratch[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out = total;
    *ticket = 0u;
  }
}
__global__ __launch_bounds__(256) void loss_sum_kernel(int n_rows, const double *loss,
                                                       double *out, double *scratch) {
  loss_sum_body(n_rows, loss, out, scratch, blockIdx.x, gridDim.x);
}

// Evaluates sigmoid_ref on an array (self-test entry point: lets the parity tests compare the
// device's expf restatement with the host C library on millions of inputs).
__global__ void sigmoid_eval_kernel(int n, const float *x, float *y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = sigmoid_ref(x[i]);
}

}  // namespace ftrl_dev
