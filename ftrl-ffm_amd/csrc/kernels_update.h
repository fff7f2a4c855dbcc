// kernels_update.h -- feature-major half of the path: the FTRL (n,z) accumulator update.
// Replaces, for a whole block of rows at once,
//   FtrlModel::update_linear_nz / update_bias_nz   src/model/ftrl_model.cpp:66-85
//   FFM::update_vector_nz                          src/model/ffm.cpp:90-136  (incl. :118)
//   FM::update_vector_nz                           src/model/fm.cpp:80-101
//
// Every distinct feature of the block has exactly one owner per accumulator.  The owner folds the
// block's touches of the accumulator -- w and tmp_grad frozen at the block start -- by reductions
// in row order (kernels_fold.h has the tree, the checker under oracle/ restates it) and writes (n, z)
// back once.  No float atomics, no locks, bit-reproducible.  Accumulators that ONE row touches twice
// (a multi-valued field, a repeated id: s.cmask / UF_DUP) keep the row-order walk.
//
// Shapes of owner (the grouping pass deals the features into lists by occurrence count):
//  * "few" (2 .. kSmallMax occurrences, and the once-only ones of a shard): ONE wave owns the whole
//    record, each lane holding 4 consecutive factors of one slot (16-byte loads of n, z, w and of
//    each partner slot), all of a record's gathers in flight together -- bandwidth-shaped.
//  * "hot" features: one wave per 64 elements of the record, sixteen touches at a time, their
//    operands loaded touch-parallel and transposed through LDS (kernels_tile.h).
#pragma once
#include "engine_types.h"
#include "kernels_touch.h"
#include "kernels_fold.h"

namespace ftrl_dev {

constexpr int kUpdThreads = 256;
constexpr int kUpdWaves = kUpdThreads / 64;
#ifndef FFM_TILE_WAVES
#define FFM_TILE_WAVES 4  // waves per workgroup of the FFM update launches (kernels_tile.h); 8 and 16 measured slower
#endif
static_assert(FFM_TILE_WAVES >= 4, "loss_sum_body sums with the first 256 threads of an update workgroup");
constexpr int kUpdMaxThreads = 64 * (FFM_TILE_WAVES > 8 ? FFM_TILE_WAVES : 8);  // their workgroups are up to this large (eight waves: small blocks)
constexpr int kFmUnroll = 8;  // touches per prefetch group in the FM update kernel


// the 64-bit float offset an LDS fact record carries in two words (lo, hi)
__device__ __forceinline__ int64_t fact_offset(int lo, int hi) {
  return (static_cast<int64_t>(hi) << 32) | static_cast<unsigned>(lo);
}


// Linear update (update_linear_nz, ftrl_model.cpp:66-77): g = tmp_grad * x, every touch plain.
// Few-occurrence features: one thread each (a single segment).  Hot features: one wave each, lane =
// segment of kSeg occurrences -- every lane sums its segment left to right, the segment totals are
// joined in lane order (v_readlane: a strictly sequential chain of adds in uniform registers).
// A feature that repeats inside a row (UF_DUP) is walked touch by touch.
// skip_once: the linear terms of the features that occur once in the block were updated by their
// row (fm_row_wave_kernel)
__device__ __forceinline__ void linear_update_body(const ModelDev &m, const Rows &rows,
                                                   const Scratch &s, int block, int n_blocks,
                                                   int skip_once = 0) {
  const int n_small = s.counters[CNT_NSMALL], n_big = s.counters[CNT_NBIG];
  const int gtid = block * blockDim.x + threadIdx.x;
  for (int li = gtid; li < n_small; li += n_blocks * blockDim.x) {
    const int u = s.small[li];
    const int4 ud = s.udesc[u];  // {feature, start, count, field}
    if (!owns_linear(m, ud.w)) continue;  // another shard's linear terms
    const int i = ud.x;
    const int start = ud.y, c = ud.z;
    if (skip_once && c == 1) continue;
    float n = m.lin_n[i], z = m.lin_z[i];
    const float w = m.lin_w[i];
    if (s.uflag[start] & UF_DUP) {
      float sqn = sqrt_cr(n);
      for (int t = 0; t < c; t++) {
        const int2 pr = s.occ2[start + t];
        nz_step_linear_carry(m.h, w, s.tg[pr.y] * rows.val[pr.x], n, z, sqn);
      }
    } else {
      Fold a;
      a.init(n);
      for (int t = 0; t < c; t++) {  // c <= kSmallMax <= kSeg: one segment
        const int2 pr = s.occ2[start + t];
        a.plain(true, s.tg[pr.y] * rows.val[pr.x]);
      }
      fold_finish_linear(m.h, a, w, n, z);
    }
    m.lin_n[i] = n;
    m.lin_z[i] = z;
  }
  const int lane = threadIdx.x & 63;
  const int upd_waves = blockDim.x >> 6;  // (256-thread launches, or the FFM update launch's larger workgroups)
  const int wave = block * upd_waves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = n_blocks * upd_waves;
  const int n_huge = s.counters[CNT_NHUGE], n_giant = s.counters[CNT_NGIANT];
  for (int li = wave; li < n_big + n_huge + n_giant; li += n_waves) {
    // (the longest lists first)
    const int u = wave_uniform(li < n_giant ? s.giant[li]
                               : li < n_giant + n_huge ? s.huge[li - n_giant] : s.big[li - n_giant - n_huge]);
    const int4 ud = s.udesc[u];
    if (!owns_linear(m, wave_uniform(ud.w))) continue;
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    float n = m.lin_n[i], z = m.lin_z[i];
    const float w = m.lin_w[i];
    if (wave_uniform(s.uflag[start]) & UF_DUP) {
      if (lane == 0) {
        float sqn = sqrt_cr(n);
        for (int t = 0; t < c; t++) {
          const int2 pr = s.occ2[start + t];
          nz_step_linear_carry(m.h, w, s.tg[pr.y] * rows.val[pr.x], n, z, sqn);
        }
        m.lin_n[i] = n;
        m.lin_z[i] = z;
      }
      continue;
    }
    Fold a;  // (uniform across the wave: every lane joins the same totals)
    a.init(n);
    const int n_seg = (c + kSeg - 1) / kSeg;
    for (int s0 = 0; s0 < n_seg; s0 += 64) {
      const int t0 = (s0 + lane) * kSeg;
      float P = -0.0f, G = -0.0f;
      constexpr int kFly = 8;  // touches whose two dependent gathers are in flight together
      for (int j0 = 0; j0 < kSeg; j0 += kFly) {
        if (t0 + j0 >= c) break;
        int2 pr[kFly];
        float tgv[kFly], xv[kFly];
#pragma unroll
        for (int j = 0; j < kFly; j++) pr[j] = s.occ2[start + min(t0 + j0 + j, c - 1)];
#pragma unroll
        for (int j = 0; j < kFly; j++) { tgv[j] = s.tg[pr[j].y]; xv[j] = rows.val[pr[j].x]; }
#pragma unroll
        for (int j = 0; j < kFly; j++) {
          if (t0 + j0 + j < c) {
            const float g = tgv[j] * xv[j];
            G = G + g;
            P = P + g * g;
          }
        }
      }
      const int cnt = min(64, n_seg - s0);
      for (int l = 0; l < cnt; l++) {
        a.P = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P), l));
        a.G = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(G), l));
        a.flush();
      }
    }
    a.any = a.head_plain = true;
    fold_finish_linear(m.h, a, w, n, z);
    if (lane == 0) {
      m.lin_n[i] = n;
      m.lin_z[i] = z;
    }
  }
}
__global__ __launch_bounds__(kUpdThreads) void linear_update_kernel(ModelDev m, Rows rows,
                                                                    Scratch s, int skip_once) {
  linear_update_body(m, rows, s, blockIdx.x, gridDim.x, skip_once);
}

// Bias update: every row of the block touches the one accumulator once, g = tmp_grad
// (update_bias_nz, ftrl_model.cpp:79-85).  As a fold: thread = segment of kSeg rows (its 64 values
// are 256 contiguous bytes, sixteen 16-byte loads), summed left to right in the thread's registers;
// thread 0 then joins the segment totals in order.  For 8192 rows: 128 threads x 64 adds, then 128
// adds -- where the row-order walk was 2 x 8192 dependent adds (74 us, the floor of FM's update).
// Called by every thread of a workgroup (of up to kUpdMaxThreads threads).
// buf: 2 * blockDim.x floats of LDS the caller has no other use for (the FFM update launch hands its
// dynamic region over -- a static array here would be allocated by every workgroup of that launch and
// cost its tile ranges a resident workgroup per CU at k = 4)
__device__ __forceinline__ void bias_update_body(const ModelDev &m, int n_rows, const Scratch &sc, float *buf) {
  if (!m.bias_own) return;             // another shard's
  if (sc.counters[CNT_ERROR]) return;  // untrainable block (ERR_ROW_TOO_LONG): a no-op
  if (n_rows <= 0) return;
  float *s_P = buf, *s_G = buf + blockDim.x;
  const int n_thr = blockDim.x;
  const float *tg = sc.tg;
  const int n_seg = (n_rows + kSeg - 1) / kSeg;
  Fold a;
  a.init(m.bias3[1]);
  for (int s0 = 0; s0 < n_seg; s0 += n_thr) {
    const int seg = s0 + threadIdx.x;
    float P = -0.0f, G = -0.0f;
    if (seg < n_seg) {
      const int r0 = seg * kSeg;
      if (r0 + kSeg <= n_rows) {
        const float4 *t4 = reinterpret_cast<const float4 *>(tg + r0);
        float4 v[kSeg / 4];
#pragma unroll
        for (int j = 0; j < kSeg / 4; j++) v[j] = t4[j];
#pragma unroll
        for (int j = 0; j < kSeg / 4; j++) {
          G = G + v[j].x; P = P + v[j].x * v[j].x;
          G = G + v[j].y; P = P + v[j].y * v[j].y;
          G = G + v[j].z; P = P + v[j].z * v[j].z;
          G = G + v[j].w; P = P + v[j].w * v[j].w;
        }
      } else {
        for (int r = r0; r < n_rows; r++) {
          const float g = tg[r];
          G = G + g;
          P = P + g * g;
        }
      }
    }
    s_P[threadIdx.x] = P;
    s_G[threadIdx.x] = G;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int cnt = min(n_thr, n_seg - s0);
      for (int l = 0; l < cnt; l++) {
        a.P = s_P[l];
        a.G = s_G[l];
        a.flush();
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float n = m.bias3[1], z = m.bias3[2];
    a.any = a.head_plain = true;
    fold_finish_linear(m.h, a, m.bias3[0], n, z);
    m.bias3[1] = n;
    m.bias3[2] = z;
  }
}
__global__ __launch_bounds__(kUpdThreads) void bias_update_kernel(ModelDev m, int n_rows, Scratch s) {
  __shared__ float s_PG[2 * kUpdThreads];
  bias_update_body(m, n_rows, s, s_PG);
}

// The general owner: work item = (distinct feature, 64 elements of its record), lane = element,
// every input gathered in place through the row tables (no float4 path, no occurrence-ordered
// streams).  Two jobs:
//  * serial_only = 0 -- engines whose n_factors is not a multiple of 4: every accumulator of every
//    distinct feature; folded (kernels_fold.h) unless its slot is serial;
//  * serial_only = 1 -- beside the vector kernels: only the SERIAL slots (s.cmask: one row touches
//    them twice -- a multi-valued partner field, a repeated id; without field masks, n_fields > 64:
//    every slot) of the features with two or more occurrences, walked in row order on the running
//    (n, z) like the reference's one-thread loop (ffm.cpp:104-120).
// (one element e of the stored record of the feature described by ud = {feature, start, count, field};
// cm = its serial slots)
__device__ __forceinline__ void ffm_generic_element(const ModelDev &m, const Rows &rows, const Scratch &s,
                                                    int4 ud, unsigned long long cm, int e, int serial_only) {
  const int RL = m.row_len, k = m.n_factors, F = m.n_fields;
  if (e >= RL) return;
  const int start = ud.y, c = ud.z, fa = ud.w;
  const int sl = e / k, kk = e - sl * k;
  const int fp = walk_field(m, fa, sl);
  if (fp < 0) return;
  const bool serial = !s.cmask || ((cm >> fp) & 1ull);
  if (serial_only && !serial) return;
  const int i = ud.x;
  float *rec = lat_row(m, i, fa);
  float n = rec[LAT_N * RL + e], z = rec[LAT_Z * RL + e];
  const float w = rec[LAT_W * RL + e];
  bool touched = false;
  if (serial) {
    for (int t = 0; t < c; t++) {
      const int2 pr = s.occ2[start + t];
      const int p = pr.x, r = pr.y;
      const int fm = rows.field[p];
      if (!owns_pair(m, fm, fp)) continue;
      const float tg = s.tg[r], xm = rows.val[p];
      for (int qq = s.head[static_cast<int64_t>(r) * F + fp]; qq >= 0; qq = s.next[qq]) {
        if (qq == p) continue;
        const float vp = m.lat[w_slot_offset(m, rows.feat[qq], fp, fm) + kk];
        ffm_touch(m.h, p < qq, tg, xm, rows.val[qq], vp, w, n, z);
        touched = true;
      }
    }
  } else {
    Fold a;
    a.init(n);
    for (int t = 0; t < c; t++) {
      if (t > 0 && t % kSeg == 0) a.flush();
      const int2 pr = s.occ2[start + t];
      const int p = pr.x, r = pr.y;
      const int fm = rows.field[p];
      if (!owns_pair(m, fm, fp)) continue;
      const int4 rt = s.rowtab[static_cast<int64_t>(r) * F + fp];
      const int q = rt.z;  // the field's only entry in the row (the slot is not serial), or none
      if (q < 0 || q == p) continue;
      const bool live[1] = {true}, first[1] = {p < q || m.h.learn != 0};
      const float tgv[1] = {s.tg[r]}, xv[1] = {rows.val[p] * __int_as_float(rt.y)};
      const float vp[1] = {m.lat[w_slot_offset(m, rt.x, fp, fm) + kk]};
      fold_ffm_group<1>(a, w, live, first, tgv, xv, vp);
    }
    touched = fold_finish_latent(m.h, a, w, n, z);
  }
  if (touched) {
    rec[LAT_N * RL + e] = n;
    rec[LAT_Z * RL + e] = z;
  }
}
__device__ __forceinline__ void ffm_generic_body(const ModelDev &m, const Rows &rows, const Scratch &s,
                                                 int serial_only, unsigned bidx, unsigned gdim) {
  const unsigned chunks = (m.row_len + 63) / 64;
  const int lane = threadIdx.x & 63;
  const unsigned upd_waves = blockDim.x >> 6;
  const unsigned wave = bidx * upd_waves + wave_uniform(threadIdx.x >> 6);
  const unsigned n_waves = gdim * upd_waves;
  if (serial_only) {
    // one lane per feature of the few / big / huge / giant lists looks at its serial slots; the wave
    // then walks the (rare) features that have any, one after another
    const int n_few = s.counters[CNT_NFEW], n_big = n_few + s.counters[CNT_NBIG];
    const int n_huge = n_big + s.counters[CNT_NHUGE], n_multi = n_huge + s.counters[CNT_NGIANT];
    for (int base = wave * 64; base < n_multi; base += n_waves * 64) {
      const int li = base + lane;
      int4 ud = make_int4(0, 0, 0, 0);
      unsigned long long cm = 0ull;
      if (li < n_multi) {
        const int u = li < n_few ? s.few[li] : li < n_big ? s.big[li - n_few]
                      : li < n_huge ? s.huge[li - n_big] : s.giant[li - n_huge];
        ud = s.udesc[u];
        cm = s.cmask ? s.cmask[ud.y] : ~0ull;
      }
      unsigned long long todo = __ballot(cm != 0ull);
      while (todo) {
        const int l = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1ull;
        const int4 d = make_int4(__builtin_amdgcn_readlane(ud.x, l), __builtin_amdgcn_readlane(ud.y, l),
                                 __builtin_amdgcn_readlane(ud.z, l), __builtin_amdgcn_readlane(ud.w, l));
        const unsigned long long cml =
            (static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(cm >> 32), l))) << 32) |
            static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(cm & 0xffffffffull), l));
        for (unsigned ci = 0; ci < chunks; ci++)
          ffm_generic_element(m, rows, s, d, cml, static_cast<int>(ci) * 64 + lane, 1);
      }
    }
    return;
  }
  const unsigned n_items = static_cast<unsigned>(s.counters[CNT_NUNIQ]) * chunks;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned u = item / chunks;
    const int4 ud = s.udesc[u];
    const unsigned long long cm = s.cmask ? s.cmask[ud.y] : ~0ull;
    ffm_generic_element(m, rows, s, ud, cm, static_cast<int>(item - u * chunks) * 64 + lane, 0);
  }
}
__global__ __launch_bounds__(kUpdThreads) void ffm_update_generic_kernel(ModelDev m, Rows rows,
                                                                         Scratch s, int serial_only) {
  ffm_generic_body(m, rows, s, serial_only, blockIdx.x, gridDim.x);
}

// ---- few-occurrence features: one wave per feature, lanes = 4 consecutive factors of a slot ------
// Requires n_factors % 4 == 0.  c <= kSmallMax occurrences: a single segment.  Kept lean in
// registers: the bandwidth comes from many resident waves, each with its record's loads in flight.
// few_only: the features that occur once are ffm_update_single_kernel's (or their row's).
// (The owners that stream a record's (n, z) through once per block -- the few-occurrence range, and on a
// shard the flat few-occurrence kernel and the once-only kernels -- read and write them with the non-temporal
// hint: they do not push the partners' gathered weights out of the L2.  C5 step -1.4 %, an emulated 8-GPU
// rank 1.69 -> 1.66 ms; profiles/r06_experiments.md section 19.)
#ifndef FFM_SMALL_BATCH
#define FFM_SMALL_BATCH 4
#endif
constexpr int kSmallBatch = FFM_SMALL_BATCH < kSmallMax ? FFM_SMALL_BATCH : kSmallMax;  // touches whose gathers fly together
#ifdef FFM_SMALL_WAVES
#define FFM_SMALL_OCC __attribute__((amdgpu_waves_per_eu(FFM_SMALL_WAVES, FFM_SMALL_WAVES)))
#else
#define FFM_SMALL_OCC
#endif
#ifndef FFM_FEW_LDS
#define FFM_FEW_LDS 1
#endif
#ifndef FFM_FOLD_REGULAR
#define FFM_FOLD_REGULAR 1
#endif
__device__ __forceinline__ void ffm_small_body(const ModelDev &m, const Rows &rows, const Scratch &s,
                                               int few_only, unsigned bidx, unsigned gdim) {
#if FFM_FEW_LDS
  __shared__ int4 few_facts[kUpdMaxThreads / 64][kSmallMax];  // per wave: {entry, row, own field, own value}
  __shared__ float few_tg[kUpdMaxThreads / 64][kSmallMax];    // ... and the row's tmp_grad
  const int wv_in_block = wave_uniform(threadIdx.x >> 6);
#endif
  const int RL = m.row_len, k = m.n_factors, F = m.n_fields;
  const int RL4 = RL >> 2, k4 = k >> 2;
  const int lane = threadIdx.x & 63;
  const int upd_waves = blockDim.x >> 6;
  const int wave = bidx * upd_waves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = gdim * upd_waves;
  const int *list = few_only ? s.few : s.small;
  const int n_small = s.counters[few_only ? CNT_NFEW : CNT_NSMALL];
  const float inv_k4 = 1.0f / static_cast<float>(k4);
  const int span4 = record_span(m, k4);
#if FFM_FOLD_REGULAR
  const bool regular = m.sort_start != nullptr && m.n_shards == 1 && m.own_n == nullptr && s.counters[CNT_IRREGULAR] == 0;
#else
  const bool regular = false;
#endif
  // (a feature is a chain of dependent loads -- list, descriptor, touches, their entries and rows,
  // row table, partner weights: the list entry is requested two features ahead, the descriptor one)
  auto uniform4 = [](int4 v) { return make_int4(wave_uniform(v.x), wave_uniform(v.y), wave_uniform(v.z), wave_uniform(v.w)); };
  int4 ud_next = make_int4(0, 0, 0, 0);
  int u_next2 = 0;
  if (wave < n_small) ud_next = uniform4(s.udesc[wave_uniform(list[wave])]);
  if (wave + n_waves < n_small) u_next2 = wave_uniform(list[wave + n_waves]);
  for (int li = wave; li < n_small; li += n_waves) {
    const int4 ud = ud_next;
    if (li + n_waves < n_small) ud_next = uniform4(s.udesc[u_next2]);
    if (li + 2 * n_waves < n_small) u_next2 = wave_uniform(list[li + 2 * n_waves]);
    const int i = ud.x;
    const int start = ud.y, c = ud.z;
    const int fa = ud.w;
    float4 *rec4 = reinterpret_cast<float4 *>(lat_row(m, i, fa));
    // slots that one row touches twice keep the row-order walk (ffm_generic_body, serial_only)
    const unsigned long long cm = s.cmask[start];
    // The feature's (at most kSmallMax) touches, read ONCE for all passes over its record: entry,
    // row, own field, own value, tmp_grad.  Lane j fetches touch j (the touches' loads fly together)
    // and parks it in the wave's LDS record; the passes read the records back as they need them.
    // (Rounds 4-5 kept these 40 wave-uniform values in scalar registers: 188 of them spilled into
    // VGPR lanes, 36 bytes per lane into scratch memory -- VERDICT r05.)
#if FFM_FEW_LDS
    {
      if (lane < kSmallMax) {
        int4 f = make_int4(0, 0, 0, 0);
        float g = 0.0f;
        if (lane < c) {
          const int2 pr = s.occ2[start + lane];
          f = make_int4(pr.x, pr.y, rows.field[pr.x], __float_as_int(rows.val[pr.x]));
          g = s.tg[pr.y];
        }
        few_facts[wv_in_block][lane] = f;
        few_tg[wv_in_block][lane] = g;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
#define FEW_TP(j) (few_facts[wv_in_block][j].x)
#define FEW_TR(j) (few_facts[wv_in_block][j].y)
#define FEW_TFM(j) (few_facts[wv_in_block][j].z)
#define FEW_TXM(j) (__int_as_float(few_facts[wv_in_block][j].w))
#define FEW_TTG(j) (few_tg[wv_in_block][j])
#else
    int tp[kSmallMax], tr[kSmallMax], tfm[kSmallMax];
    float txm[kSmallMax], ttg[kSmallMax];
#pragma unroll
    for (int j = 0; j < kSmallMax; j++) {
      tp[j] = tr[j] = tfm[j] = 0;
      txm[j] = ttg[j] = 0.0f;
      if (j < c) {
        const int2 pr = s.occ2[start + j];
        tp[j] = wave_uniform(pr.x);
        tr[j] = wave_uniform(pr.y);
        tfm[j] = wave_uniform(rows.field[tp[j]]);
        txm[j] = __int_as_float(wave_uniform(__float_as_int(rows.val[tp[j]])));
        ttg[j] = __int_as_float(wave_uniform(__float_as_int(s.tg[tr[j]])));
      }
    }
#define FEW_TP(j) (tp[j])
#define FEW_TR(j) (tr[j])
#define FEW_TFM(j) (tfm[j])
#define FEW_TXM(j) (txm[j])
#define FEW_TTG(j) (ttg[j])
#endif
    for (int l0 = 0; l0 < span4; l0 += 64) {
      const int l = l0 + lane;  // 16-byte vector of the stored record
      const int lc = l < span4 ? l : 0;
      int sl = static_cast<int>((lc + 0.5f) * inv_k4);  // its slot
      sl += (sl + 1) * k4 <= lc ? 1 : (sl * k4 > lc ? -1 : 0);
      int fp = l < span4 ? walk_field(m, fa, sl) : -1;  // partner field of this lane's slot
      if (fp >= 0 && ((cm >> fp) & 1ull)) fp = -1;      // serial
      const bool mine = fp >= 0;
      fp = mine ? fp : 0;
      const int kq = lc - sl * k4;  // which 16-byte quarter of the slot
      const unsigned long long own_bits = owner_bits(m, fp);
      float n[4] = {0.0f, 0.0f, 0.0f, 0.0f}, z[4] = {0.0f, 0.0f, 0.0f, 0.0f}, w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (mine) {
        // (the record's (n, z) pass through once per block: the non-temporal hint keeps them from pushing the
        // partners' gathered weights out of the L2 -- few launch 350 -> 340 us, C5 step -1.4 %; the same hint on
        // the hot and giant ranges' records, or on w, bought nothing more)
        const float4 n4 = load_nt(rec4 + LAT_N * RL4 + lc), z4 = load_nt(rec4 + LAT_Z * RL4 + lc), w4 = rec4[LAT_W * RL4 + lc];
        n[0] = n4.x; n[1] = n4.y; n[2] = n4.z; n[3] = n4.w;
        z[0] = z4.x; z[1] = z4.y; z[2] = z4.z; z[3] = z4.w;
        w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
      }
      FoldFew<4> a;
      a.init();
      const bool lv = mine && fp != fa, q118 = lv && !m.h.learn && fp < fa;
      // kSmallBatch touches at a time: their row-table entries in flight together, then their
      // partners' weights, then the touches in order
#pragma unroll
      for (int j0 = 0; j0 < kSmallMax; j0 += kSmallBatch) {
        if (j0 >= c) break;
        int4 rt[kSmallBatch];
#pragma unroll
        for (int jj = 0; jj < kSmallBatch; jj++) {
          const int j = j0 + jj;
          rt[jj] = make_int4(0, 0, -1, 0);  // "no entry of that field in the row"
          if (mine && j < c && owns_bit(own_bits, FEW_TFM(j))) rt[jj] = s.rowtab[static_cast<int64_t>(FEW_TR(j)) * F + fp];
        }
        float4 vp4[kSmallBatch];
#pragma unroll
        for (int jj = 0; jj < kSmallBatch; jj++) {
          const int j = j0 + jj;
          vp4[jj] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (j < c && rt[jj].z >= 0 && rt[jj].z != FEW_TP(j))
            vp4[jj] = reinterpret_cast<const float4 *>(lat_row(m, rt[jj].x, fp))[LAT_W * RL4 + slot_of(m, fp, FEW_TFM(j)) * k4 + kq];
        }
#pragma unroll
        for (int jj = 0; jj < kSmallBatch; jj++) {
          const int j = j0 + jj;
          if (j >= c) continue;
          const int p = FEW_TP(j), q = rt[jj].z;
          const float vp[4] = {vp4[jj].x, vp4[jj].y, vp4[jj].z, vp4[jj].w};
          if (regular) {  // (every row one entry per field in field order: the flags are the lane's)
            a.touch_regular(n, w, lv, q118, FEW_TTG(j), FEW_TXM(j) * __int_as_float(rt[jj].y), vp);
            continue;
          }
          const bool live = q >= 0 && q != p;  // (q == -2, several entries, only on serial slots)
          a.touch(n, w, live, p < q || m.h.learn != 0, FEW_TTG(j), FEW_TXM(j) * __int_as_float(rt[jj].y), vp);
        }
      }
      if (a.finish(m.h, w, n, z) && mine) {
        store_nt(rec4 + LAT_N * RL4 + lc, make_float4(n[0], n[1], n[2], n[3]));
        store_nt(rec4 + LAT_Z * RL4 + lc, make_float4(z[0], z[1], z[2], z[3]));
      }
    }
#if FFM_FEW_LDS
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next feature overwrites the records)
#endif
  }
#undef FEW_TP
#undef FEW_TR
#undef FEW_TFM
#undef FEW_TXM
#undef FEW_TTG
}
__global__ __launch_bounds__(kUpdThreads) FFM_SMALL_OCC void ffm_update_small_kernel(ModelDev m, Rows rows,
                                                                       Scratch s, int few_only) {
  ffm_small_body(m, rows, s, few_only, blockIdx.x, gridDim.x);
}

// The same for SHORT stored records (a compact shard's), a lane = one (feature, vector) of a flat
// index space -- see ffm_update_single_flat_kernel below.
#ifndef FFM_FLAT_BATCH
#define FFM_FLAT_BATCH 4
#endif
constexpr int kFlatBatch = FFM_FLAT_BATCH < kSmallMax ? FFM_FLAT_BATCH : kSmallMax;
__global__ __launch_bounds__(kUpdThreads) void ffm_update_small_flat_kernel(ModelDev m, Rows rows,
                                                                            Scratch s, int few_only) {
  const int RL4 = m.row_len >> 2, k4 = m.n_factors >> 2, F = m.n_fields;
  const int span4 = record_span(m, k4);
  const int *list = few_only ? s.few : s.small;
  const unsigned total =
      static_cast<unsigned>(s.counters[few_only ? CNT_NFEW : CNT_NSMALL]) * static_cast<unsigned>(span4);
  const double inv_span = 1.0 / static_cast<double>(span4);
  const float inv_k4 = 1.0f / static_cast<float>(k4);
  const unsigned stride = gridDim.x * blockDim.x;
  const unsigned rounds = (total + stride - 1) / stride;  // (whole waves stay together: the fold votes)
  for (unsigned rd = 0; rd < rounds; rd++) {
    const unsigned t = rd * stride + blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = t < total;
    const unsigned tt = in ? t : 0u;
    int li = static_cast<int>((static_cast<double>(tt) + 0.5) * inv_span);  // t / span4, corrected
    li += static_cast<unsigned>(li + 1) * span4 <= tt ? 1 : (static_cast<unsigned>(li) * span4 > tt ? -1 : 0);
    const int l = static_cast<int>(tt - static_cast<unsigned>(li) * span4);  // vector of the stored record
    int sl = static_cast<int>((l + 0.5f) * inv_k4);                          // its slot
    sl += (sl + 1) * k4 <= l ? 1 : (sl * k4 > l ? -1 : 0);
    const int kq = l - sl * k4;
    int4 ud = make_int4(0, 0, 0, 0);
    if (in) ud = s.udesc[list[li]];  // {feature, start, count, field}
    const int i = ud.x, start = ud.y, c = ud.z, fa = ud.w;
    int fp = in ? walk_field(m, fa, sl) : -1;  // partner field of this vector's slot
    if (fp >= 0 && ((s.cmask[start] >> fp) & 1ull)) fp = -1;  // serial: ffm_generic_body's
    const bool mine = fp >= 0;
    fp = mine ? fp : 0;
    const unsigned long long own_bits = owner_bits(m, fp);
    float4 *rec4 = reinterpret_cast<float4 *>(lat_row(m, i, fa));
    float n[4] = {0.0f, 0.0f, 0.0f, 0.0f}, z[4] = {0.0f, 0.0f, 0.0f, 0.0f}, w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (mine) {
      const float4 n4 = load_nt(rec4 + LAT_N * RL4 + l), z4 = load_nt(rec4 + LAT_Z * RL4 + l), w4 = rec4[LAT_W * RL4 + l];
      n[0] = n4.x; n[1] = n4.y; n[2] = n4.z; n[3] = n4.w;
      z[0] = z4.x; z[1] = z4.y; z[2] = z4.z; z[3] = z4.w;
      w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
    }
    FoldFew<4> a;
    a.init();
    // kFlatBatch touches at a time: their {entry, row}, then their fields / row-table entries / values /
    // tmp_grad, then the partners' weights -- three dependent trips per BATCH, not per touch (one touch at a
    // time the kernel was a chain of 3 c trips per item: 428 us per 65 536-row block of an 8-GPU rank,
    // the end of the rank's step) -- and then the touches in order
    for (int j0 = 0; j0 < kSmallMax; j0 += kFlatBatch) {
      if (!__any(mine && j0 < c)) break;
      int2 pr[kFlatBatch];
#pragma unroll
      for (int jj = 0; jj < kFlatBatch; jj++) {
        pr[jj] = make_int2(0, 0);
        if (mine && j0 + jj < c) pr[jj] = s.occ2[start + j0 + jj];
      }
      int fm[kFlatBatch];
      int4 rt[kFlatBatch];
      float tgv[kFlatBatch], xv[kFlatBatch];
      bool own[kFlatBatch];
#pragma unroll
      for (int jj = 0; jj < kFlatBatch; jj++) {
        fm[jj] = 0;
        own[jj] = false;
        rt[jj] = make_int4(0, 0, -1, 0);
        tgv[jj] = xv[jj] = 0.0f;
        if (mine && j0 + jj < c) {
          fm[jj] = rows.field[pr[jj].x];
          rt[jj] = s.rowtab[static_cast<int64_t>(pr[jj].y) * F + fp];
          tgv[jj] = s.tg[pr[jj].y];
          xv[jj] = rows.val[pr[jj].x];
        }
      }
      float4 v4[kFlatBatch];
      bool live[kFlatBatch];
#pragma unroll
      for (int jj = 0; jj < kFlatBatch; jj++) {
        own[jj] = mine && j0 + jj < c && owns_bit(own_bits, fm[jj]);
        live[jj] = own[jj] && rt[jj].z >= 0 && rt[jj].z != pr[jj].x;
        v4[jj] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (live[jj])
          v4[jj] = reinterpret_cast<const float4 *>(lat_row(m, rt[jj].x, fp))[LAT_W * RL4 + slot_of(m, fp, fm[jj]) * k4 + kq];
      }
#pragma unroll
      for (int jj = 0; jj < kFlatBatch; jj++) {
        if (!__any(mine && j0 + jj < c)) break;
        const float vp[4] = {v4[jj].x, v4[jj].y, v4[jj].z, v4[jj].w};
        const bool first = live[jj] && (pr[jj].x < rt[jj].z || m.h.learn != 0);
        a.touch(n, w, live[jj], first, live[jj] ? tgv[jj] : 0.0f, live[jj] ? xv[jj] * __int_as_float(rt[jj].y) : 0.0f, vp);
      }
    }
    if (a.finish(m.h, w, n, z) && mine) {
      store_nt(rec4 + LAT_N * RL4 + l, make_float4(n[0], n[1], n[2], n[3]));
      store_nt(rec4 + LAT_Z * RL4 + l, make_float4(z[0], z[1], z[2], z[3]));
    }
  }
}

// ---- features that occur ONCE in the block (most of the distinct features) ----------------------
// (On one shard the row kernel applies these touches itself -- kernels_row.h, refreshed == 3 --
// and this kernel is not launched; a shard of several only has tmp_grad after the all-reduce.)
// One wave per feature, driven by a 16-byte descriptor {feature, entry, row, field} the grouping
// wrote for it, so the dependent-load chain is three deep instead of six (descriptor -> record
// vectors + the row's per-field table -> partner weights): each lane takes up to kSingleTrips
// 16-byte vectors of the record and has all their loads in flight together.
// (kSingleTrips: 3 covers a full 39x16 record in one pass; a compact shard's record is a quarter
// of that, and the vectors it would never use cost registers -- 125 against 73 -- and wave slots:
// the engine picks the instantiation by the stored record's length)
template <int kSingleTrips>
__global__ __launch_bounds__(kUpdThreads) void ffm_update_single_kernel(ModelDev m, Rows rows,
                                                                        Scratch s) {
  const int RL = m.row_len, k = m.n_factors, F = m.n_fields;
  const int RL4 = RL >> 2, k4 = k >> 2;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = gridDim.x * kUpdWaves;
  const int n_single = s.counters[CNT_NSINGLE];
  const float inv_k4 = 1.0f / static_cast<float>(k4);
  const int span4 = record_span(m, k4);
  for (int li = wave; li < n_single; li += n_waves) {
    const int4 d = s.sdesc[li];
    const int i = wave_uniform(d.x), p = wave_uniform(d.y), r = wave_uniform(d.z);
    const int fa = wave_uniform(d.w);
    const float xm = rows.val[p], tg = s.tg[r];
    float4 *rec4 = reinterpret_cast<float4 *>(lat_row(m, i, fa));
    const unsigned long long own_fa = owner_bits(m, fa);
    const int4 *rtab = s.rowtab + static_cast<int64_t>(r) * F;
    for (int l0 = 0; l0 < span4; l0 += 64 * kSingleTrips) {
      int l[kSingleTrips], fp[kSingleTrips], kq[kSingleTrips];
      float4 n4[kSingleTrips], z4[kSingleTrips], w4[kSingleTrips], vp[kSingleTrips];
      int4 rt[kSingleTrips];
#pragma unroll
      for (int t = 0; t < kSingleTrips; t++) {
        const int lc = l0 + t * 64 + lane;  // 16-byte vector of the stored record
        const int ll = lc < span4 ? lc : 0;
        int sl = static_cast<int>((ll + 0.5f) * inv_k4);  // its slot
        sl += (sl + 1) * k4 <= ll ? 1 : (sl * k4 > ll ? -1 : 0);
        int f = lc < span4 ? walk_field(m, fa, sl) : -1;  // partner field of this vector's slot
        if (f >= 0 && !owns_bit(own_fa, f)) f = -1;
        l[t] = f >= 0 ? ll : -1;
        kq[t] = ll - sl * k4;
        f = f < 0 ? 0 : f;
        fp[t] = f;
        if (l[t] >= 0) {
          n4[t] = load_nt(rec4 + LAT_N * RL4 + ll);
          z4[t] = load_nt(rec4 + LAT_Z * RL4 + ll);
          w4[t] = rec4[LAT_W * RL4 + ll];
          rt[t] = rtab[f];
        }
      }
#pragma unroll
      for (int t = 0; t < kSingleTrips; t++) {
        if (l[t] >= 0 && rt[t].z >= 0 && rt[t].z != p)
          vp[t] = reinterpret_cast<const float4 *>(
              lat_row(m, rt[t].x, fp[t]))[LAT_W * RL4 + slot_of(m, fp[t], fa) * k4 + kq[t]];
      }
#pragma unroll
      for (int t = 0; t < kSingleTrips; t++) {
        if (l[t] < 0) continue;
        const int q = rt[t].z;
        bool touched = false;
        if (q >= 0) {
          if (q != p) {
            ffm_touch4(m.h, p < q, tg, xm, __int_as_float(rt[t].y), vp[t], w4[t], n4[t], z4[t]);
            touched = true;
          }
        } else if (q == -2) {  // several entries of that field in the row: walk them in row order
          for (int qq = s.head[static_cast<int64_t>(r) * F + fp[t]]; qq >= 0; qq = s.next[qq]) {
            if (qq == p) continue;
            const float4 vq = reinterpret_cast<const float4 *>(
                lat_row(m, rows.feat[qq], fp[t]))[LAT_W * RL4 + slot_of(m, fp[t], fa) * k4 + kq[t]];
            ffm_touch4(m.h, p < qq, tg, xm, rows.val[qq], vq, w4[t], n4[t], z4[t]);
            touched = true;
          }
        }
        if (touched) {
          store_nt(rec4 + LAT_N * RL4 + l[t], n4[t]);
          store_nt(rec4 + LAT_Z * RL4 + l[t], z4[t]);
        }
      }
    }
  }
}

// The same for SHORT stored records (a compact shard's: 40 vectors at 10 slots x 16 factors): one
// wave per feature would leave a third of its lanes idle, so here a lane is one (feature, vector)
// of a flat index space and a wave covers the end of one record and the start of the next.  What
// was wave-uniform (descriptor, row table, tmp_grad) becomes a broadcast load per lane.
__global__ __launch_bounds__(kUpdThreads) void ffm_update_single_flat_kernel(ModelDev m, Rows rows,
                                                                             Scratch s) {
  const int RL4 = m.row_len >> 2, k4 = m.n_factors >> 2, F = m.n_fields;
  const int span4 = record_span(m, k4);
  const unsigned total = static_cast<unsigned>(s.counters[CNT_NSINGLE]) * static_cast<unsigned>(span4);
  const double inv_span = 1.0 / static_cast<double>(span4);
  const float inv_k4 = 1.0f / static_cast<float>(k4);
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    // li = t / span4 through the double reciprocal (exact after one correction for t < 2^31)
    int li = static_cast<int>((static_cast<double>(t) + 0.5) * inv_span);
    li += static_cast<unsigned>(li + 1) * span4 <= t ? 1 : (static_cast<unsigned>(li) * span4 > t ? -1 : 0);
    const int ll = static_cast<int>(t - static_cast<unsigned>(li) * span4);  // vector of the stored record
    int sl = static_cast<int>((ll + 0.5f) * inv_k4);                         // its slot
    sl += (sl + 1) * k4 <= ll ? 1 : (sl * k4 > ll ? -1 : 0);
    const int kq = ll - sl * k4;
    const int4 d = s.sdesc[li];  // {feature, entry, row, field}
    const int i = d.x, p = d.y, r = d.z, fa = d.w;
    const int f = walk_field(m, fa, sl);  // partner field of this vector's slot
    if (f < 0 || !owns_bit(owner_bits(m, fa), f)) continue;
    const int4 rt = s.rowtab[static_cast<int64_t>(r) * F + f];
    const int q = rt.z;
    if (q == -1 || q == p) continue;  // no entry of that field in the row (or only this one)
    float4 *rec4 = reinterpret_cast<float4 *>(lat_row(m, i, fa));
    float4 n4 = load_nt(rec4 + LAT_N * RL4 + ll), z4 = load_nt(rec4 + LAT_Z * RL4 + ll);
    const float4 w4 = rec4[LAT_W * RL4 + ll];
    const float xm = rows.val[p], tg = s.tg[r];
    bool touched = false;
    if (q >= 0) {
      const float4 vp = reinterpret_cast<const float4 *>(
          lat_row(m, rt.x, f))[LAT_W * RL4 + slot_of(m, f, fa) * k4 + kq];
      ffm_touch4(m.h, p < q, tg, xm, __int_as_float(rt.y), vp, w4, n4, z4);
      touched = true;
    } else {  // several entries of that field in the row: walk them in row order
      for (int qq = s.head[static_cast<int64_t>(r) * F + f]; qq >= 0; qq = s.next[qq]) {
        if (qq == p) continue;
        const float4 vq = reinterpret_cast<const float4 *>(
            lat_row(m, rows.feat[qq], f))[LAT_W * RL4 + slot_of(m, f, fa) * k4 + kq];
        ffm_touch4(m.h, p < qq, tg, xm, rows.val[qq], vq, w4, n4, z4);
        touched = true;
      }
    }
    if (touched) {
      store_nt(rec4 + LAT_N * RL4 + ll, n4);
      store_nt(rec4 + LAT_Z * RL4 + ll, z4);
    }
  }
}

// FM latent update (fm.cpp:84-95): g = tmp_grad * (x * s_vx - v * x * x), every touch plain.
// Work item = (distinct feature u, chunk of 64 factors), the longest lists first; lane = factor.
// skip_once: the features that occur once in the block were updated by their row (fm_row_wave_kernel)
//
// The operands of a group of kFmUnroll touches (touch descriptor -> value, tmp_grad, the row's s_vx)
// are two dependent trips to memory: they are fetched one group ahead of the arithmetic, the
// descriptors two groups ahead.  The arithmetic of a touch is the gradient and two adds (sum g,
// sum g*g); segments of kSeg touches are joined as they end; one square-root pair per factor and
// block at the very end (kernels_fold.h).  A feature that repeats inside a row (UF_DUP) is walked.
struct FmTouchOps {
  float x[kFmUnroll], tg[kFmUnroll], sv[kFmUnroll];
};
static_assert(kSeg % kFmUnroll == 0, "segments end between groups of touches");
// (block of n_blocks: the workgroups of a launch that walk these lists)
__device__ __forceinline__ void fm_update_body(const ModelDev &m, const Rows &rows, const Scratch &s,
                                               int skip_once, int block, int n_blocks) {
  const int k = m.n_factors;
  const int chunks = (k + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int wave = block * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = n_blocks * kUpdWaves;
  // skip_once: only the features with two or more occurrences are left -- the giant, huge, big and
  // few lists, in that order -- instead of a walk over every distinct feature of the block
  const int n_giant = s.counters[CNT_NGIANT], n_huge = n_giant + s.counters[CNT_NHUGE];
  const int n_big = n_huge + s.counters[CNT_NBIG], n_few = n_big + s.counters[CNT_NFEW];
  const int64_t n_items = static_cast<int64_t>(skip_once ? n_few : s.counters[CNT_NUNIQ]) * chunks;
  for (int64_t item = wave; item < n_items; item += n_waves) {
    int u = static_cast<int>(item / chunks);
    const int e = static_cast<int>(item - static_cast<int64_t>(u) * chunks) * 64 + lane;
    if (e >= k) continue;
    if (skip_once)
      u = wave_uniform(u < n_giant ? s.giant[u] : u < n_huge ? s.huge[u - n_giant]
                       : u < n_big ? s.big[u - n_huge] : s.few[u - n_big]);
    const int4 ud = s.udesc[u];
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (skip_once && c == 1) continue;
    const bool dup = (wave_uniform(s.uflag[start]) & UF_DUP) != 0;
    if (c >= m.giant_min && !dup) continue;  // folded as ranges side by side (fm_range_items)
    float *rec = lat_row(m, i, 0);
    float n = rec[LAT_N * k + e], z = rec[LAT_Z * k + e];
    const float w = rec[LAT_W * k + e];
    if (dup) {  // the row-order walk
      float sqn = sqrt_cr(n);
      for (int t = 0; t < c; t++) {
        const int2 pr = s.occ2[start + t];
        const float x = rows.val[pr.x];
        const float g = s.tg[pr.y] * (x * s.svx[static_cast<int64_t>(pr.y) * k + e] - w * x * x);
        nz_step_latent_carry(m.h, w, g, n, z, sqn);
      }
      rec[LAT_N * k + e] = n;
      rec[LAT_Z * k + e] = z;
      continue;
    }
    Fold a;
    a.init(n);
    const int nb = (c + kFmUnroll - 1) / kFmUnroll;
    auto load_desc = [&](int b, int2 (&pr)[kFmUnroll]) {
#pragma unroll
      for (int j = 0; j < kFmUnroll; j++) pr[j] = s.occ2[start + min(b * kFmUnroll + j, c - 1)];
    };
    auto load_ops = [&](const int2 (&pr)[kFmUnroll], FmTouchOps &o) {
#pragma unroll
      for (int j = 0; j < kFmUnroll; j++) {
        o.x[j] = rows.val[pr[j].x];
        o.tg[j] = s.tg[pr[j].y];
        o.sv[j] = s.svx[static_cast<int64_t>(pr[j].y) * k + e];
      }
    };
    auto compute = [&](const FmTouchOps &o, int b) {
      if (b > 0 && (b * kFmUnroll) % kSeg == 0) a.flush();
      const int cnt = min(kFmUnroll, c - b * kFmUnroll);
#pragma unroll
      for (int j = 0; j < kFmUnroll; j++) {  // fm.cpp:84-95
        const float x = o.x[j];
        a.plain(j < cnt, o.tg[j] * (x * o.sv[j] - w * x * x));
      }
    };
    int2 prA[kFmUnroll], prB[kFmUnroll];
    FmTouchOps opA, opB;
    load_desc(0, prA);
    load_ops(prA, opA);
    load_desc(1, prB);
    for (int b = 0; b < nb; b += 2) {
      // here: opA = operands of group b (in flight), prB = descriptors of group b + 1 (in flight)
      load_ops(prB, opB);
      load_desc(b + 2, prA);
      compute(opA, b);
      if (b + 1 >= nb) break;
      load_ops(prA, opA);
      load_desc(b + 3, prB);
      compute(opB, b + 1);
    }
    if (fold_finish_latent(m.h, a, w, n, z)) {
      rec[LAT_N * k + e] = n;
      rec[LAT_Z * k + e] = z;
    }
  }
}
// Giant FM features (giant_min occurrences or more: under a Zipf law over all ids the top features
// sit in most rows of a block): their occurrences are cut into ranges of kFmRange that waves all over
// the chip fold side by side -- per segment the sums of g and g*g from -0.0f into s.segP / s.segG --
// and a second short launch joins the segments of each (feature, chunk) left to right.  Every
// touch is plain, so the join needs nothing else.
__device__ __forceinline__ void fm_range_items(const ModelDev &m, const Rows &rows, const Scratch &s,
                                               int block, int n_blocks) {
  const int k = m.n_factors;
  const int chunks = (k + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int wave = block * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = n_blocks * kUpdWaves;
  const int64_t n_items = static_cast<int64_t>(s.counters[CNT_NRANGE]) * chunks;
  for (int64_t item = wave; item < n_items; item += n_waves) {
    const int ri = static_cast<int>(item / chunks);
    const int e = static_cast<int>(item - static_cast<int64_t>(ri) * chunks) * 64 + lane;
    const int2 gr = s.grange[ri];  // {index into giant, range number}
    const int gi = wave_uniform(gr.x), r = wave_uniform(gr.y);
    const int4 ud = s.udesc[wave_uniform(s.giant[gi])];
    const int i = wave_uniform(ud.x), start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (wave_uniform(s.uflag[start]) & UF_DUP) continue;  // walked in row order by fm_update_body
    if (e >= k) continue;
    const float w = lat_row(m, i, 0)[LAT_W * k + e];
    const int seg0 = wave_uniform(s.gseg[gi]) + r * (m.range_len / kSeg);
    const int t_lo = r * m.range_len, t_hi = min(c, t_lo + m.range_len);
    const int nb = (t_hi - t_lo + kFmUnroll - 1) / kFmUnroll;
    auto load_desc = [&](int b, int2 (&pr)[kFmUnroll]) {
#pragma unroll
      for (int j = 0; j < kFmUnroll; j++) pr[j] = s.occ2[start + min(t_lo + b * kFmUnroll + j, c - 1)];
    };
    auto load_ops = [&](const int2 (&pr)[kFmUnroll], FmTouchOps &o) {
#pragma unroll
      for (int j = 0; j < kFmUnroll; j++) {
        o.x[j] = rows.val[pr[j].x];
        o.tg[j] = s.tg[pr[j].y];
        o.sv[j] = s.svx[static_cast<int64_t>(pr[j].y) * k + e];
      }
    };
    float P = -0.0f, G = -0.0f;
    auto compute = [&](const FmTouchOps &o, int b) {
      const int cnt = min(kFmUnroll, t_hi - t_lo - b * kFmUnroll);
#pragma unroll
      for (int j = 0; j < kFmUnroll; j++) {  // fm.cpp:84-95
        const float x = o.x[j];
        const float g = o.tg[j] * (x * o.sv[j] - w * x * x);
        G = j < cnt ? G + g : G;
        P = j < cnt ? P + g * g : P;
      }
      if (((b + 1) * kFmUnroll) % kSeg == 0 || b == nb - 1) {  // the segment ends
        const int64_t at = static_cast<int64_t>(seg0 + b * kFmUnroll / kSeg) * k + e;
        s.segP[at] = P;
        s.segG[at] = G;
        P = G = -0.0f;
      }
    };
    int2 prA[kFmUnroll], prB[kFmUnroll];
    FmTouchOps opA, opB;
    load_desc(0, prA);
    load_ops(prA, opA);
    load_desc(1, prB);
    for (int b = 0; b < nb; b += 2) {
      load_ops(prB, opB);
      load_desc(b + 2, prA);
      compute(opA, b);
      if (b + 1 >= nb) break;
      load_ops(prA, opA);
      load_desc(b + 3, prB);
      compute(opB, b + 1);
    }
  }
}
__global__ __launch_bounds__(kUpdThreads) void fm_update_join_kernel(ModelDev m, Scratch s) {
  const int k = m.n_factors;
  const int chunks = (k + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = gridDim.x * kUpdWaves;
  const int64_t n_items = static_cast<int64_t>(s.counters[CNT_NGIANT]) * chunks;
  for (int64_t item = wave; item < n_items; item += n_waves) {
    const int gi = static_cast<int>(item / chunks);
    const int e = static_cast<int>(item - static_cast<int64_t>(gi) * chunks) * 64 + lane;
    const int4 ud = s.udesc[wave_uniform(s.giant[gi])];
    const int i = wave_uniform(ud.x), start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (wave_uniform(s.uflag[start]) & UF_DUP) continue;
    if (e >= k) continue;
    float *rec = lat_row(m, i, 0);
    float n = rec[LAT_N * k + e], z = rec[LAT_Z * k + e];
    const float w = rec[LAT_W * k + e];
    const int64_t base = static_cast<int64_t>(wave_uniform(s.gseg[gi])) * k + e;
    const int n_seg = (c + kSeg - 1) / kSeg;
    Fold a;
    a.init(n);
    constexpr int kFly = 32;  // segments whose sums are in flight together (a chain of dependent loads
                              // on an otherwise idle chip: what is in flight is all that counts)
    for (int s0 = 0; s0 < n_seg; s0 += kFly) {
      float p[kFly], g[kFly];
#pragma unroll
      for (int j = 0; j < kFly; j++) {
        const int64_t at = base + static_cast<int64_t>(min(s0 + j, n_seg - 1)) * k;
        p[j] = s.segP[at];
        g[j] = s.segG[at];
      }
#pragma unroll
      for (int j = 0; j < kFly; j++) {
        if (s0 + j < n_seg) {
          a.P = p[j];
          a.G = g[j];
          a.flush();
        }
      }
    }
    a.any = a.head_plain = true;
    if (fold_finish_latent(m.h, a, w, n, z)) {
      rec[LAT_N * k + e] = n;
      rec[LAT_Z * k + e] = z;
    }
  }
}

// One launch for FM's whole update: the first side_blocks workgroups carry the bias fold (block 0)
// and the linear update, the others the latent lists.
__global__ __launch_bounds__(kUpdThreads) void fm_update_kernel(ModelDev m, Rows rows, Scratch s,
                                                                int skip_once, int side_blocks) {
  __shared__ float s_PG[2 * kUpdThreads];  // (the bias fold's segment totals)
  if (static_cast<int>(blockIdx.x) < side_blocks) {
    if (blockIdx.x == 0) bias_update_body(m, rows.n_rows, s, s_PG);
    else linear_update_body(m, rows, s, blockIdx.x - 1, side_blocks - 1, skip_once);
    return;
  }
  fm_range_items(m, rows, s, blockIdx.x - side_blocks, gridDim.x - side_blocks);
  fm_update_body(m, rows, s, skip_once, blockIdx.x - side_blocks, gridDim.x - side_blocks);
}

}  // namespace ftrl_dev
