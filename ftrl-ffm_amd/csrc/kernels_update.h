// kernels_update.h -- feature-major half of the path: the FTRL (n,z) accumulator update.
// Replaces, for a whole block of rows at once,
//   FtrlModel::update_linear_nz / update_bias_nz   src/model/ftrl_model.cpp:66-85
//   FFM::update_vector_nz                          src/model/ffm.cpp:90-136  (incl. :118)
//   FM::update_vector_nz                           src/model/fm.cpp:80-101
//
// Every distinct feature of the block has exactly one owner.  The owner keeps (n, z, w) in
// registers and applies the block's touches to them in row order -- the same sequence of fp32
// operations the reference's one-thread loop performs with w and tmp_grad frozen at the block
// start -- then writes (n, z) back once.  No float atomics, no locks, bit-reproducible.
//
// Two shapes of owner (the grouping pass sorts features into two lists by occurrence count):
//  * "small" features (<= kSmallMax occurrences, the vast majority): ONE wave owns the whole
//    record, each lane holding 4 consecutive factors of one slot (16-byte loads of n, z, w and of
//    each partner slot), all of a record's gathers in flight together -- bandwidth-shaped.
//  * "hot" features: one wave per 64 elements of the record, sixteen touches at a time, their
//    operands loaded touch-parallel and transposed through LDS (kernels_tile.h); the giant ones --
//    thousands of occurrences -- as touch-parallel DPP chains (kernels_chain.h).
#pragma once
#include "engine_types.h"
#include "kernels_touch.h"

namespace ftrl_dev {

constexpr int kUpdThreads = 256;
constexpr int kUpdWaves = kUpdThreads / 64;
constexpr int kFmUnroll = 8;  // touches per prefetch group in the FM update kernel

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// the 64-bit float offset a haux entry carries in its .z (lo) and .w (hi)
__device__ __forceinline__ int64_t haux_offset(int lo, int hi) {
  return (static_cast<int64_t>(hi) << 32) | static_cast<unsigned>(lo);
}

// (Very hot features -- more than kHugeMin occurrences -- are kernels_chain.h's.)

__device__ __forceinline__ float dpp_row_shr1(float keep, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v),
                                                    0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

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

// Two such prefixes at once, interleaved instruction by instruction: each chain's add fills the
// other's DPP wait state, so two chains cost what one costs (measured, tools/dpp_probe2.hip:
// 3.5 ns per dependent step for either).  sa / sb in, running prefixes out.
__device__ __forceinline__ void wave_sequential_prefix2(float carry_a, float a, float &sa,
                                                        float carry_b, float b, float &sb) {
  const int lane = threadIdx.x & 63;
  sa = lane == 0 ? carry_a + a : a;
  sb = lane == 0 ? carry_b + b : b;
#define FTRL_WSHR2 "v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
                   "v_add_f32_dpp %1, %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
#define FTRL_REP7(x) x x x x x x x
  asm volatile("s_nop 1\n\t" FTRL_REP7(FTRL_REP7(FTRL_WSHR2)) FTRL_REP7(FTRL_WSHR2) FTRL_REP7(FTRL_WSHR2)
               : "+v"(sa), "+v"(sb)
               : "v"(a), "v"(b));
#undef FTRL_REP7
#undef FTRL_WSHR2
}

// The linear/bias accumulator chain over up to 64 gradients held one per lane (lane j = the j-th
// touch, in order; lanes >= count idle).  Sequential semantics of nz_step_linear, evaluated as:
// (1) running n by a strictly sequential prefix sum, (2) every touch's z increment in parallel,
// (3) z by a second sequential prefix -- the same operations on the same values in the same
// order as the one-thread loop, so bit-identical.
__device__ __forceinline__ void linear_chain64(const Hyper &h, float w, float g, int count,
                                               float &n, float &z) {
  const int lane = threadIdx.x & 63;
  const bool live = lane < count;
  const float n_after = wave_sequential_prefix(n, live ? g * g : -0.0f);
  float n_before = __int_as_float(__builtin_amdgcn_update_dpp(
      __float_as_int(n), __float_as_int(n_after), 0x138, 0xf, 0xf, false));
  if (lane == 0) n_before = n;
  const float sgm = div_alpha(h, sqrt_cr(n_after) - sqrt_cr(n_before));  // n_after = n_before + g*g
  const float z_run = wave_sequential_prefix(z, live ? g - sgm * w : -0.0f);
  n = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(n_after), 63));
  z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z_run), 63));
}

// Linear update (update_linear_nz, ftrl_model.cpp:66-77).  Small features: one thread each,
// touches applied one after another; hot features: one wave each, 64 touches per pass.
// ph of `phases` (row phases, engine_types.h): a hot feature's touches of this phase's rows; the
// small features' few touches all in the last phase.
// skip_once: the linear terms of the features that occur once in the block were updated by their
// row (fm_row_wave_kernel)
__device__ __forceinline__ void linear_update_body(const ModelDev &m, const Rows &rows,
                                                   const Scratch &s, int block, int n_blocks,
                                                   int ph = 0, int phases = 1, int skip_once = 0) {
  const int n_small = ph == phases - 1 ? s.counters[CNT_NSMALL] : 0, n_big = s.counters[CNT_NBIG];
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
    float sqn = sqrt_cr(n);
    for (int t = 0; t < c; t++) {
      const int2 pr = s.occ2[start + t];
      nz_step_linear_carry(m.h, w, s.tg[pr.y] * rows.val[pr.x], n, z, sqn);
    }
    m.lin_n[i] = n;
    m.lin_z[i] = z;
  }
  const int lane = threadIdx.x & 63;
  const int wave = block * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = n_blocks * kUpdWaves;
  const int n_huge = s.counters[CNT_NHUGE], n_giant = s.counters[CNT_NGIANT];
  for (int li = wave; li < n_big + n_huge + n_giant; li += n_waves) {
    const int u = wave_uniform(li < n_big ? s.big[li]
                               : li < n_big + n_huge ? s.huge[li - n_big] : s.giant[li - n_big - n_huge]);
    const int4 ud = s.udesc[u];
    if (!owns_linear(m, wave_uniform(ud.w))) continue;
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y);
    int t_lo, c;  // this phase's touches [t_lo, c)
    phase_touches(s, start, wave_uniform(ud.z), ph, phases, t_lo, c);
    t_lo = wave_uniform(t_lo);
    c = wave_uniform(c);
    if (t_lo >= c) continue;
    float n = m.lin_n[i], z = m.lin_z[i];
    const float w = m.lin_w[i];
    for (int t0 = t_lo; t0 < c; t0 += 64) {
      const int cnt = min(64, c - t0);
      float g = 0.0f;
      if (lane < cnt) {
        const int2 pr = s.occ2[start + t0 + lane];
        g = s.tg[pr.y] * rows.val[pr.x];
      }
      linear_chain64(m.h, w, g, cnt, n, z);
    }
    if (lane == 0) {
      m.lin_n[i] = n;
      m.lin_z[i] = z;
    }
  }
}
__global__ __launch_bounds__(kUpdThreads) void linear_update_kernel(ModelDev m, Rows rows,
                                                                    Scratch s, int skip_once) {
  linear_update_body(m, rows, s, blockIdx.x, gridDim.x, 0, 1, skip_once);
}

// Bias update: all rows of the block in order (update_bias_nz, ftrl_model.cpp:79-85) -- n_rows
// dependent touches of ONE accumulator, the longest serial chain of a block.  Rows [row_lo, row_hi)
// of the block (one row phase, or all of them).  Called by every thread of a 256-thread workgroup.
//
// The two recurrences -- n += g*g and z += g - sigma*w -- are 2 * n_rows dependent fp32 adds
// whatever the layout (5.75 cycles each on gfx950, tools/issue_probe.hip: 39 us for 8192 rows); what
// the r03 version added on top was the price of keeping touch t in lane t: 63 dependent
// v_add_f32_dpp wave_shr:1 per 64 rows and chain at 15.5 cycles each (12 ns per row, 100 us per
// 8192-row block -- the floor of FM's whole update phase).  Here a chain runs in the registers of
// one lane instead: 64 increments go through LDS into 64 registers of lane 0 (16 ds_read_b128),
// then 64 plain dependent adds.  Four waves form a pipeline over passes of 64 rows,
// one workgroup barrier per pass:
//   wave 0  pass k   : g*g of its rows (lane = row) -> LDS
//   wave 1  pass k-1 : the running n, every prefix kept -> LDS
//   wave 2  pass k-2 : lane = row: sigma from the n before and after the row, g - sigma*w -> LDS
//   wave 3  pass k-3 : the running z
// Same operations on the same values in the same order as the one-thread loop: bit-identical.
__device__ __forceinline__ void bias_update_body(const ModelDev &m, int row_lo, int row_hi, const Scratch &sc) {
  if (!m.bias_own) return;             // another shard's
  if (sc.counters[CNT_ERROR]) return;  // untrainable block (ERR_ROW_TOO_LONG): a no-op
  constexpr int kRing = 4;
  __shared__ __attribute__((aligned(16))) float s_gg[kRing][64];   // g*g of a pass (-0 for rows past the end)
  __shared__ __attribute__((aligned(16))) float s_na[kRing][64];   // n after row j of the pass
  __shared__ float s_n0[kRing];                                    // n before the pass
  __shared__ __attribute__((aligned(16))) float s_inc[kRing][64];  // g - sigma*w (-0 past the end)
  const float *tg = sc.tg + row_lo;
  const int n_rows = row_hi - row_lo;
  const int lane = threadIdx.x & 63;
  const int wv = wave_uniform(threadIdx.x >> 6);
  const int passes = (n_rows + 63) >> 6;
  const float w = m.bias3[0];
  float acc = wv == 1 ? m.bias3[1] : wv == 3 ? m.bias3[2] : 0.0f;  // wave 1: running n; wave 3: running z
  // the 64 values of a pass added one after another in the registers of LANE 0 (the whole chain
  // under one branch: a predicated store per group of four cost 39 cycles each, tools/issue_probe.hip)
  auto chain64 = [&](const float *src, float *after_out) {
    if (lane != 0) return;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *o4 = reinterpret_cast<float4 *>(after_out);
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = s4[j];
#pragma unroll
    for (int j = 0; j < 16; j++) {  // v[j] becomes the running sum after each of its four rows
      v[j].x = acc + v[j].x;
      v[j].y = v[j].x + v[j].y;
      v[j].z = v[j].y + v[j].z;
      v[j].w = v[j].z + v[j].w;
      acc = v[j].w;
    }
    // (the stores after the whole chain: a store placed behind the add that produces its data holds
    // the next add back -- in-order issue -- and doubled the pass: 1450 against 770 cycles)
    __builtin_amdgcn_sched_barrier(0);
    if (after_out) {
#pragma unroll
      for (int j = 0; j < 16; j++) o4[j] = v[j];
    }
  };
  // waves 0 and 2 read tmp_grad of their rows four passes ahead (a pass takes ~600 cycles, a load
  // that misses the L2 longer): ga = the current four passes' values, gb = the next four's.  The loop
  // is unrolled by four so that no register in flight is ever moved.
  const int off = wv == 2 ? 2 : 0;  // wave 2 works on pass k - 2
  float ga[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto load4 = [&](int pass0, float (&g)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = ((pass0 + q) << 6) + lane;
      g[q] = (pass0 + q >= 0 && r < n_rows) ? tg[r] : 0.0f;
    }
  };
  if (wv == 0 || wv == 2) load4(-off, ga);
  auto step = [&](int k, float g) {
    if (wv == 0) {
      if (k < passes) s_gg[k % kRing][lane] = (k << 6) + lane < n_rows ? g * g : -0.0f;  // x + -0.0f == x bit for bit
    } else if (wv == 1) {
      if (k >= 1 && k - 1 < passes) {
        if (lane == 0) s_n0[(k - 1) % kRing] = acc;
        chain64(s_gg[(k - 1) % kRing], s_na[(k - 1) % kRing]);
      }
    } else if (wv == 2) {
      if (k >= 2 && k - 2 < passes) {
        const int r0 = (k - 2) << 6, rb = (k - 2) % kRing;
        const float na = s_na[rb][lane];
        const float nb = lane == 0 ? s_n0[rb] : s_na[rb][lane == 0 ? 0 : lane - 1];
        const float sgm = div_alpha(m.h, sqrt_cr(na) - sqrt_cr(nb));  // na = nb + g*g
        s_inc[rb][lane] = r0 + lane < n_rows ? g - sgm * w : -0.0f;
      }
    } else {
      if (k >= 3) chain64(s_inc[(k - 3) % kRing], nullptr);
    }
    __syncthreads();
  };
  for (int k0 = 0; k0 < passes + 3; k0 += 8) {
    if (wv == 0 || wv == 2) load4(k0 + 4 - off, gb);
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (k0 + q < passes + 3) step(k0 + q, ga[q]);
    if (wv == 0 || wv == 2) load4(k0 + 8 - off, ga);
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (k0 + 4 + q < passes + 3) step(k0 + 4 + q, gb[q]);
  }
  if (lane == 0 && n_rows > 0) {
    if (wv == 1) m.bias3[1] = acc;
    if (wv == 3) m.bias3[2] = acc;
  }
}
__global__ __launch_bounds__(kUpdThreads) void bias_update_kernel(ModelDev m, int n_rows, Scratch s) {
  bias_update_body(m, 0, n_rows, s);
}

// n_factors not a multiple of 4: every distinct feature, hot or not, is owned per 64 elements
// and gathers its inputs in place (no float4 path, no occurrence-ordered streams).
__global__ __launch_bounds__(kUpdThreads) void ffm_update_generic_kernel(ModelDev m, Rows rows,
                                                                         Scratch s) {
  const int RL = m.row_len, k = m.n_factors, F = m.n_fields;
  const unsigned chunks = (RL + 63) / 64;
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const unsigned n_waves = gridDim.x * kUpdWaves;
  const unsigned n_items = static_cast<unsigned>(s.counters[CNT_NUNIQ]) * chunks;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned u = item / chunks;
    const int e = static_cast<int>(item - u * chunks) * 64 + lane;
    if (e >= RL) continue;
    const int4 ud = s.udesc[u];
    const int fa = ud.w;
    const int sl = e / k, kk = e - sl * k;
    const int fp = walk_field(m, fa, sl);
    if (fp < 0) continue;
    const int i = ud.x;
    const int start = ud.y, c = ud.z;
    float *rec = lat_row(m, i, fa);
    float n = rec[LAT_N * RL + e], z = rec[LAT_Z * RL + e];
    const float w = rec[LAT_W * RL + e];
    bool touched = false;
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
    if (touched) {
      rec[LAT_N * RL + e] = n;
      rec[LAT_Z * RL + e] = z;
    }
  }
}

// ---- small features: one wave per feature, lanes = 4 consecutive factors of a slot ---------
// Requires n_factors % 4 == 0.  c <= kSmallMax occurrences.  Kept lean in registers: the
// bandwidth comes from many resident waves, each with its record's loads in flight.
// few_only: the features that occur once are ffm_update_single_kernel's
#ifndef FFM_SMALL_BATCH
#define FFM_SMALL_BATCH 4
#endif
constexpr int kSmallBatch = FFM_SMALL_BATCH < kSmallMax ? FFM_SMALL_BATCH : kSmallMax;  // touches whose gathers fly together
#ifdef FFM_SMALL_WAVES
#define FFM_SMALL_OCC __attribute__((amdgpu_waves_per_eu(FFM_SMALL_WAVES, FFM_SMALL_WAVES)))
#else
#define FFM_SMALL_OCC
#endif
__device__ __forceinline__ void ffm_small_body(const ModelDev &m, const Rows &rows, const Scratch &s,
                                               int few_only, unsigned bidx, unsigned gdim) {
  const int RL = m.row_len, k = m.n_factors, F = m.n_fields;
  const int RL4 = RL >> 2, k4 = k >> 2;
  const int lane = threadIdx.x & 63;
  const int wave = bidx * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = gdim * kUpdWaves;
  const int *list = few_only ? s.few : s.small;
  const int n_small = s.counters[few_only ? CNT_NFEW : CNT_NSMALL];
  const float inv_k4 = 1.0f / static_cast<float>(k4);
  const int span4 = record_span(m, k4);
  for (int li = wave; li < n_small; li += n_waves) {
    const int u = wave_uniform(list[li]);
    const int4 ud = s.udesc[u];
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    const int fa = wave_uniform(ud.w);
    float4 *rec4 = reinterpret_cast<float4 *>(lat_row(m, i, fa));
    // The feature's (at most kSmallMax) touches, read ONCE for all three passes over its record:
    // entry, row, own field, own value, tmp_grad -- wave-uniform, scalar loads.  (r02: every pass
    // walked occ2 -> rowtab -> partner weights touch after touch, three dependent round trips per
    // touch and pass: 80 % of the kernel's wave-cycles waited on memory with 28 M VALU
    // instructions to issue -- VERDICT r02 weak #6.)
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
    for (int l0 = 0; l0 < span4; l0 += 64) {
      const int l = l0 + lane;  // 16-byte vector of the stored record
      if (l >= span4) continue;
      int sl = static_cast<int>((l + 0.5f) * inv_k4);  // its slot
      sl += (sl + 1) * k4 <= l ? 1 : (sl * k4 > l ? -1 : 0);
      const int fp = walk_field(m, fa, sl);  // partner field of this lane's slot
      if (fp < 0) continue;
      const int kq = l - sl * k4;  // which 16-byte quarter of the slot
      const unsigned long long own_bits = owner_bits(m, fp);
      float4 n4 = rec4[LAT_N * RL4 + l], z4 = rec4[LAT_Z * RL4 + l];
      const float4 w4 = rec4[LAT_W * RL4 + l];
      // kSmallBatch touches at a time: their row-table entries in flight together, then their
      // partners' weights, then the (n, z) steps in touch order
      bool touched = false;
#pragma unroll
      for (int j0 = 0; j0 < kSmallMax; j0 += kSmallBatch) {
        if (j0 >= c) break;
        int4 rt[kSmallBatch];
#pragma unroll
        for (int jj = 0; jj < kSmallBatch; jj++) {
          const int j = j0 + jj;
          rt[jj] = make_int4(0, 0, -1, 0);  // "no entry of that field in the row"
          if (j < c && owns_bit(own_bits, tfm[j])) rt[jj] = s.rowtab[static_cast<int64_t>(tr[j]) * F + fp];
        }
        float4 vp[kSmallBatch];
#pragma unroll
        for (int jj = 0; jj < kSmallBatch; jj++) {
          const int j = j0 + jj;
          vp[jj] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (j < c && rt[jj].z >= 0 && rt[jj].z != tp[j])
            vp[jj] = reinterpret_cast<const float4 *>(lat_row(m, rt[jj].x, fp))[LAT_W * RL4 + slot_of(m, fp, tfm[j]) * k4 + kq];
        }
#pragma unroll
        for (int jj = 0; jj < kSmallBatch; jj++) {
          const int j = j0 + jj;
          if (j >= c) continue;
          const int p = tp[j], q = rt[jj].z;
          if (q >= 0) {
            if (q != p) {
              ffm_touch4(m.h, p < q, ttg[j], txm[j], __int_as_float(rt[jj].y), vp[jj], w4, n4, z4);
              touched = true;
            }
          } else if (q == -2) {  // several entries of that field in the row: walk them in row order
            for (int qq = s.head[static_cast<int64_t>(tr[j]) * F + fp]; qq >= 0; qq = s.next[qq]) {
              if (qq == p) continue;
              const float4 vq = reinterpret_cast<const float4 *>(
                  lat_row(m, rows.feat[qq], fp))[LAT_W * RL4 + slot_of(m, fp, tfm[j]) * k4 + kq];
              ffm_touch4(m.h, p < qq, ttg[j], txm[j], rows.val[qq], vq, w4, n4, z4);
              touched = true;
            }
          }
        }
      }
      if (touched) {
        rec4[LAT_N * RL4 + l] = n4;
        rec4[LAT_Z * RL4 + l] = z4;
      }
    }
  }
}
__global__ __launch_bounds__(kUpdThreads) FFM_SMALL_OCC void ffm_update_small_kernel(ModelDev m, Rows rows,
                                                                       Scratch s, int few_only) {
  ffm_small_body(m, rows, s, few_only, blockIdx.x, gridDim.x);
}

// The same for SHORT stored records (a compact shard's), a lane = one (feature, vector) of a flat
// index space -- see ffm_update_single_flat_kernel below.
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
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    int li = static_cast<int>((static_cast<double>(t) + 0.5) * inv_span);  // t / span4, corrected
    li += static_cast<unsigned>(li + 1) * span4 <= t ? 1 : (static_cast<unsigned>(li) * span4 > t ? -1 : 0);
    const int l = static_cast<int>(t - static_cast<unsigned>(li) * span4);  // vector of the stored record
    int sl = static_cast<int>((l + 0.5f) * inv_k4);                         // its slot
    sl += (sl + 1) * k4 <= l ? 1 : (sl * k4 > l ? -1 : 0);
    const int kq = l - sl * k4;
    const int4 ud = s.udesc[list[li]];  // {feature, start, count, field}
    const int i = ud.x, start = ud.y, c = ud.z, fa = ud.w;
    const int fp = walk_field(m, fa, sl);  // partner field of this vector's slot
    if (fp < 0) continue;
    const unsigned long long own_bits = owner_bits(m, fp);
    float4 *rec4 = reinterpret_cast<float4 *>(lat_row(m, i, fa));
    float4 n4 = rec4[LAT_N * RL4 + l], z4 = rec4[LAT_Z * RL4 + l];
    const float4 w4 = rec4[LAT_W * RL4 + l];
    bool touched = false;
    for (int j = 0; j < c; j++) {
      const int2 pr = s.occ2[start + j];
      const int p = pr.x, r = pr.y;
      const int fm = rows.field[p];
      if (!owns_bit(own_bits, fm)) continue;
      const float xm = rows.val[p], tg = s.tg[r];
      const int4 rt = s.rowtab[static_cast<int64_t>(r) * F + fp];
      const int q = rt.z;
      if (q >= 0) {
        if (q != p) {
          const float4 vp = reinterpret_cast<const float4 *>(
              lat_row(m, rt.x, fp))[LAT_W * RL4 + slot_of(m, fp, fm) * k4 + kq];
          ffm_touch4(m.h, p < q, tg, xm, __int_as_float(rt.y), vp, w4, n4, z4);
          touched = true;
        }
      } else if (q == -2) {
        for (int qq = s.head[static_cast<int64_t>(r) * F + fp]; qq >= 0; qq = s.next[qq]) {
          if (qq == p) continue;
          const float4 vp = reinterpret_cast<const float4 *>(
              lat_row(m, rows.feat[qq], fp))[LAT_W * RL4 + slot_of(m, fp, fm) * k4 + kq];
          ffm_touch4(m.h, p < qq, tg, xm, rows.val[qq], vp, w4, n4, z4);
          touched = true;
        }
      }
    }
    if (touched) {
      rec4[LAT_N * RL4 + l] = n4;
      rec4[LAT_Z * RL4 + l] = z4;
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
          n4[t] = rec4[LAT_N * RL4 + ll];
          z4[t] = rec4[LAT_Z * RL4 + ll];
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
          rec4[LAT_N * RL4 + l[t]] = n4[t];
          rec4[LAT_Z * RL4 + l[t]] = z4[t];
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
    float4 n4 = rec4[LAT_N * RL4 + ll], z4 = rec4[LAT_Z * RL4 + ll];
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
      rec4[LAT_N * RL4 + ll] = n4;
      rec4[LAT_Z * RL4 + ll] = z4;
    }
  }
}

// FM latent update.  Work item = (distinct feature u, chunk of 64 factors), the longest lists first.
// skip_huge: the features with more than kHugeMin occurrences belong to fm_update_huge_kernel
// skip_once: the features that occur once in the block were updated by their row (fm_row_wave_kernel)
//
// A feature's touches form one dependent chain per factor (fm.cpp:84-95), but only two recurrences
// are serial: n += g*g and z = (z + g) - s*w.  The gradients depend on the frozen w alone, so a
// group of kFmUnroll touches is: the gradients, the running n (kFmUnroll dependent adds), ONE range
// vote, kFmUnroll independent square roots and alpha divides, the running z.  The operands of a
// group (touch descriptor -> value, tmp_grad, the row's s_vx) are two dependent trips to memory:
// they are fetched one group ahead of the arithmetic, the descriptors two groups ahead.
struct FmTouchOps {
  float x[kFmUnroll], tg[kFmUnroll], sv[kFmUnroll];
};
// (block of n_blocks: the workgroups of a launch that walk these lists)
__device__ __forceinline__ void fm_update_body(const ModelDev &m, const Rows &rows, const Scratch &s,
                                               int skip_huge, int skip_once, int block, int n_blocks) {
  const int k = m.n_factors;
  const int chunks = (k + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int wave = block * kUpdWaves + wave_uniform(threadIdx.x >> 6);
  const int n_waves = n_blocks * kUpdWaves;
  // skip_once: only the features with 2 .. kHugeMin occurrences are left -- the `few` and `big`
  // lists -- instead of a walk over every distinct feature of the block
  const int n_few = s.counters[CNT_NFEW], n_big = s.counters[CNT_NBIG];
  const bool lists = skip_once && skip_huge;
  const int64_t n_items = static_cast<int64_t>(lists ? n_few + n_big : s.counters[CNT_NUNIQ]) * chunks;
  for (int64_t it = wave; it < n_items; it += n_waves) {
    // (lists: big after few -- walk backwards so that the long chains start first)
    const int64_t item = lists ? n_items - 1 - it : it;
    int u = static_cast<int>(item / chunks);
    const int e = static_cast<int>(item - static_cast<int64_t>(u) * chunks) * 64 + lane;
    if (e >= k) continue;
    if (lists) u = wave_uniform(u < n_few ? s.few[u] : s.big[u - n_few]);
    const int4 ud = s.udesc[u];
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (skip_huge && c > m.huge_min) continue;
    if (skip_once && c == 1) continue;
    float *rec = lat_row(m, i, 0);
    float n = rec[LAT_N * k + e], z = rec[LAT_Z * k + e];
    const float w = rec[LAT_W * k + e];
    float sqn = sqrt_cr(n);
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
      const int cnt = min(kFmUnroll, c - b * kFmUnroll);
      float g[kFmUnroll], na[kFmUnroll];
      bool ok = m.h.fast_div != 0 && cnt == kFmUnroll && chain_operand_ok(n);
      float run = n;
#pragma unroll
      for (int j = 0; j < kFmUnroll; j++) {  // fm.cpp:84-95
        const float x = o.x[j];
        g[j] = o.tg[j] * (x * o.sv[j] - w * x * x);
        run = run + g[j] * g[j];
        na[j] = run;
        ok = ok && chain_operand_ok(run);
      }
      if (__all(ok)) {
        float sa[kFmUnroll], sg[kFmUnroll];
#pragma unroll
        for (int j = 0; j < kFmUnroll; j++) sa[j] = sqrt_fast(na[j]);
#pragma unroll
        for (int j = 0; j < kFmUnroll; j++) sg[j] = div_alpha_fast(m.h, sa[j] - (j ? sa[j - 1] : sqn));
#pragma unroll
        for (int j = 0; j < kFmUnroll; j++) z = (z + g[j]) - sg[j] * w;
        n = na[kFmUnroll - 1];
        sqn = sa[kFmUnroll - 1];
      } else {  // a partial group, or operands outside the short forms' range: touch by touch
#pragma unroll
        for (int j = 0; j < kFmUnroll; j++)
          if (j < cnt) nz_step_latent_carry(m.h, w, g[j], n, z, sqn);
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
    rec[LAT_N * k + e] = n;
    rec[LAT_Z * k + e] = z;
  }
}
// side_blocks: the first workgroups carry the bias chain (block 0) and the linear update -- short
// serial chains that would otherwise need a stream of their own (as in ffm_update_tile_kernel).
__global__ __launch_bounds__(kUpdThreads) void fm_update_kernel(ModelDev m, Rows rows, Scratch s,
                                                                int skip_huge, int skip_once,
                                                                int side_blocks) {
  if (static_cast<int>(blockIdx.x) < side_blocks) {
    if (blockIdx.x == 0) {
      __builtin_amdgcn_s_setprio(3);  // one wave, n_rows dependent touches
      bias_update_body(m, 0, rows.n_rows, s);
    } else {
      linear_update_body(m, rows, s, blockIdx.x - 1, side_blocks - 1, 0, 1, skip_once);
    }
    return;
  }
  fm_update_body(m, rows, s, skip_huge, skip_once, blockIdx.x - side_blocks, gridDim.x - side_blocks);
}

}  // namespace ftrl_dev
