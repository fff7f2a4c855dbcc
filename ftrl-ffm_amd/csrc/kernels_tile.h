// kernels_tile.h -- the FTRL (n, z) update of HOT features: those that occur in more than kSmallMax
// rows of the block (FFM::update_vector_nz src/model/ffm.cpp:90-136 incl. :118, applied to one
// element by tens to thousands of rows).
//
// A hot feature's touches form one sequential chain per element: n_t = n_{t-1} + g_t^2,
// z_t = (z_{t-1} + g_t) - sigma_t * w.  Two things are wanted at once and ask for opposite layouts:
//  * the chain itself is cheapest with lane = element: the two recurrences are then plain dependent
//    v_add_f32 / v_sub_f32 in one lane, three instructions per touch for 64 elements (the previous
//    very-hot kernel laid the touches across the lanes of a DPP row and paid fifteen dependent
//    v_add_f32_dpp per recurrence and 16 touches -- 0.48 serial instructions per element-touch, half
//    of its wave-cycles issue stalls; this shape pays 0.05);
//  * the operands of a touch (its fact, then -- at the address the fact carries -- the partner's
//    weights) are two dependent gathers; they are cheapest touch-parallel: one load instruction
//    fetches them for sixteen touches, so two instructions in flight cover 32 touches of latency.
// So a wave works on TILES of 16 touches x 64 elements: the stager half of the loop loads a tile's
// facts and partner weights with lane = (touch, 16-byte quad), writes them to the wave's own LDS
// region, and the chain half reads them back with lane = element ("transposed through LDS") and
// runs every arithmetic step of the touch in that layout.  Same operations on the same values in
// the same order as the one-thread loop: bit-identical.
//
// Work item = (hot feature, chunk of its stored record): a chunk is 64 / k whole slots (k <= 64) or
// 64 factors of one slot (k > 64).  Pipeline per wave, tile st (D = kTileDV):
//   partner weights of tile st -> LDS transposer
//   facts of tile st+D (registers, loaded a tile ago) -> LDS records {tmp_grad, x_own*x_other,
//                         flags, offset}
//   facts of tile st+D+1 requested (global -> registers)
//   partner weights of tile st+D requested at the offsets just staged (global -> registers): with
//                         the tiles st+1 .. st+D-1 that is 16 D touches of gathers in flight
//   chain arithmetic of tile st
// The ORDER of the two requests matters: the memory counter (vmcnt) retires loads in issue order,
// so the facts the next tile's staging waits for must be requested BEFORE this tile's gathers --
// requested after them, every tile waited for the gathers it had just issued (a gather into a
// 247 GB tensor is a page walk and a trip to HBM).
// The loop is unrolled by D so that the D sets of weight registers keep their names (no moves).
// Chunks with a multi-valued partner field somewhere (s.cmask, rare) take a plain tile-after-tile
// loop that walks the chains.
#pragma once
#include "engine_types.h"
#include "kernels_touch.h"
#include "kernels_update.h"
#include "kernels_chain.h"

namespace ftrl_dev {

constexpr int kTileT = 16;    // touches per tile
constexpr int kTileRow = 80;  // floats per transposer row: 64 + 16 of padding (16-byte writes of
                              // consecutive touches land 16 banks apart)
#ifndef FFM_TILE_DV
#define FFM_TILE_DV 1
#endif
constexpr int kTileDV = FFM_TILE_DV;  // tiles whose partner weights are in flight (1 .. 3)
constexpr int kTileNR = kTileDV == 1 ? 2 : 4;  // fact-record buffers (a power of two): the tiles in flight and the one being applied
                                              // (k = 4: 4 KB each per wave -- two of them leave three workgroups per CU, four leave one)
#ifndef FFM_TILE_G
#define FFM_TILE_G 4
#endif
constexpr int kTileG = FFM_TILE_G;  // touches per arithmetic group (one range vote per group)

// flags word of an LDS fact record: bits 0..7 high bits of the weights' offset, 8..10 HF_* flags,
// 16..23 the touch's own field
__device__ __forceinline__ int tile_hiword(int off_hi, int flags, int own_field) {
  return (off_hi & 0xff) | (flags << 8) | (own_field << 16);
}

// One plain touch given the product of the two values (x_own * x_other == x_other * x_own bit for bit).
__device__ __forceinline__ void ffm_touch_x(const Hyper &h, bool own_first, float tg, float x, float vp,
                                            float w, float &n, float &z) {
  if (own_first) {
    nz_step_latent(h, w, tg * vp * x, n, z);  // ffm.cpp:112-115
  } else {
    nz_step_latent_jside(h, w, tg * vp * x, tg * w * x, n, z);  // ffm.cpp:117-120 with the :118 quirk
  }
}

// kTileG consecutive plain touches of one element (lane = element).  hw: the records' flag words.
// Per group: the gradients, the running n (plain dependent adds), ONE range vote, the touches'
// sigma * w from their n-before (independent given the running n), the running z.
__device__ __forceinline__ void tile_touch_group(const Hyper &h, const float (&tg)[kTileG],
                                                 const float (&x)[kTileG], const int (&hw)[kTileG],
                                                 const float (&vp)[kTileG], float w, float &n, float &z,
                                                 float &sqn, bool &sq_valid, bool &touched) {
  bool live[kTileG], gg_side = true;
  float gj[kTileG], aj[kTileG], nbv[kTileG], naf[kTileG], mj[kTileG];
#pragma unroll
  for (int j = 0; j < kTileG; j++) {
    live[j] = (hw[j] & (HF_SIMPLE << 8)) != 0;
    const bool first = (hw[j] & (HF_FIRST << 8)) != 0;
    const float g = tg[j] * vp[j] * x[j];  // own slot's gradient (g1 if own entry first, else g2)
    const float g1 = tg[j] * w * x[j];     // second-entry case: the first entry's gradient
    gj[j] = g;
    aj[j] = first ? g * g : g * g1;        // what the square root sees added to n (ffm.cpp:113 / :118)
    gg_side = gg_side && (!live[j] || first);
  }
#pragma unroll
  for (int j = 0; j < kTileG; j++) {
    nbv[j] = n;
    if (live[j]) n = n + gj[j] * gj[j];
    naf[j] = n;
  }
  // Every live touch on the g*g side (own entry first): sqrt(n + g*g) of a touch IS sqrt(n-before)
  // of the next, so one square root per touch, the last one carried into the next group.
  bool done = false;
  if (__all(gg_side)) {
    bool ok = h.fast_div != 0 && (sq_valid || chain_operand_ok(nbv[0]));
#pragma unroll
    for (int j = 0; j < kTileG; j++) ok = ok && chain_operand_ok(naf[j]);
    if (__all(ok)) {
      float sq = sqn;
      if (!sq_valid) {
        asm volatile("" ::: "memory");  // (a real branch, not a select with the root always evaluated)
        sq = sqrt_fast(nbv[0]);
      }
#pragma unroll
      for (int j = 0; j < kTileG; j++) {
        const float sa = sqrt_fast(naf[j]);
        mj[j] = div_alpha_fast(h, sa - sq) * w;
        sq = sa;
      }
      sqn = sq;
      done = true;
    }
  }
  sq_valid = done;
  if (!done) {
    float arg[kTileG];
    bool ok = h.fast_div != 0;
#pragma unroll
    for (int j = 0; j < kTileG; j++) {
      arg[j] = nbv[j] + aj[j];
      ok = ok && chain_operand_ok(arg[j]) && chain_operand_ok(nbv[j]);
    }
    if (__all(ok)) {
#pragma unroll
      for (int j = 0; j < kTileG; j++)
        mj[j] = div_alpha_fast(h, sqrt_fast(arg[j]) - sqrt_fast(nbv[j])) * w;
    } else {
#pragma unroll
      for (int j = 0; j < kTileG; j++) mj[j] = ((sqrtf(arg[j]) - sqrtf(nbv[j])) / h.alpha) * w;
    }
  }
#pragma unroll
  for (int j = 0; j < kTileG; j++)
    if (live[j]) { z = (z + gj[j]) - mj[j]; touched = true; }
}

struct TileWeights {  // a loader lane's four 16-byte quads of one tile's partner weights
  float4 q0, q1, q2, q3;
};
struct TileFacts {  // what a stager lane holds of one tile: NF facts of its touch and the touch's meta
  int4 ax[4];
  float2 mt;
};

// NF: facts a stager lane carries per tile = ceil(slots per chunk / 4): 1 for k >= 16, 2 for
// k = 8 / 12, 4 for k = 4.  T: the wave's transposer [kTileT][kTileRow]; R: its fact records
// [kTileNR][kTileT * 4 * NF].
template <int NF>
__device__ __forceinline__ void ffm_tile_items(const ModelDev &m, const Rows &rows, const Scratch &s,
                                               unsigned wave, unsigned n_waves, int ph, int phases,
                                               int lists, float *T, float4 *R) {
  constexpr int RS = 4 * NF;  // records per touch in R (>= slots per chunk)
  const int K = m.n_factors, F = m.n_fields, RL = m.row_len;
  const int lane = threadIdx.x & 63;
  const int tl = lane >> 2, cq = lane & 3;  // stager / loader layout: touch, 16-byte quad column
  const int SPC = K <= 64 ? 64 / K : 1;     // slots per chunk
  const int cps = K <= 64 ? 1 : (K + 63) >> 6;  // chunks per slot
  const int slots = record_span(m, 1);
  const unsigned per_feat = K <= 64 ? (slots + SPC - 1) / SPC : slots * cps;
  // the lists this launch takes (bits of `lists`: 1 big, 2 huge, 4 giant; the engine: big and huge --
  // the giant list is the DPP kernel's), longest chains first: [giant | huge | big]
  const unsigned n_giant = (lists & 4) ? static_cast<unsigned>(s.counters[CNT_NGIANT]) : 0u;
  const unsigned n_huge = n_giant + ((lists & 2) ? static_cast<unsigned>(s.counters[CNT_NHUGE]) : 0u);
  const unsigned n_items = (n_huge + ((lists & 1) ? static_cast<unsigned>(s.counters[CNT_NBIG]) : 0u)) * per_feat;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned li = item / per_feat;
    const int ci = static_cast<int>(item - li * per_feat);
    // (the very hot list first: its long chains start first and run at raised issue priority)
    const int u = wave_uniform(li < n_giant ? s.giant[li] : li < n_huge ? s.huge[li - n_giant] : s.big[li - n_huge]);
    const int4 ud = s.udesc[u];  // {feature, start, count, field}
    const int fa = wave_uniform(ud.w);
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y);
    int t_lo, c;  // this row phase's touches [t_lo, c) of the feature's occurrences
    phase_touches(s, start, wave_uniform(ud.z), ph, phases, t_lo, c);
    t_lo = wave_uniform(t_lo);
    c = wave_uniform(c);
    if (t_lo >= c) continue;
    const int sb = K <= 64 ? ci * SPC : ci / cps;                    // first slot of the chunk
    const int kk0 = K <= 64 ? 0 : (ci - sb * cps) * 64;              // first factor (k > 64)
    const int width = K <= 64 ? SPC * K : min(64, K - kk0);          // elements of the chunk
    const int fp0 = wave_uniform(walk_field(m, fa, sb));
    if (fp0 < 0) continue;  // (valid walk positions are a prefix: nothing stored from here on)
    // ---- chain layout: lane = element of the chunk (idle lanes repeat the first four) ----
    const bool inw = lane < width;
    const int le = inw ? lane : (lane & 3);
    const int es = K <= 64 ? le / K : 0;
    const int ekk = K <= 64 ? le - es * K : kk0 + le;
    const int fp = sb + es < slots ? walk_field(m, fa, sb + es) : -1;
    const bool active = inw && fp >= 0;
    if (s.gmask) {  // no row of the block touches any slot of the chunk
      const unsigned long long gm = s.gmask[start];
      if (!__any(active && ((gm >> fp) & 1ull))) continue;
    }
    // issue priority by chain length: the longest chains bound the update phase
#ifndef FFM_TILE_PRIO_LONG
#define FFM_TILE_PRIO_LONG 640
#endif
    if (c - t_lo > FFM_TILE_PRIO_LONG) __builtin_amdgcn_s_setprio(3);
    else if (li < n_huge) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
    float *rec = lat_row(m, i, fa) + (fp >= 0 ? sb + es : sb) * K + ekk;
    float n = rec[LAT_N * RL], z = rec[LAT_Z * RL];
    const float w = rec[LAT_W * RL];
    float sqn = 0.0f;
    bool sq_valid = false, touched = false;
    // ---- stager layout: lane (touch tl, column cq) carries the facts of slots cq + 4 j ----
    const int4 *acol[NF];
    bool okS[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const int sj = cq + 4 * j;
      const int f = (sj < SPC && sb + sj < slots) ? walk_field(m, fa, sb + sj) : -1;
      okS[j] = f >= 0;
      acol[j] = s.haux + static_cast<int64_t>(start) * F + (okS[j] ? f : fp0);
    }
    const float2 *mcol = s.hmeta + start;
    // ---- loader layout: lane (touch tl, column cq) fetches the quads 4 r + cq of the chunk ----
    int sQ[4], kkQ[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int e4 = 4 * (4 * r + cq);
      const bool inq = e4 < width;
      sQ[r] = inq && K <= 64 ? e4 / K : 0;
      kkQ[r] = inq ? (K <= 64 ? e4 - sQ[r] * K : kk0 + e4) : 0;
    }
    const int steps = (c - t_lo + kTileT - 1) / kTileT;
    auto load_facts = [&](int st, TileFacts &f) {
      const int t = min(t_lo + st * kTileT + tl, c - 1);  // past the end: repeats, staged as dead
#pragma unroll
      for (int j = 0; j < NF; j++) f.ax[j] = acol[j][static_cast<int64_t>(t) * F];
      f.mt = mcol[t];
    };
    // facts of tile st -> LDS records; true when a live touch of the tile has a multi-valued partner field
    auto stage_facts = [&](int st, const TileFacts &f) {
      const bool in_range = t_lo + st * kTileT + tl < c;
      float4 *Rb = R + (st & (kTileNR - 1)) * (kTileT * RS) + tl * RS;
      bool chainy = false;
#pragma unroll
      for (int j = 0; j < NF; j++) {
        const int sj = cq + 4 * j;
        if (sj >= SPC) continue;
        int fl = f.ax[j].y & 0xff;
        const int own_field = f.ax[j].y >> 8;
        if (!in_range || !okS[j]) fl = 0;
        if (m.h.learn && (fl & HF_SIMPLE)) fl |= HF_FIRST;  // the variant uses g2*g2 at ffm.cpp:118
        chainy = chainy || (fl & HF_CHAIN);
        const float x = f.mt.y * __int_as_float(f.ax[j].x);
        Rb[sj] = make_float4(f.mt.x, x, __int_as_float(tile_hiword(f.ax[j].w, fl, own_field)),
                             __int_as_float(f.ax[j].z));
      }
      return __any(chainy);
    };
    // (named members, handed over by value: as arrays behind references these sixteen registers
    // ended up in scratch memory, every load waited for at once)
    auto issue_weight = [&](const float4 *Rb, int r) {
      const float4 rc = Rb[sQ[r]];
      const int64_t off = haux_offset(__float_as_int(rc.w), __float_as_int(rc.z) & 0xff);
      return *reinterpret_cast<const float4 *>(m.lat + off + kkQ[r]);
    };
    auto issue_weights = [&](int st) {
      const float4 *Rb = R + (st & (kTileNR - 1)) * (kTileT * RS) + tl * RS;
      TileWeights v;
      v.q0 = issue_weight(Rb, 0);
      v.q1 = issue_weight(Rb, 1);
      v.q2 = issue_weight(Rb, 2);
      v.q3 = issue_weight(Rb, 3);
      return v;
    };

    auto transpose = [&](const TileWeights &v) {
      float4 *Tw = reinterpret_cast<float4 *>(T + tl * kTileRow + 4 * cq);
      Tw[0] = v.q0;
      Tw[4] = v.q1;
      Tw[8] = v.q2;
      Tw[12] = v.q3;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // the chain arithmetic of tile st, all of its touches plain
    auto apply_tile = [&](int st) {
      const float4 *Rc = R + (st & (kTileNR - 1)) * (kTileT * RS) + es;
      const float *Tc = T + lane;
      const int cnt = min(kTileT, c - t_lo - st * kTileT);  // live touches of this tile
      for (int g0 = 0; g0 < cnt; g0 += kTileG) {
        float tg[kTileG], x[kTileG], vp[kTileG];
        int hw[kTileG];
#pragma unroll
        for (int j = 0; j < kTileG; j++) {
          const float4 rc = Rc[(g0 + j) * RS];
          tg[j] = rc.x;
          x[j] = rc.y;
          hw[j] = __float_as_int(rc.z);
          vp[j] = Tc[(g0 + j) * kTileRow];
        }
        tile_touch_group(m.h, tg, x, hw, vp, w, n, z, sqn, sq_valid, touched);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // the same for a tile that may hold touches with a multi-valued partner field: one touch after
    // another, chains walked
    auto apply_tile_walk = [&](int st) {
      const float4 *Rc = R + (st & (kTileNR - 1)) * (kTileT * RS) + es;
      const float *Tc = T + lane;
      const int cnt = min(kTileT, c - t_lo - st * kTileT);
      sq_valid = false;
      for (int tt = 0; tt < cnt; tt++) {
        const float4 rc = Rc[tt * RS];
        const int hw = __float_as_int(rc.z);
        if (!active) continue;
        if (hw & (HF_SIMPLE << 8)) {
          ffm_touch_x(m.h, (hw & (HF_FIRST << 8)) != 0, rc.x, rc.y, Tc[tt * kTileRow], w, n, z);
          touched = true;
        } else if (hw & (HF_CHAIN << 8)) {
          const int fm = (hw >> 16) & 0xff;
          const int p = s.occ2[start + t_lo + st * kTileT + tt].x;  // the touch's own entry
          const int r = s.row_of[p];
          const float xm = rows.val[p];
          for (int qq = s.head[static_cast<int64_t>(r) * F + fp]; qq >= 0; qq = s.next[qq]) {
            if (qq == p) continue;
            const float vq = m.lat[w_slot_offset(m, rows.feat[qq], fp, fm) + ekk];
            ffm_touch(m.h, p < qq, rc.x, xm, rows.val[qq], vq, w, n, z);
            touched = true;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // Does any slot of the chunk have a multi-valued partner field in some row of the feature?
    bool walk = true;
    if (s.cmask) {
      const unsigned long long cm = s.cmask[start];
      walk = __any(active && ((cm >> fp) & 1ull));
    }
    if (walk) {
      for (int st = 0; st < steps; st++) {
        TileFacts f;
        load_facts(st, f);
        (void)stage_facts(st, f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const TileWeights v = issue_weights(st);
        transpose(v);
        apply_tile_walk(st);
      }
    } else {
      static_assert(kTileDV >= 1 && kTileDV <= 3 && kTileDV + 1 <= kTileNR, "weights of 1 .. 3 tiles in flight");
      TileFacts fN;
      TileWeights V0, V1, V2;
      {
        TileFacts f0, f1, f2;
        load_facts(0, f0);
        if (kTileDV > 1 && steps > 1) load_facts(1, f1);
        if (kTileDV > 2 && steps > 2) load_facts(2, f2);
        if (steps > kTileDV) load_facts(kTileDV, fN);
        (void)stage_facts(0, f0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        V0 = issue_weights(0);
        if (kTileDV > 1 && steps > 1) {
          (void)stage_facts(1, f1);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          V1 = issue_weights(1);
        }
        if (kTileDV > 2 && steps > 2) {
          (void)stage_facts(2, f2);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          V2 = issue_weights(2);
        }
      }
#define FTRL_TILE_STEP(ST, V)                                                      \
      if ((ST) < steps) {                                                          \
        transpose(V);                                                              \
        const bool more__ = (ST) + kTileDV < steps;                                \
        if (more__) {                                                              \
          (void)stage_facts((ST) + kTileDV, fN);                                   \
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                   \
        }                                                                          \
        if ((ST) + kTileDV + 1 < steps) load_facts((ST) + kTileDV + 1, fN);        \
        if (more__) V = issue_weights((ST) + kTileDV);                             \
        apply_tile(ST);                                                            \
      }
      for (int st = 0; st < steps; st += kTileDV) {
        FTRL_TILE_STEP(st, V0)
        if (kTileDV > 1) { FTRL_TILE_STEP(st + 1, V1) }
        if (kTileDV > 2) { FTRL_TILE_STEP(st + 2, V2) }
      }
#undef FTRL_TILE_STEP
    }
    if (touched && active) {
      rec[LAT_N * RL] = n;
      rec[LAT_Z * RL] = z;
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

// One launch for the hot features of a block (the huge and big lists: below the block's giant_min
// occurrences; longer chains take the touch-parallel DPP chains of kernels_chain.h).  side_blocks > 0: the first
// workgroups carry the block's two short serial jobs -- workgroup 0 the bias chain, workgroups
// 1..side_blocks-1 the linear update -- so that they run beside the latent chains without a stream
// (and a hardware queue) of their own.
template <int NF>
__device__ __forceinline__ void ffm_tile_part(const ModelDev &m, const Rows &rows, const Scratch &s,
                                              int side_blocks, int ph, int phases, int lists,
                                              unsigned bidx, unsigned gdim) {
  __shared__ __attribute__((aligned(16))) float lds_T[kUpdWaves][kTileT * kTileRow];
  __shared__ float4 lds_R[kUpdWaves][kTileNR * kTileT * 4 * NF];
  if (static_cast<int>(bidx) < side_blocks) {
    if (bidx == 0) {
      // 2 n_rows dependent adds: let its waves win the issue arbitration on their SIMDs
      __builtin_amdgcn_s_setprio(3);
      // (the whole chain in the last row phase: the row kernel of a later phase still derives the
      // bias weight from the block-start bias_n / bias_z)
      if (ph == phases - 1) bias_update_body(m, 0, rows.n_rows, s);
    } else {
      linear_update_body(m, rows, s, bidx - 1, side_blocks - 1, ph, phases);
    }
    return;
  }
  const unsigned wv = wave_uniform(threadIdx.x >> 6);
  ffm_tile_items<NF>(m, rows, s, (bidx - side_blocks) * kUpdWaves + wv, (gdim - side_blocks) * kUpdWaves,
                     ph, phases, lists, lds_T[wv], lds_R[wv]);
}
template <int NF>
__global__ __launch_bounds__(kUpdThreads) void ffm_update_tile_kernel(ModelDev m, Rows rows, Scratch s,
                                                                      int side_blocks, int ph, int phases,
                                                                      int lists) {
  ffm_tile_part<NF>(m, rows, s, side_blocks, ph, phases, lists, blockIdx.x, gridDim.x);
}

// The whole FFM update of a block in ONE launch on the main stream (one shard, one row phase): the
// workgroup ranges [bias + linear | giant and very hot chains | hot tiles | few-occurrence | loss sum].
// No fork / join between streams: the two event hops per block are a quarter of a small block's step
// (engine_step.h fuses blocks whose update phase is estimated under 100 us).  G: interleaved DPP
// chains per wave (kernels_chain.h).
template <int NF, int G>
__global__ __launch_bounds__(kUpdThreads) void ffm_update_all_tile_kernel(ModelDev m, Rows rows, Scratch s,
                                                                          int side_blocks, int giant_blocks,
                                                                          int nc, int nt, int ns, int few_only,
                                                                          int loss_blocks, double *loss_out,
                                                                          double *loss_scratch) {
  int r = blockIdx.x;
  if (r < side_blocks) { ffm_tile_part<NF>(m, rows, s, side_blocks, 0, 1, 3, r, side_blocks + nt); return; }
  r -= side_blocks;
  if (r < giant_blocks + nc) { ffm_chain_body<G>(m, rows, s, giant_blocks, 0, 1, r, giant_blocks + nc); return; }
  r -= giant_blocks + nc;
  if (r < nt) { ffm_tile_part<NF>(m, rows, s, side_blocks, 0, 1, 3, side_blocks + r, side_blocks + nt); return; }
  r -= nt;
  if (r < ns) { ffm_small_body(m, rows, s, few_only, r, ns); return; }
  loss_sum_body(rows.n_rows, s.loss, loss_out, loss_scratch, r - ns, loss_blocks);
}

}  // namespace ftrl_dev
