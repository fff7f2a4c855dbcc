// kernels_tile.h -- the FTRL (n, z) update of HOT features: those that occur in more than kSmallMax
// rows of the block (FFM::update_vector_nz src/model/ffm.cpp:90-136 incl. :118, applied to one
// element by tens to thousands of rows), as a fold (kernels_fold.h): per element two running sums
// (g, g*g) over the feature's touches in row order, per-touch step sizes only from the element's
// first ffm.cpp:118 touch on, one square-root pair at the end.
//
// The fold is cheapest with lane = element (plain adds in one lane, no cross-lane traffic), but the
// operands of a touch (its fact, then -- at the address the fact carries -- the partner's weights)
// are two dependent gathers that are cheapest touch-parallel: one load instruction fetches them for
// sixteen touches, so two instructions in flight cover 32 touches of latency.  So a wave works on
// TILES of 16 touches x 64 elements: the stager half of the loop loads a tile's facts and partner
// weights with lane = (touch, 16-byte quad), writes them to the wave's own LDS region, and the fold
// half reads them back with lane = element ("transposed through LDS").
//
// Work item = (hot feature, chunk of its stored record): a chunk is 64 / k whole slots (k <= 64) or
// 64 factors of one slot (k > 64).  Pipeline per wave, tile st:
//   partner weights of tile st -> LDS transposer
//   facts of tile st+1 (registers, loaded a tile ago) -> LDS records {tmp_grad, x_own*x_other,
//                         flags, offset}
//   facts of tile st+2 requested (row table -> registers) at the {entry, row} loaded a tile ago
//   {entry, row} of the touches of tile st+3 requested
//   partner weights of tile st+1 requested at the offsets just staged (global -> registers)
//   fold of tile st
// The ORDER of the requests matters: the memory counter (vmcnt) retires loads in issue order, so
// what the next tile's staging waits for must be requested BEFORE this tile's gathers -- requested
// after them, every tile waited for the gathers it had just issued (a gather into a 247 GB tensor
// is a page walk and a trip to HBM).
// Slots that one row touches twice (s.cmask) are left to the row-order walk (ffm_generic_body).
#pragma once
#include "engine_types.h"
#include "kernels_touch.h"
#include "kernels_update.h"
#include "kernels_fold.h"

namespace ftrl_dev {

constexpr int kTileT = 16;    // touches per tile
constexpr int kTileRow = 80;  // floats per transposer row: 64 + 16 of padding (16-byte writes of
                              // consecutive touches land 16 banks apart)
constexpr int kTileNR = 2;  // fact-record buffers (a power of two): the tile in flight and the one being
                            // applied (k = 4: 4 KB each per wave).  One tile of partner weights in flight:
                            // two and three measured slower (profiles/archive/r05_experiments.md section 2)
#ifndef FFM_TILE_G
#define FFM_TILE_G 4
#endif
#ifndef FFM_FOLD_REGULAR
#define FFM_FOLD_REGULAR 1
#endif
constexpr int kTileG = FFM_TILE_G;  // touches per arithmetic group (one range vote per group)

// flags word of an LDS fact record: bits 0..7 high bits of the weights' offset, bit 31 the touch is
// live (exactly one entry of the partner field in the row), bit 30 live and the own entry is the pair's first --
// so that the fold reads them with one compare each (hw < 0, hw >= 0xC0000000 unsigned)
__device__ __forceinline__ int tile_hiword(int off_hi, bool live, bool first) {
  return (off_hi & 0xff) | (live ? static_cast<int>(0x80000000u) : 0) | (live && first ? 0x40000000 : 0);
}

struct TileWeights {  // a loader lane's four 16-byte quads of one tile's partner weights
  float4 q0, q1, q2, q3;
};
// The stager takes a touch's facts from the row's per-field table (s.rowtab, 16 B per (row, field): 5 MB
// per 8192 x 39 block, read by every entry of the row: it stays on-die) through the occurrence's
// {entry, row} (s.occ2).  (Rounds 3-4 had the row kernel write an occurrence-ordered fact stream per
// (hot entry, field): 104 MB per block to HBM and back.)
struct TileFacts {  // what a stager lane holds of one tile: NF facts of its touch and the touch's own entry
  int4 ax[4];
  float2 mt;   // {tmp_grad, own value}
  int p, fm;   // own entry, its field
};

// Geometry of the (feature, chunk) work items.
struct TileGeom {
  int K, F, RL, SPC, cps, slots;
  unsigned per_feat;  // chunks per stored record
};
__device__ __forceinline__ TileGeom tile_geom(const ModelDev &m) {
  TileGeom g;
  g.K = m.n_factors;
  g.F = m.n_fields;
  g.RL = m.row_len;
  g.SPC = g.K <= 64 ? 64 / g.K : 1;          // slots per chunk
  g.cps = g.K <= 64 ? 1 : (g.K + 63) >> 6;   // chunks per slot
  g.slots = record_span(m, 1);
  g.per_feat = g.K <= 64 ? (g.slots + g.SPC - 1) / g.SPC : g.slots * g.cps;
  return g;
}
// Chunk ci of the stored record of a feature whose field is fa, as this lane sees it in the fold
// layout (lane = element of the chunk; idle lanes repeat the first four).
struct TileChunk {
  int sb, kk0, width, fp0;  // first slot, first factor (k > 64), elements, partner field of the first slot
  int es, ekk, fp;          // this lane's slot within the chunk, factor, partner field (-1: none)
  bool inw;
};
__device__ __forceinline__ TileChunk tile_chunk(const ModelDev &m, const TileGeom &g, int fa, int ci) {
  const int lane = threadIdx.x & 63;
  TileChunk c;
  c.sb = g.K <= 64 ? ci * g.SPC : ci / g.cps;
  c.kk0 = g.K <= 64 ? 0 : (ci - c.sb * g.cps) * 64;
  c.width = g.K <= 64 ? g.SPC * g.K : min(64, g.K - c.kk0);
  c.fp0 = wave_uniform(walk_field(m, fa, c.sb));  // (< 0: valid walk positions are a prefix -- nothing stored here)
  c.inw = lane < c.width;
  const int le = c.inw ? lane : (lane & 3);
  c.es = g.K <= 64 ? le / g.K : 0;
  c.ekk = g.K <= 64 ? le - c.es * g.K : c.kk0 + le;
  c.fp = c.sb + c.es < g.slots ? walk_field(m, fa, c.sb + c.es) : -1;
  return c;
}

// Streams tiles of one hot feature's c occurrences (they start at `start` in the grouped order) for
// the chunk `ch` through the wave's LDS (see the header): the tiles tile0, tile0 + tstride, ... --
// `steps` of them; tile number T covers the occurrences [16 T, 16 T + 16), those at or beyond c are
// staged dead -- and hands every tile to apply(st, Rc, Tc): st = 0 .. steps-1, Rc = the tile's fact records as this
// lane's slot sees them (record of touch j at Rc[j * RS]: {tmp_grad, x_own * x_other, flags, ..}),
// Tc = this lane's column of the transposer (partner weight of touch j at Tc[j * kTileRow]).
// NF: facts a stager lane carries per tile = ceil(slots per chunk / 4): 1 for k >= 16, 2 for
// k = 8 / 12, 4 for k = 4.  T: the wave's transposer [kTileT][kTileRow]; R: its fact records
// [kTileNR][kTileT * 4 * NF].
template <int NF, typename Apply>
__device__ __forceinline__ void tile_stream(const ModelDev &m, const Scratch &s, const TileGeom &g,
                                            const TileChunk &ch, const Rows &rows, int fa, int start, int tile0,
                                            int tstride, int steps, int c, float *T, float4 *R, Apply &&apply) {
  constexpr int RS = 4 * NF;  // records per touch in R (>= slots per chunk)
  const int K = g.K, F = g.F;
  const int lane = threadIdx.x & 63;
  const int tl = lane >> 2, cq = lane & 3;  // stager / loader layout: touch, 16-byte quad column
  // ---- stager layout: lane (touch tl, column cq) carries the facts of slots cq + 4 j ----
  bool okS[NF];
  int fS[NF];
#pragma unroll
  for (int j = 0; j < NF; j++) {
    const int sj = cq + 4 * j;
    const int f = (sj < g.SPC && ch.sb + sj < g.slots) ? walk_field(m, fa, ch.sb + sj) : -1;
    okS[j] = f >= 0;
    fS[j] = okS[j] ? f : ch.fp0;
  }
  // ---- loader layout: lane (touch tl, column cq) fetches the quads 4 r + cq of the chunk ----
  int sQ[4], kkQ[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int e4 = 4 * (4 * r + cq);
    const bool inq = e4 < ch.width;
    sQ[r] = inq && K <= 64 ? e4 / K : 0;
    kkQ[r] = inq ? (K <= 64 ? e4 - sQ[r] * K : ch.kk0 + e4) : 0;
  }
  // {entry, row} of the stager lane's touch of tile st
  auto load_desc = [&](int st) {
    const int t = min((tile0 + st * tstride) * kTileT + tl, c - 1);  // past the end: repeats, staged as dead
    return s.occ2[start + t];
  };
  auto load_facts = [&](const int2 d, TileFacts &f) {
#pragma unroll
    for (int j = 0; j < NF; j++) f.ax[j] = s.rowtab[static_cast<int64_t>(d.y) * F + fS[j]];
    f.mt = make_float2(s.tg[d.y], rows.val[d.x]);
    f.p = d.x;
    f.fm = rows.field[d.x];
  };
  // facts of tile st -> LDS records {tmp_grad, x_own * x_other, flags | offset}: a touch is live
  // when the row holds exactly one entry of the slot's partner field, another than the own one, and
  // this shard owns the pair (several entries: only on serial slots, which are not folded here)
  auto stage_facts = [&](int st, const TileFacts &f) {
    const bool in_range = (tile0 + st * tstride) * kTileT + tl < c;
    float4 *Rb = R + (st & (kTileNR - 1)) * (kTileT * RS) + tl * RS;
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const int sj = cq + 4 * j;
      if (sj >= g.SPC) continue;
      const int4 rt = f.ax[j];  // {partner feature, its value, its entry (-1 none, -2 several), count}
      const bool live = in_range && okS[j] && rt.z >= 0 && rt.z != f.p && owns_pair(m, f.fm, fS[j]);
      const bool first = f.p < rt.z || m.h.learn;  // (the learning variant: g2*g2 at ffm.cpp:118)
      // (a dead touch gathers from the start of the tensor: any valid address)
      const int64_t off = live ? w_slot_offset(m, rt.x, fS[j], f.fm) : 0;
      const float x = f.mt.y * __int_as_float(rt.y);
      Rb[sj] = make_float4(f.mt.x, x, __int_as_float(tile_hiword(static_cast<int>(off >> 32), live, first)),
                           __int_as_float(static_cast<int>(off & 0xffffffff)));
    }
  };
  // (named members, handed over by value: as arrays behind references these sixteen registers
  // ended up in scratch memory, every load waited for at once)
  auto issue_weight = [&](const float4 *Rb, int r) {
    const float4 rc = Rb[sQ[r]];
    const int64_t off = fact_offset(__float_as_int(rc.w), __float_as_int(rc.z) & 0xff);
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
  auto apply_tile = [&](int st) {
    apply(st, R + (st & (kTileNR - 1)) * (kTileT * RS) + ch.es, T + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  };
  static_assert(kSeg % kTileT == 0 && kTileT % kTileG == 0, "segments are whole tiles, tiles whole groups");
  // One tile of weights in flight, facts one tile further, descriptors one more (deeper pipelines
  // measured slower: profiles/archive/r05_experiments.md).  The requests of a step in consumption order --
  // vmcnt retires loads in issue order: facts(ST+2), descriptors(ST+3), weights(ST+1).
  TileFacts fN;
  TileWeights V0;
  int2 dN = make_int2(0, 0);
  {
    TileFacts f0;
    const int2 d0 = load_desc(0);
    const int2 d1 = steps > 1 ? load_desc(1) : d0;
    if (steps > 2) dN = load_desc(2);
    load_facts(d0, f0);
    if (steps > 1) load_facts(d1, fN);
    stage_facts(0, f0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    V0 = issue_weights(0);
  }
  for (int st = 0; st < steps; st++) {
    transpose(V0);
    const bool more = st + 1 < steps;
    if (more) {
      stage_facts(st + 1, fN);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (st + 2 < steps) load_facts(dN, fN);
    if (st + 3 < steps) dN = load_desc(st + 3);
    if (more) V0 = issue_weights(st + 1);
    apply_tile(st);
  }
}

// The touches of one tile as the fold sees them (lane = element): kTileG at a time.
#define FTRL_TILE_READ_GROUP(Rc, Tc, g0, active)                                         \
  float tg[kTileG], x[kTileG], vp[kTileG];                                               \
  bool live[kTileG], first[kTileG];                                                      \
  _Pragma("unroll") for (int j = 0; j < kTileG; j++) {                                   \
    const float4 rc = (Rc)[((g0) + j) * RS]; /* (records past the end are staged dead) */ \
    const int hw = __float_as_int(rc.z);                                                 \
    tg[j] = rc.x;                                                                        \
    x[j] = rc.y;                                                                         \
    live[j] = (active) && hw < 0;                                                        \
    first[j] = (active) && static_cast<unsigned>(hw) >= 0xC0000000u;                     \
    vp[j] = (Tc)[((g0) + j) * kTileRow];                                                 \
  }

// ... of a regular block (kernels_fold.h: fold_ffm_group_regular): no flags to decode
#define FTRL_TILE_READ_GROUP_REG(Rc, Tc, g0)                                             \
  float tg[kTileG], x[kTileG], vp[kTileG];                                               \
  _Pragma("unroll") for (int j = 0; j < kTileG; j++) {                                   \
    const float2 rc = *reinterpret_cast<const float2 *>(&(Rc)[((g0) + j) * RS]);         \
    tg[j] = rc.x;                                                                        \
    x[j] = rc.y;                                                                         \
    vp[j] = (Tc)[((g0) + j) * kTileRow];                                                 \
  }
// Is the block regular (see there)?  One scalar load per workgroup range.
__device__ __forceinline__ bool tile_block_regular(const ModelDev &m, const Scratch &s) {
#if FFM_FOLD_REGULAR
  return m.sort_start != nullptr && m.n_shards == 1 && m.own_n == nullptr && s.counters[CNT_IRREGULAR] == 0;
#else
  return false;
#endif
}

// ---- hot features below giant_min occurrences: one wave folds (feature, chunk) whole --------------
template <int NF>
__device__ __forceinline__ void ffm_tile_items(const ModelDev &m, const Rows &rows, const Scratch &s, unsigned wave,
                                               unsigned n_waves, float *T, float4 *R) {
  constexpr int RS = 4 * NF;
  const TileGeom g = tile_geom(m);
  // longest first: [huge | big]
  const unsigned n_huge = static_cast<unsigned>(s.counters[CNT_NHUGE]);
  const unsigned n_items = (n_huge + static_cast<unsigned>(s.counters[CNT_NBIG])) * g.per_feat;
  // (an item is a chain of dependent loads -- list, descriptor, masks / record / touch descriptors,
  // row table, partner weights -- and most items are one or two tiles long: the list entry is
  // requested two items ahead and the descriptor one, in scalar registers)
  auto list_at = [&](unsigned item) {
    const unsigned li = item / g.per_feat;
    return wave_uniform(li < n_huge ? s.huge[li] : s.big[li - n_huge]);
  };
  auto uniform4 = [](int4 v) { return make_int4(wave_uniform(v.x), wave_uniform(v.y), wave_uniform(v.z), wave_uniform(v.w)); };
  const bool regular = tile_block_regular(m, s);
  int4 ud_next = make_int4(0, 0, 1, 0);
  int u_next2 = 0;
  if (wave < n_items) ud_next = uniform4(s.udesc[list_at(wave)]);
  if (wave + n_waves < n_items) u_next2 = list_at(wave + n_waves);
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const int ci = static_cast<int>(item % g.per_feat);
    const int4 ud = ud_next;  // {feature, start, count, field}
    if (item + n_waves < n_items) ud_next = uniform4(s.udesc[u_next2]);
    if (item + 2 * n_waves < n_items) u_next2 = list_at(item + 2 * n_waves);
    const int fa = ud.w, i = ud.x;
    const int start = ud.y, c = ud.z;
    const TileChunk ch = tile_chunk(m, g, fa, ci);
    if (ch.fp0 < 0) continue;
    // (slots that one row touches twice are ffm_generic_body's: the row-order walk)
    const bool active = ch.inw && ch.fp >= 0 && !((s.cmask[start] >> ch.fp) & 1ull);
    // no row of the block touches any slot of the chunk that is folded here
    if (!__any(active && ((s.gmask[start] >> ch.fp) & 1ull))) continue;
    float *rec = lat_row(m, i, fa) + (ch.fp >= 0 ? ch.sb + ch.es : ch.sb) * g.K + ch.ekk;
    float n = rec[LAT_N * g.RL], z = rec[LAT_Z * g.RL];
    const float w = rec[LAT_W * g.RL];
    Fold acc;
    acc.init(n);
    // (a regular block: the touches' flags are the lane's)
    const bool lv = active && ch.fp != fa, q118 = lv && !m.h.learn && ch.fp < fa;
    tile_stream<NF>(m, s, g, ch, rows, fa, start, 0, 1, (c + kTileT - 1) / kTileT, c, T, R,
                    [&](int st, const float4 *Rc, const float *Tc) {
      if (st > 0) acc.flush();  // a segment (= a tile of kSeg occurrences) ends
      const int cnt = min(kTileT, c - st * kTileT);
      int g0 = 0;
      if (regular)
        for (; g0 + kTileG <= cnt; g0 += kTileG) {
          FTRL_TILE_READ_GROUP_REG(Rc, Tc, g0)
          fold_ffm_group_regular<kTileG>(acc, w, lv, q118, tg, x, vp);
        }
      for (; g0 < cnt; g0 += kTileG) {
        FTRL_TILE_READ_GROUP(Rc, Tc, g0, active)
        fold_ffm_group<kTileG>(acc, w, live, first, tg, x, vp);
      }
    });
    if (fold_finish_latent(m.h, acc, w, n, z) && active) {
      rec[LAT_N * g.RL] = n;
      rec[LAT_Z * g.RL] = z;
    }
  }
}

// ---- giant features (giant_min occurrences or more): the waves of a workgroup fold ONE (feature,
// chunk) together.  A segment is a tile, so the heavy work of a tile -- gathers, gradients, root
// differences -- needs from the tiles before it only n at its start (B_s) and whether the element has
// met a :118 touch yet; everything else it sums from -0.0f.  Super-step k: wave w takes tile
// k W + w;  (A) the tile's sums of g and g*g and its lane masks -> LDS;  barrier;  (B) B_s and the
// :118 flag from the running state plus the tiles of the super-step before this one, then the
// tile's root differences -> LDS;  barrier;  every wave joins the W tiles, in order, to its copy of
// the running state.  Same tree as one wave walking the tiles: bit-identical.
template <int W>
struct CoopLds {
  float P[2][W][64], G[2][W][64], D[2][W][64];
  unsigned long long fl[2][W][3];  // lanes with a live touch / whose first live touch is plain / with a :118 touch
  float ncap[64];
};
template <int NF, int W>
__device__ __forceinline__ void ffm_coop_items(const ModelDev &m, const Rows &rows, const Scratch &s, unsigned bidx,
                                               unsigned gdim, float *T, float4 *R, CoopLds<W> &cl) {
  constexpr int RS = 4 * NF;
  const TileGeom g = tile_geom(m);
  const int lane = threadIdx.x & 63;
  const int wv = wave_uniform(threadIdx.x >> 6);
  const unsigned n_items = static_cast<unsigned>(s.counters[CNT_NGIANT]) * g.per_feat;
  const bool regular = tile_block_regular(m, s);
  for (unsigned item = bidx; item < n_items; item += gdim) {  // (the whole workgroup on one item)
    const unsigned li = item / g.per_feat;
    const int ci = static_cast<int>(item - li * g.per_feat);
    const int4 ud = s.udesc[wave_uniform(s.giant[li])];  // {feature, start, count, field}
    const int fa = wave_uniform(ud.w), i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (c >= m.super_min) continue;  // folded as ranges all over the chip (ffm_range_items_a / _b)
    const TileChunk ch = tile_chunk(m, g, fa, ci);
    if (ch.fp0 < 0) continue;
    const bool active = ch.inw && ch.fp >= 0 && !((s.cmask[start] >> ch.fp) & 1ull);
    if (!__any(active && ((s.gmask[start] >> ch.fp) & 1ull))) continue;
    float *rec = lat_row(m, i, fa) + (ch.fp >= 0 ? ch.sb + ch.es : ch.sb) * g.K + ch.ekk;
    float n = rec[LAT_N * g.RL], z = rec[LAT_Z * g.RL];
    const float w = rec[LAT_W * g.RL];
    Fold run;  // the running state, the same in every wave
    run.init(n);
    // (a regular block: the touches' flags are the lane's -- kernels_fold.h: fold_ffm_group_regular)
    const bool lv = active && ch.fp != fa, q118 = lv && !m.h.learn && ch.fp < fa;
    const int n_tiles = (c + kTileT - 1) / kTileT, n_super = (n_tiles + W - 1) / W;
    tile_stream<NF>(m, s, g, ch, rows, fa, start, wv, W, n_super, c, T, R, [&](int k, const float4 *Rc, const float *Tc) {
      const int buf = k & 1;
      const int cnt = max(0, min(kTileT, c - (k * W + wv) * kTileT));
      // (A) the tile's plain sums and masks
      {
        float P = -0.0f, G = -0.0f;
        bool any = false, hp = false, q = false;
        int g0 = 0;
        if (regular) {
          for (; g0 + kTileG <= cnt; g0 += kTileG) {
            FTRL_TILE_READ_GROUP_REG(Rc, Tc, g0)
#pragma unroll
            for (int j = 0; j < kTileG; j++) {
              const float gj = tg[j] * vp[j] * x[j];
              G = G + gj;
              P = P + gj * gj;
            }
          }
          if (g0 > 0) { any = lv; hp = lv && !q118; q = q118; }
        }
        for (; g0 < cnt; g0 += kTileG) {
          FTRL_TILE_READ_GROUP(Rc, Tc, g0, active)
#pragma unroll
          for (int j = 0; j < kTileG; j++) {
            const float gj = tg[j] * vp[j] * x[j];
            G = live[j] ? G + gj : G;
            P = live[j] ? P + gj * gj : P;
            if (!any) hp = first[j];
            any = any || live[j];
            q = q || (live[j] && !first[j]);
          }
        }
        cl.P[buf][wv][lane] = P;
        cl.G[buf][wv][lane] = G;
        const unsigned long long ma = __ballot(any), mh = __ballot(any && hp), mq = __ballot(q);
        if (lane == 0) { cl.fl[buf][wv][0] = ma; cl.fl[buf][wv][1] = mh; cl.fl[buf][wv][2] = mq; }
      }
      __syncthreads();
      // (B) this tile's start from the running state and the super-step's earlier tiles -- only when
      // some element of the chunk has met a :118 touch by the end of this super-step (every wave
      // sees the same masks and running state: a uniform decision); else the root differences
      // telescope and D stays -0.0f
      bool quirky = run.seen;
#pragma unroll
      for (int w2 = 0; w2 < W; w2++) quirky = quirky || ((cl.fl[buf][w2][2] >> lane) & 1ull);
      if (!__any(quirky)) {
        cl.D[buf][wv][lane] = -0.0f;
      } else {
        Fold acc;
        acc.init(run.B);
        bool seen0 = run.seen;
        for (int w2 = 0; w2 < wv; w2++) {
          acc.B = acc.B + cl.P[buf][w2][lane];
          seen0 = seen0 || ((cl.fl[buf][w2][2] >> lane) & 1ull);
        }
        acc.seen = seen0;
        int g0 = 0;
        if (regular)
          for (; g0 + kTileG <= cnt; g0 += kTileG) {
            FTRL_TILE_READ_GROUP_REG(Rc, Tc, g0)
            fold_ffm_group_regular<kTileG>(acc, w, lv, q118, tg, x, vp);
          }
        for (; g0 < cnt; g0 += kTileG) {
          FTRL_TILE_READ_GROUP(Rc, Tc, g0, active)
          fold_ffm_group<kTileG>(acc, w, live, first, tg, x, vp);
        }
        cl.D[buf][wv][lane] = acc.D;
        if (acc.seen && !seen0) cl.ncap[lane] = acc.ncap;  // (an element's first :118 touch: once per item)
      }
      __syncthreads();
      // the W tiles of the super-step join the running state, in order
#pragma unroll
      for (int w2 = 0; w2 < W; w2++) {
        const bool any_s = (cl.fl[buf][w2][0] >> lane) & 1ull, hp_s = (cl.fl[buf][w2][1] >> lane) & 1ull;
        run.P = cl.P[buf][w2][lane];
        run.G = cl.G[buf][w2][lane];
        run.D = cl.D[buf][w2][lane];
        if (!run.any) run.head_plain = hp_s;
        run.any = run.any || any_s;
        run.seen = run.seen || ((cl.fl[buf][w2][2] >> lane) & 1ull);
        run.flush();
      }
    });
    __syncthreads();
    if (wv == 0) {
      run.ncap = run.seen ? cl.ncap[lane] : 0.0f;
      if (fold_finish_latent(m.h, run, w, n, z) && active) {
        rec[LAT_N * g.RL] = n;
        rec[LAT_Z * g.RL] = z;
      }
    }
  }
}
// ---- the longest features (super_min occurrences or more): ranges of kRange occurrences folded by
// waves all over the chip, through global partial sums per segment (= tile) ---------------------------
__device__ __forceinline__ int64_t seg_elem(const TileGeom &g, int seg, int e) {
  return static_cast<int64_t>(seg) * g.RL + e;
}
__device__ __forceinline__ unsigned long long *seg_flags(const Scratch &s, const TileGeom &g, int seg, int ci) {
  return s.segF + (static_cast<int64_t>(seg) * g.per_feat + ci) * 3;
}
// Pass A, work item = (range, chunk): per tile the sums of g and g*g from -0.0f and three lane masks
// (live touch seen / first live touch plain / :118 touch seen) -> s.segP / segG / segF.
template <int NF>
__device__ __forceinline__ void ffm_range_items_a(const ModelDev &m, const Rows &rows, const Scratch &s, unsigned wave,
                                                  unsigned n_waves, float *T, float4 *R) {
  constexpr int RS = 4 * NF;
  const TileGeom g = tile_geom(m);
  const int lane = threadIdx.x & 63;
  const unsigned n_items = static_cast<unsigned>(s.counters[CNT_NRANGE]) * g.per_feat;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned ri = item / g.per_feat;
    const int ci = static_cast<int>(item - ri * g.per_feat);
    const int2 gr = s.grange[ri];  // {index into giant, range number}
    const int gi = wave_uniform(gr.x), r = wave_uniform(gr.y);
    const int4 ud = s.udesc[wave_uniform(s.giant[gi])];
    const int fa = wave_uniform(ud.w), start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (c < m.super_min) continue;  // a workgroup folds it together (ffm_coop_items)
    const int seg0 = wave_uniform(s.gseg[gi]) + r * kRangeSegs;
    const TileChunk ch = tile_chunk(m, g, fa, ci);
    if (ch.fp0 < 0) continue;
    const bool active = ch.inw && ch.fp >= 0 && !((s.cmask[start] >> ch.fp) & 1ull);
    const bool has = ch.inw && ch.sb + ch.es < g.slots;  // this lane is an element of the stored record
    const int e = has ? (ch.sb + ch.es) * g.K + ch.ekk : 0;
    const int t_lo = r * kRange, t_hi = min(c, t_lo + kRange);
    const int steps = (t_hi - t_lo + kTileT - 1) / kTileT;
    if (!__any(active && ((s.gmask[start] >> ch.fp) & 1ull))) {
      // untouched chunk: pass B and the join still read its tiles' masks
      for (int st = 0; st < steps; st++)
        if (lane < 3) seg_flags(s, g, seg0 + st, ci)[lane] = 0ull;
      continue;
    }
    tile_stream<NF>(m, s, g, ch, rows, fa, start, t_lo / kTileT, 1, steps, c, T, R,
                    [&](int st, const float4 *Rc, const float *Tc) {
      const int cnt = min(kTileT, t_hi - t_lo - st * kTileT);
      float P = -0.0f, G = -0.0f;
      bool any = false, hp = false, q = false;
      for (int g0 = 0; g0 < cnt; g0 += kTileG) {
        FTRL_TILE_READ_GROUP(Rc, Tc, g0, active)
#pragma unroll
        for (int j = 0; j < kTileG; j++) {
          const float gj = tg[j] * vp[j] * x[j];
          G = live[j] ? G + gj : G;
          P = live[j] ? P + gj * gj : P;
          if (!any) hp = first[j];
          any = any || live[j];
          q = q || (live[j] && !first[j]);
        }
      }
      if (has) {
        s.segP[seg_elem(g, seg0 + st, e)] = P;
        s.segG[seg_elem(g, seg0 + st, e)] = G;
      }
      const unsigned long long ma = __ballot(any), mh = __ballot(any && hp), mq = __ballot(q);
      if (lane == 0) {
        unsigned long long *fl = seg_flags(s, g, seg0 + st, ci);
        fl[0] = ma; fl[1] = mh; fl[2] = mq;
      }
    });
  }
}
// Pass B (second launch), same work items, only where the chunk has an element with a :118 touch: n
// at the start of the range from the tiles before it (joined left to right), then the range's tiles
// again for their root differences -> s.segD; the n_t at an element's first :118 touch -> s.gcap.
template <int NF>
__device__ __forceinline__ void ffm_range_items_b(const ModelDev &m, const Rows &rows, const Scratch &s, unsigned wave,
                                                  unsigned n_waves, float *T, float4 *R) {
  constexpr int RS = 4 * NF;
  constexpr int kFly = 8;  // tiles whose sums are in flight together
  const TileGeom g = tile_geom(m);
  const int lane = threadIdx.x & 63;
  const unsigned n_items = static_cast<unsigned>(s.counters[CNT_NRANGE]) * g.per_feat;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned ri = item / g.per_feat;
    const int ci = static_cast<int>(item - ri * g.per_feat);
    const int2 gr = s.grange[ri];
    const int gi = wave_uniform(gr.x), r = wave_uniform(gr.y);
    const int4 ud = s.udesc[wave_uniform(s.giant[gi])];
    const int fa = wave_uniform(ud.w), i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (c < m.super_min) continue;
    const int segb = wave_uniform(s.gseg[gi]), n_seg = (c + kSeg - 1) / kSeg, seg_lo = r * kRangeSegs;
    const TileChunk ch = tile_chunk(m, g, fa, ci);
    if (ch.fp0 < 0) continue;
    // which lanes meet a :118 touch anywhere in the feature / before this range
    unsigned long long q_all = 0ull, q_before = 0ull;
    for (int s0 = 0; s0 < n_seg; s0 += kFly) {
      unsigned long long mq[kFly];
#pragma unroll
      for (int j = 0; j < kFly; j++) mq[j] = seg_flags(s, g, segb + min(s0 + j, n_seg - 1), ci)[2];
#pragma unroll
      for (int j = 0; j < kFly; j++) {
        q_all |= mq[j];
        if (s0 + j < seg_lo) q_before |= mq[j];
      }
    }
    if (q_all == 0ull) continue;  // every touch of the chunk is plain: the join telescopes
    const bool active = ch.inw && ch.fp >= 0 && !((s.cmask[start] >> ch.fp) & 1ull);
    const bool has = ch.inw && ch.sb + ch.es < g.slots;
    const int e = has ? (ch.sb + ch.es) * g.K + ch.ekk : 0;
    const int t_lo = r * kRange, t_hi = min(c, t_lo + kRange);
    const int steps = (t_hi - t_lo + kTileT - 1) / kTileT;
    const float *rec = lat_row(m, i, fa) + (ch.fp >= 0 ? ch.sb + ch.es : ch.sb) * g.K + ch.ekk;
    const float w = rec[LAT_W * g.RL];
    Fold acc;
    acc.init(rec[LAT_N * g.RL]);
    for (int s0 = 0; s0 < seg_lo; s0 += kFly) {  // B at the start of the range
      float p[kFly];
#pragma unroll
      for (int j = 0; j < kFly; j++) p[j] = s.segP[seg_elem(g, segb + min(s0 + j, seg_lo - 1), e)];
#pragma unroll
      for (int j = 0; j < kFly; j++)
        if (s0 + j < seg_lo) acc.B = acc.B + p[j];
    }
    const bool seen_before = (q_before >> lane) & 1ull;
    acc.seen = seen_before;
    tile_stream<NF>(m, s, g, ch, rows, fa, start, t_lo / kTileT, 1, steps, c, T, R,
                    [&](int st, const float4 *Rc, const float *Tc) {
      const int cnt = min(kTileT, t_hi - t_lo - st * kTileT);
      for (int g0 = 0; g0 < cnt; g0 += kTileG) {
        FTRL_TILE_READ_GROUP(Rc, Tc, g0, active)
        fold_ffm_group<kTileG>(acc, w, live, first, tg, x, vp);
      }
      if (has) s.segD[seg_elem(g, segb + seg_lo + st, e)] = acc.D;
      acc.flush();  // the segment (= tile) ends
    });
    if (has && acc.seen && !seen_before) s.gcap[static_cast<int64_t>(gi) * g.RL + e] = acc.ncap;
  }
}
// The join (third launch), work item = (feature, chunk), lane = element: the tiles' sums left to
// right, then the accumulator's (n_T, z_T) as for any other fold.
__device__ __forceinline__ void ffm_range_join(const ModelDev &m, const Scratch &s, unsigned wave, unsigned n_waves) {
  constexpr int kFly = 16;  // (a few latency-bound chains on an otherwise idle chip: loads in flight are all that counts)
  const TileGeom g = tile_geom(m);
  const int lane = threadIdx.x & 63;
  const unsigned n_items = static_cast<unsigned>(s.counters[CNT_NGIANT]) * g.per_feat;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned gi = item / g.per_feat;
    const int ci = static_cast<int>(item - gi * g.per_feat);
    const int4 ud = s.udesc[wave_uniform(s.giant[gi])];
    const int fa = wave_uniform(ud.w), i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y), c = wave_uniform(ud.z);
    if (c < m.super_min) continue;
    const int segb = wave_uniform(s.gseg[gi]), n_seg = (c + kSeg - 1) / kSeg;
    const TileChunk ch = tile_chunk(m, g, fa, ci);
    if (ch.fp0 < 0) continue;
    const bool active = ch.inw && ch.fp >= 0 && !((s.cmask[start] >> ch.fp) & 1ull);
    if (!__any(active && ((s.gmask[start] >> ch.fp) & 1ull))) continue;
    const int e = ch.inw && ch.sb + ch.es < g.slots ? (ch.sb + ch.es) * g.K + ch.ekk : 0;
    unsigned long long q_all = 0ull;
    for (int s0 = 0; s0 < n_seg; s0 += kFly) {
      unsigned long long mq[kFly];
#pragma unroll
      for (int j = 0; j < kFly; j++) mq[j] = seg_flags(s, g, segb + min(s0 + j, n_seg - 1), ci)[2];
#pragma unroll
      for (int j = 0; j < kFly; j++) q_all |= mq[j];
    }
    float *rec = lat_row(m, i, fa) + (ch.fp >= 0 ? ch.sb + ch.es : ch.sb) * g.K + ch.ekk;
    float n = rec[LAT_N * g.RL], z = rec[LAT_Z * g.RL];
    const float w = rec[LAT_W * g.RL];
    Fold acc;
    acc.init(n);
    for (int s0 = 0; s0 < n_seg; s0 += kFly) {
      float p[kFly], gsum[kFly], d[kFly];
      unsigned long long fa_[kFly], fh_[kFly];
#pragma unroll
      for (int j = 0; j < kFly; j++) {
        const int sg = segb + min(s0 + j, n_seg - 1);
        p[j] = s.segP[seg_elem(g, sg, e)];
        gsum[j] = s.segG[seg_elem(g, sg, e)];
        d[j] = q_all ? s.segD[seg_elem(g, sg, e)] : -0.0f;
        const unsigned long long *fl = seg_flags(s, g, sg, ci);
        fa_[j] = fl[0];
        fh_[j] = fl[1];
      }
#pragma unroll
      for (int j = 0; j < kFly; j++) {
        if (s0 + j >= n_seg) continue;
        const bool any_s = (fa_[j] >> lane) & 1ull, hp_s = (fh_[j] >> lane) & 1ull;
        acc.P = p[j];
        acc.G = gsum[j];
        acc.D = d[j];
        if (!acc.any) acc.head_plain = hp_s;
        acc.any = acc.any || any_s;
        acc.flush();
      }
    }
    acc.seen = (q_all >> lane) & 1ull;
    acc.ncap = acc.seen ? s.gcap[static_cast<int64_t>(gi) * g.RL + e] : 0.0f;
    if (fold_finish_latent(m.h, acc, w, n, z) && active) {
      rec[LAT_N * g.RL] = n;
      rec[LAT_Z * g.RL] = z;
    }
  }
}
#undef FTRL_TILE_READ_GROUP

// The whole FFM update of a block: ONE launch on the main stream.  Workgroup ranges:
//   [bias fold + linear update | giant features (a workgroup per item) | hot features (huge, big
//    lists: a wave per item) | few-occurrence features | serial slots (the row-order walk) | loss sum]
// There are no long dependent chains left in the phase, so nothing needs a stream (and a hardware
// queue hop, ~44 us for a fork + join) of its own.
// WAVES per workgroup: the waves that fold ONE giant feature's chunk together (every other range
// treats its waves as independent).  Four: with sixteen (one workgroup per CU, 1024-thread barriers,
// sixteen tiles joined per super-step) a super-step took three times as long and the giant range
// of a C5 block 506 us instead of 170 (profiles/archive/r05_experiments.md).
// Dynamic LDS: the transposers [WAVES][kTileT * kTileRow] floats, then the fact records.
constexpr int tile_waves(int nf) { return nf == 4 && FFM_TILE_WAVES > 4 ? 4 : nf == 2 && FFM_TILE_WAVES > 8 ? 8 : FFM_TILE_WAVES; }
constexpr size_t tile_lds_bytes(int nf, int waves) {
  return static_cast<size_t>(waves) * (kTileT * kTileRow * sizeof(float) + kTileNR * kTileT * 4 * nf * sizeof(float4));
}
constexpr size_t tile_lds_bytes(int nf) { return tile_lds_bytes(nf, tile_waves(nf)); }
// Small blocks (a 4096 x 8 block: ~1000-occurrence giants are 16 super-steps of four tiles) run the
// launch with EIGHT waves per workgroup: half the super-steps -- 80 -> 70 us per launch; a C5 block
// that way 438 -> 693 us (profiles/archive/r05_experiments.md).
constexpr int kWideWaves = 8;
// KINDS: which ranges this instantiation can run (their arguments must be 0 otherwise) -- the register
// allocation, and with it the waves per SIMD, of a kernel is that of its hungriest path.
enum { UPD_GIANT = 1, UPD_HOT = 2, UPD_FEW = 4, UPD_SIDE = 8, UPD_REST = 16, UPD_ALL = 31 };
// waves per SIMD the register allocation of an instantiation aims at (0: the compiler's choice)
#ifndef FFM_OCC_ALL
#define FFM_OCC_ALL 0
#endif
#ifndef FFM_OCC_HOT
#define FFM_OCC_HOT 4  // (137 registers by itself since the fold's range votes are branch-free: 3 waves; held to 128)
#endif
#ifndef FFM_OCC_FEW
#define FFM_OCC_FEW 0
#endif
#ifndef FFM_OCC_GIANT
#define FFM_OCC_GIANT 4  // (131 registers by itself: 3 waves)
#endif
// (k = 4 -- four facts per stager lane, sixteen slots per chunk -- does not fit 128 registers: held to
// four waves its hot and giant instantiations spill 39-46 VGPRs into 160-188 bytes of scratch per lane;
// at three waves C3's three launches side by side are 0.476 ms against 0.505, and against 0.488 as one launch)
constexpr int upd_occ(int kinds, int nf) {
  return kinds == UPD_ALL ? FFM_OCC_ALL
         : (kinds & UPD_GIANT) ? (nf == 4 ? 3 : FFM_OCC_GIANT)
         : (kinds & UPD_HOT) ? (nf == 4 ? 3 : FFM_OCC_HOT)
         : (kinds & UPD_FEW) ? FFM_OCC_FEW : 0;
}
constexpr int upd_occ_min(int kinds, int nf) { return upd_occ(kinds, nf) > 0 ? upd_occ(kinds, nf) : 1; }
constexpr int upd_occ_max(int kinds, int nf) { return upd_occ(kinds, nf) > 0 ? upd_occ(kinds, nf) : 8; }
#define FFM_UPD_OCC __attribute__((amdgpu_waves_per_eu(upd_occ_min(KINDS, NF), upd_occ_max(KINDS, NF))))
template <int NF, int KINDS = UPD_ALL, int WAVES = tile_waves(NF)>
__global__ __launch_bounds__(64 * WAVES) FFM_UPD_OCC void ffm_update_all_kernel(ModelDev m, Rows rows, Scratch s,
                                                                     int side_blocks, int ng, int nt, int ns,
                                                                     int few_only, int nw, int loss_blocks,
                                                                     double *loss_out, double *loss_scratch, int order) {
  extern __shared__ __attribute__((aligned(16))) char lds_dyn[];
  __shared__ CoopLds<WAVES> lds_coop;
  int r = blockIdx.x;
  const unsigned wv = wave_uniform(threadIdx.x >> 6);
  float *T = reinterpret_cast<float *>(lds_dyn) + wv * (kTileT * kTileRow);
  float4 *R = reinterpret_cast<float4 *>(lds_dyn + WAVES * kTileT * kTileRow * sizeof(float)) + wv * (kTileNR * kTileT * 4 * NF);
  if ((KINDS & UPD_SIDE) && r < side_blocks) {
    if (r == 0) bias_update_body(m, rows.n_rows, s, reinterpret_cast<float *>(lds_dyn));
    else linear_update_body(m, rows, s, r - 1, side_blocks - 1);
    return;
  }
  r -= side_blocks;
  // the three big ranges in the order `order` names (digits, first range last: 0 giant, 1 hot, 2 few);
  // which range first, then ONE call site per range (inside the loop the unrolled copies cost the
  // one-launch instantiations nine registers)
  int kind = -1;
  for (int o = order, left = 3; left > 0 && kind < 0; left--, o /= 10) {
    const int kd = o % 10;
    const int n = kd == 0 ? ng : kd == 1 ? nt : ns;
    if (r < n) kind = kd;
    else r -= n;
  }
  if (kind == 0) {
    if (KINDS & UPD_GIANT) ffm_coop_items<NF, WAVES>(m, rows, s, r, ng, T, R, lds_coop);
  } else if (kind == 1) {
    if (KINDS & UPD_HOT) {
      ffm_range_items_a<NF>(m, rows, s, r * WAVES + wv, nt * WAVES, T, R);
      ffm_tile_items<NF>(m, rows, s, r * WAVES + wv, nt * WAVES, T, R);
    }
  } else if (kind == 2) {
    if (KINDS & UPD_FEW) ffm_small_body(m, rows, s, few_only, r, ns);
  }
  if (kind >= 0) return;
  if (KINDS & UPD_REST) {
    if (r < nw) { ffm_generic_body(m, rows, s, 1, r, nw); return; }
    loss_sum_body(rows.n_rows, s.loss, loss_out, loss_scratch, r - nw, loss_blocks);
  }
}
// The longest features' pass B and their join (launched after ffm_update_all_kernel when the block is
// large enough to have any; both return at once when it has none).
template <int NF>
__global__ __launch_bounds__(kUpdThreads) void ffm_update_super_b_kernel(ModelDev m, Rows rows, Scratch s) {
  __shared__ __attribute__((aligned(16))) float lds_T[kUpdWaves][kTileT * kTileRow];
  __shared__ float4 lds_R[kUpdWaves][kTileNR * kTileT * 4 * NF];
  const unsigned wv = wave_uniform(threadIdx.x >> 6);
  ffm_range_items_b<NF>(m, rows, s, blockIdx.x * kUpdWaves + wv, gridDim.x * kUpdWaves, lds_T[wv], lds_R[wv]);
}
__global__ __launch_bounds__(kUpdThreads) void ffm_update_super_join_kernel(ModelDev m, Scratch s) {
  ffm_range_join(m, s, blockIdx.x * kUpdWaves + wave_uniform(threadIdx.x >> 6), gridDim.x * kUpdWaves);
}

}  // namespace ftrl_dev
