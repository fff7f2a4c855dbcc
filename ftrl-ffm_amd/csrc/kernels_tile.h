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
// 64 factors of one slot (k > 64).  Pipeline per wave, tile st (D = kTileDV):
//   partner weights of tile st -> LDS transposer
//   facts of tile st+D (registers, loaded a tile ago) -> LDS records {tmp_grad, x_own*x_other,
//                         flags, offset}
//   facts of tile st+D+1 requested (global -> registers)
//   partner weights of tile st+D requested at the offsets just staged (global -> registers): with
//                         the tiles st+1 .. st+D-1 that is 16 D touches of gathers in flight
//   fold of tile st
// The ORDER of the two requests matters: the memory counter (vmcnt) retires loads in issue order,
// so the facts the next tile's staging waits for must be requested BEFORE this tile's gathers --
// requested after them, every tile waited for the gathers it had just issued (a gather into a
// 247 GB tensor is a page walk and a trip to HBM).
// The loop is unrolled by D so that the D sets of weight registers keep their names (no moves).
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
                                               unsigned wave, unsigned n_waves, int lists, float *T,
                                               float4 *R) {
  constexpr int RS = 4 * NF;  // records per touch in R (>= slots per chunk)
  const int K = m.n_factors, F = m.n_fields, RL = m.row_len;
  const int lane = threadIdx.x & 63;
  const int tl = lane >> 2, cq = lane & 3;  // stager / loader layout: touch, 16-byte quad column
  const int SPC = K <= 64 ? 64 / K : 1;     // slots per chunk
  const int cps = K <= 64 ? 1 : (K + 63) >> 6;  // chunks per slot
  const int slots = record_span(m, 1);
  const unsigned per_feat = K <= 64 ? (slots + SPC - 1) / SPC : slots * cps;
  // the lists this launch takes (bits of `lists`: 1 big, 2 huge, 4 giant), longest first:
  // [giant | huge | big]
  const unsigned n_giant = (lists & 4) ? static_cast<unsigned>(s.counters[CNT_NGIANT]) : 0u;
  const unsigned n_huge = n_giant + ((lists & 2) ? static_cast<unsigned>(s.counters[CNT_NHUGE]) : 0u);
  const unsigned n_items = (n_huge + ((lists & 1) ? static_cast<unsigned>(s.counters[CNT_NBIG]) : 0u)) * per_feat;
  for (unsigned item = wave; item < n_items; item += n_waves) {
    const unsigned li = item / per_feat;
    const int ci = static_cast<int>(item - li * per_feat);
    const int u = wave_uniform(li < n_giant ? s.giant[li] : li < n_huge ? s.huge[li - n_giant] : s.big[li - n_huge]);
    const int4 ud = s.udesc[u];  // {feature, start, count, field}
    const int fa = wave_uniform(ud.w);
    const int i = wave_uniform(ud.x);
    const int start = wave_uniform(ud.y);
    const int t_lo = 0, c = wave_uniform(ud.z);  // the feature's occurrences [t_lo, c)
    const int sb = K <= 64 ? ci * SPC : ci / cps;                    // first slot of the chunk
    const int kk0 = K <= 64 ? 0 : (ci - sb * cps) * 64;              // first factor (k > 64)
    const int width = K <= 64 ? SPC * K : min(64, K - kk0);          // elements of the chunk
    const int fp0 = wave_uniform(walk_field(m, fa, sb));
    if (fp0 < 0) continue;  // (valid walk positions are a prefix: nothing stored from here on)
    // ---- fold layout: lane = element of the chunk (idle lanes repeat the first four) ----
    const bool inw = lane < width;
    const int le = inw ? lane : (lane & 3);
    const int es = K <= 64 ? le / K : 0;
    const int ekk = K <= 64 ? le - es * K : kk0 + le;
    const int fp = sb + es < slots ? walk_field(m, fa, sb + es) : -1;
    // (slots that one row touches twice are ffm_generic_body's: the row-order walk)
    const bool active = inw && fp >= 0 && !((s.cmask[start] >> fp) & 1ull);
    {  // no row of the block touches any slot of the chunk that is folded here
      const unsigned long long gm = s.gmask[start];
      if (!__any(active && ((gm >> fp) & 1ull))) continue;
    }
    // issue priority by length: a wave that shares its SIMD gets a fraction of the issue slots, and
    // the longest folds bound the phase (FFM_TILE_PRIO=0: off)
#ifndef FFM_TILE_PRIO
#define FFM_TILE_PRIO 1
#endif
    if (FFM_TILE_PRIO) {
      if (c > 640) __builtin_amdgcn_s_setprio(3);
      else if (c > 192) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
    float *rec = lat_row(m, i, fa) + (fp >= 0 ? sb + es : sb) * K + ekk;
    float n = rec[LAT_N * RL], z = rec[LAT_Z * RL];
    const float w = rec[LAT_W * RL];
    Fold acc;
    acc.init(n);
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
    // facts of tile st -> LDS records (touches whose partner field holds several entries in the row
    // -- HF_CHAIN -- only occur on serial slots, which are not folded here: staged dead)
    auto stage_facts = [&](int st, const TileFacts &f) {
      const bool in_range = t_lo + st * kTileT + tl < c;
      float4 *Rb = R + (st & (kTileNR - 1)) * (kTileT * RS) + tl * RS;
#pragma unroll
      for (int j = 0; j < NF; j++) {
        const int sj = cq + 4 * j;
        if (sj >= SPC) continue;
        int fl = f.ax[j].y & 0xff;
        const int own_field = f.ax[j].y >> 8;
        if (!in_range || !okS[j]) fl = 0;
        if (m.h.learn && (fl & HF_SIMPLE)) fl |= HF_FIRST;  // the variant uses g2*g2 at ffm.cpp:118
        const float x = f.mt.y * __int_as_float(f.ax[j].x);
        Rb[sj] = make_float4(f.mt.x, x, __int_as_float(tile_hiword(f.ax[j].w, fl, own_field)),
                             __int_as_float(f.ax[j].z));
      }
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
    // the fold of tile st (lane = element): kTileG touches per group, one range vote per group
    auto apply_tile = [&](int st) {
      if (st > 0 && (st * kTileT) % kSeg == 0) acc.flush();  // a segment of kSeg occurrences ends
      const float4 *Rc = R + (st & (kTileNR - 1)) * (kTileT * RS) + es;
      const float *Tc = T + lane;
      const int cnt = min(kTileT, c - t_lo - st * kTileT);  // live touches of this tile
      for (int g0 = 0; g0 < cnt; g0 += kTileG) {
        float tg[kTileG], x[kTileG], vp[kTileG];
        bool live[kTileG], first[kTileG];
#pragma unroll
        for (int j = 0; j < kTileG; j++) {  // (records past the end of the feature are staged dead)
          const float4 rc = Rc[(g0 + j) * RS];
          const int hw = __float_as_int(rc.z);
          tg[j] = rc.x;
          x[j] = rc.y;
          live[j] = active && (hw & (HF_SIMPLE << 8)) != 0;
          first[j] = (hw & (HF_FIRST << 8)) != 0;
          vp[j] = Tc[(g0 + j) * kTileRow];
        }
        fold_ffm_group<kTileG>(m.h, acc, w, live, first, tg, x, vp);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    {
      static_assert(kTileDV >= 1 && kTileDV <= 3 && kTileDV + 1 <= kTileNR, "weights of 1 .. 3 tiles in flight");
      static_assert(kSeg % kTileT == 0 && kTileT % kTileG == 0, "segments are whole tiles, tiles whole groups");
      TileFacts fN;
      TileWeights V0, V1, V2;
      {
        TileFacts f0, f1, f2;
        load_facts(0, f0);
        if (kTileDV > 1 && steps > 1) load_facts(1, f1);
        if (kTileDV > 2 && steps > 2) load_facts(2, f2);
        if (steps > kTileDV) load_facts(kTileDV, fN);
        stage_facts(0, f0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        V0 = issue_weights(0);
        if (kTileDV > 1 && steps > 1) {
          stage_facts(1, f1);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          V1 = issue_weights(1);
        }
        if (kTileDV > 2 && steps > 2) {
          stage_facts(2, f2);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          V2 = issue_weights(2);
        }
      }
#define FTRL_TILE_STEP(ST, V)                                                      \
      if ((ST) < steps) {                                                          \
        transpose(V);                                                              \
        const bool more__ = (ST) + kTileDV < steps;                                \
        if (more__) {                                                              \
          stage_facts((ST) + kTileDV, fN);                                         \
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
    const bool touched = fold_finish_latent(m.h, acc, w, n, z);
    if (touched && active) {
      rec[LAT_N * RL] = n;
      rec[LAT_Z * RL] = z;
    }
  }
  if (FFM_TILE_PRIO) __builtin_amdgcn_s_setprio(0);
}

// The whole FFM update of a block in ONE launch on the main stream.  Workgroup ranges:
//   [bias fold + linear update | hot features (giant, huge, big lists: tiles) | few-occurrence
//    features | serial slots (the row-order walk) | loss sum]
// There are no long dependent chains left in the phase, so nothing needs a stream (and a hardware
// queue hop, ~44 us for a fork + join) of its own.
// ns_flat: the few-occurrence range runs over a flat (feature, vector) space instead (compact shards).
template <int NF>
__global__ __launch_bounds__(kUpdThreads) void ffm_update_all_kernel(ModelDev m, Rows rows, Scratch s,
                                                                     int side_blocks, int nt, int ns, int few_only,
                                                                     int nw, int loss_blocks, double *loss_out,
                                                                     double *loss_scratch) {
  __shared__ __attribute__((aligned(16))) float lds_T[kUpdWaves][kTileT * kTileRow];
  __shared__ float4 lds_R[kUpdWaves][kTileNR * kTileT * 4 * NF];
  int r = blockIdx.x;
  if (r < side_blocks) {
    if (r == 0) bias_update_body(m, rows.n_rows, s);
    else linear_update_body(m, rows, s, r - 1, side_blocks - 1);
    return;
  }
  r -= side_blocks;
  if (r < nt) {
    const unsigned wv = wave_uniform(threadIdx.x >> 6);
    ffm_tile_items<NF>(m, rows, s, r * kUpdWaves + wv, nt * kUpdWaves, 7, lds_T[wv], lds_R[wv]);
    return;
  }
  r -= nt;
  if (r < ns) { ffm_small_body(m, rows, s, few_only, r, ns); return; }
  r -= ns;
  if (r < nw) { ffm_generic_body(m, rows, s, 1, r, nw); return; }
  loss_sum_body(rows.n_rows, s.loss, loss_out, loss_scratch, r - nw, loss_blocks);
}

}  // namespace ftrl_dev
