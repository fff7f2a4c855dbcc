// engine_types.h -- plain structs handed by value to the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "ftrl_math.h"

namespace ftrl_dev {

// One block of rows in CSR (the wire format of include/ffm_engine.h), device pointers.
struct Rows {
  int n_rows;
  int nnz;
  const int *row_ptr;
  const int *field;  // may be null (LR / FM: field 0)
  const int *feat;
  const float *val;
  const int *label;  // may be null for predict
};

// Model state in HBM.
//   bias3  = {bias, bias_n, bias_z}
//   lin_*  = [n_feats] each
//   lat    = [records][3][row_len] floats; component 0 = n, 1 = z, 2 = w.  One feature's
//            accumulators and weights are one contiguous 3*row_len*4-byte record (7488 B at
//            n_fields=39, n_factors=16), so a row's gather is nnz coalesced streams.
//
// Field-pair sharding (include/ffm_engine.h: n_shards / shard_rank, ffm_engine_shard_plan): this
// shard owns, for a feature whose field is fa, the latent slots of the partner fields
// [own_lo[fa], own_lo[fa] + own_n[fa]) -- one contiguous range per field, symmetric
// (fb in range(fa) <=> fa in range(fb)), so both slots of a pair live on one shard.
//   * With per-field id ranges (field_start: field(i) is a function of the id) the storage is
//     COMPACT: only features of fields with own_n > 0 have a record, and a record holds just the
//     owned slots (rec_slots = the widest range): about 1/n_shards of the tensor per shard.
//     Record index = rec_base[fa] + (i - field_start[fa]); slot s <-> partner field own_lo[fa] + s.
//   * Without id ranges every shard keeps full-length records of all features (slot = partner
//     field) and only the ownership test shards the work.
// An unsharded model is the second layout with every pair owned.
struct ModelDev {
  int type;  // FFM_MODEL_*
  int n_feats;
  int n_fields;
  int n_factors;
  int row_len;    // floats per component of one STORED record: rec_slots * n_factors (FM: n_factors)
  int rec_slots;  // slots per stored record: n_fields, or the widest owned range (compact)
  int n_shards, shard_rank;
  Hyper h;
  float *bias3;
  float *lin_n, *lin_z, *lin_w;
  float *lat;
  // ownmask[fa] = bit fp set when this shard owns the field pair {fa, fp} (all ones when the
  // model is not sharded); n_fields <= 64 only, else null
  const unsigned long long *ownmask;
  const int *own_lo, *own_n;  // [n_fields] owned partner-field range per own field; null: all pairs
  const int *field_start;     // [n_fields + 1] id range of every field: compact storage; else null
  const long long *rec_base;  // [n_fields] first stored record of a field (compact); -1: none
  const int *lin_own;         // [n_fields] 1 when this shard owns the field's linear terms; null: all
  int bias_own;               // 1 when this shard owns the bias
  int huge_min;               // occurrences per block above which a hot feature is listed as "very hot" (kHugeMin)
  int giant_min;              // ... from which it is listed as "giant" (kGiantMin): FFM -- a workgroup folds one
                              //     together; FM, and FFM from super_min on -- ranges folded all over the chip
  int super_min;              // FFM: occurrences from which a giant feature's ranges are folded side by side
  const int *sort_start;      // [n_fields + 1] the id ranges the grouping sorts by (kernels_sort.h), or null
  int range_len;              // occurrences per range of such a feature (a multiple of kSeg): FFM kRange,
                              //     FM kFmRange -- an FM touch is three dependent loads and a dozen instructions,
                              //     its ranges are short so that many waves share one feature
};

enum { LAT_N = 0, LAT_Z = 1, LAT_W = 2 };

// 16-byte loads / stores with the non-temporal hint (global_load / store ... nt): for record data
// that one owner streams through once per block, so that it does not push the shared w rows out
// of an XCD's 4 MB L2.
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_nt(const float4 *p) {
  const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_nt(float4 *p, float4 x) {
  const v4f_nt v = {x.x, x.y, x.z, x.w};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_nt *>(p));
}


// Field-pair ownership: both latent slots of a pair (i, field_j) and (j, field_i) belong to the
// shard that owns {field_i, field_j}.
__device__ __forceinline__ bool owns_pair(const ModelDev &m, int fa, int fb) {
  if (!m.own_n) return true;
  return static_cast<unsigned>(fb - m.own_lo[fa]) < static_cast<unsigned>(m.own_n[fa]);
}
// The same test for a FIXED partner field fp against varying own fields, without memory traffic
// inside a loop: own_bits = owner_bits(m, fp) once, then owns_bit(own_bits, fa) per touch.  (By
// symmetry "fa owns fp" == "fp owns fa" == bit fa of ownmask[fp]; sharding needs n_fields <= 64.)
__device__ __forceinline__ unsigned long long owner_bits(const ModelDev &m, int fp) {
  return m.own_n ? m.ownmask[fp] : ~0ull;
}
__device__ __forceinline__ bool owns_bit(unsigned long long own_bits, int fa) {
  return (own_bits >> (fa & 63)) & 1ull;
}
__device__ __forceinline__ bool owns_linear(const ModelDev &m, int fa) {
  return !m.lin_own || m.lin_own[fa] != 0;
}
// Does this shard look at entries of field fa at all?  (Compact shards drop the other columns.)
__device__ __forceinline__ bool keeps_field(const ModelDev &m, int fa) {
  return !m.field_start || m.own_n[fa] > 0 || m.lin_own[fa] != 0;
}

// The stored record of feature `feat` whose field is fa (fa is ignored unless storage is compact).
__device__ __forceinline__ float *lat_row(const ModelDev &m, int feat, int fa) {
  const long long rec = m.field_start ? m.rec_base[fa] + (feat - m.field_start[fa]) : feat;
  return m.lat + rec * 3 * m.row_len;
}
// Slot, inside the record of a feature of field fa, that holds partner field fb.
__device__ __forceinline__ int slot_of(const ModelDev &m, int fa, int fb) {
  return m.field_start ? fb - m.own_lo[fa] : fb;
}
// Float offset of the w row of that slot from the start of m.lat (what the fact stream carries).
__device__ __forceinline__ long long w_slot_offset(const ModelDev &m, int feat, int fa, int fb) {
  const long long rec = m.field_start ? m.rec_base[fa] + (feat - m.field_start[fa]) : feat;
  return rec * 3 * m.row_len + LAT_W * m.row_len + slot_of(m, fa, fb) * m.n_factors;
}

// A record as the feature-major kernels walk it: walk position c = storage slot c, for
// c < record_span.  walk_field gives the partner field at that position, or -1 when the record of
// a feature of field fa has no (owned) slot there.  Full-length records are walked whole (slot =
// partner field; the ownership test then filters touch by touch, which is also right when one id
// shows up under several fields); compact records hold exactly the owned range.
__device__ __forceinline__ int record_span(const ModelDev &m, int unit) { return m.rec_slots * unit; }
__device__ __forceinline__ int walk_field(const ModelDev &m, int fa, int c) {
  if (!m.field_start) return c < m.n_fields ? c : -1;
  return c < m.own_n[fa] ? m.own_lo[fa] + c : -1;
}

// Per-block scratch: the block grouped by feature ("postings"), per-row field chains, outputs.
struct Scratch {
  unsigned *key;  // [nnz] sort key per entry: feature id, n_feats for entries remove_out_range erases
  unsigned *skey; // [nnz] the keys sorted ascending (s.occ holds the entries in the same order)
  int *row_of;    // [nnz] row of each entry
  int *occ;       // [nnz] entry indices grouped by feature; ascending inside a group
  int2 *occ2;     // [nnz] the same groups as {entry, row of the entry}
  int4 *udesc;    // [nnz] the block's distinct features (arbitrary order), one 16-byte descriptor
                  //      each -- ONE load per owner instead of four dependent ones:
                  //      {feature id, start of its group in occ, number of occurrences, the field
                  //      its first occurrence carries (FFM)}
  int *small;     // [nnz] indices into uniq of the features with <= kSmallMax occurrences
  int4 *sdesc;    // [nnz] one descriptor per feature that occurs once: {feature, entry, row, field}
  int *few;       // [nnz] ... with 2..kSmallMax occurrences (the small list minus the features
                  //      that occur once, which ffm_update_single_kernel owns)
  int *big;       // [nnz] ... with kSmallMax < occurrences <= kHugeMin ("hot" features)
  int *huge;      // [nnz] ... with more, below ModelDev::giant_min ("very hot": present in many rows)
  int *giant;     // [nnz / kChainMin + 1] ... with giant_min or more.  FFM below super_min: a workgroup folds
                  //      one together (kernels_tile.h).  FM, and FFM from super_min on: their occurrences
                  //      are cut into RANGES of kRange that waves all over the chip fold side by side, the
                  //      segments' partial sums joined by a later launch; per giant feature:
  int *gseg;      // [nnz / kChainMin + 1] first of its segment slots in segP / segG / segD / segF
  int2 *grange;   // [nnz / kRange + nnz / giant_min + 2] {index into giant, range number}: one entry per range
  // partial sums of those folds, per segment and element of the stored record (shared by the scratch
  // sets: one block's update ends before the next one's starts)
  float *segP, *segG, *segD;    // [max_segs * row_len] sum g*g, sum g, sum of root differences (FFM)
  unsigned long long *segF;     // [max_segs * chunks * 3] FFM, per (segment, 64-element chunk): lanes with a
                                //      live touch / whose first live touch is plain / with a :118 touch
  float *gcap;                  // [(nnz / giant_min + 2) * row_len] FFM: n_t at an element's first :118 touch
  int *counters;  // [kNumCounters] CNT_* below
  int *n_super;   // [1] in page-locked HOST memory: how many features of the block have super_min
                  //     occurrences or more -- so that the host can skip their two extra launches
  int *err;       // [1] sticky ERR_* flags, shared by all sets: what ffm_engine_sync reports
  int *head;      // [n_rows*n_fields] first entry of each field in each row (FFM), -1 if none
  int *next;      // [nnz] next entry of the same row with the same field, -1 at the end
  int4 *rowtab;   // [n_rows*n_fields] {feat, val bits, entry, count} of the field's entry in the
                  //      row; entry = -1 none, -2 several (then head/next list them)
  // Hot features (more than kSmallMax occurrences in the block) have their touches' facts laid
  // out by occurrence position t (= index into occ), so their owners stream them:
  // Which slots of a feature's record does the block touch?  Slot (feature of entry p, field fp)
  // is touched when p's row holds another entry of field fp (ffm.cpp:72-88 refreshes exactly
  // those).  n_fields <= 64 only.
  unsigned long long *rowmask;  // [2*n_rows] per row: fields with >= 1 / >= 2 surviving entries
  unsigned long long *gmask;    // [nnz] per distinct feature, at index ustart[u]: bit fp set when
                                //      some row of the block touches slot fp of the feature
  unsigned long long *cmask;    // [nnz] the same for slots whose partner field is MULTI-VALUED in some
                                //      row of the feature (the row holds >= 2 entries of field fp)
  int *uflag;     // [nnz] per distinct feature, at index ustart[u]: UF_* bits (all model types)
  int *occpos;    // [nnz] entry -> its position t in occ when its feature is hot, else
                  //      OCC_FEW (2..kSmallMax occurrences) or OCC_ONCE (this entry only)
  float *logit;   // [n_rows] this shard's (partial) logit
  float *tg;      // [n_rows] tmp_grad = sigmoid(logit) - y
  double *loss;   // [n_rows] logloss per row
  float *svx;     // [n_rows*n_factors] FM per-row factor sums (sum_vx, fm.h:24)
};

// UF_DUP: the feature occurs twice (or more) in ONE row of the block -- the reference's running
// update inside a row is not a sum, so everything of that feature which the row touches keeps the
// row-order walk for the block (the checker under oracle/: mark_serial); for FFM the affected slots are in
// cmask as well.
enum { UF_DUP = 1 };

// The block update folds the touches of an accumulator by reductions over SEGMENTS of kSeg
// consecutive occurrences of its feature in the block (rows of the block for the bias), joined left
// to right -- the tree the checker under oracle/ (FO_SEG) restates; see kernels_fold.h.
constexpr int kSeg = 16;

// Every counter that takes atomics sits on a 64-byte line of its own (kLineInts apart):
// device-scope atomics are performed at the memory side, one line at a time, and the grouping's
// per-workgroup list reservations (seven lists, ~1250 workgroups) used to queue on ONE line.
constexpr int kLineInts = 16;
enum { CNT_NUNIQ = 0, CNT_CURSOR = 1 * kLineInts, CNT_NMULTI = 2 * kLineInts, CNT_ERROR = 3 * kLineInts,
       CNT_NSMALL = 4 * kLineInts, CNT_NBIG = 5 * kLineInts, CNT_NHUGE = 6 * kLineInts,
       CNT_NFEW = 7 * kLineInts, CNT_NSINGLE = 9 * kLineInts, CNT_NGIANT = 10 * kLineInts,
       CNT_SORT_BAR = 11 * kLineInts,  // (the sort's grid barrier, kernels_sort.h)
       CNT_NSEG = 12 * kLineInts, CNT_NRANGE = 13 * kLineInts,  // the giant features' segment slots / ranges
       CNT_IRREGULAR = 14 * kLineInts };  // set when the block is NOT "every row = one entry per field, in field
                                          // order, every id inside its field's range" (the range sort's short cut)
constexpr int kNumCounters = 15 * kLineInts;
enum { OCC_FEW = -1, OCC_ONCE = -2 };
// Occurrence classes of a block's hot features (more than kSmallMax occurrences):
//   big   (.. huge_min]             one wave folds all touches of (feature, 64 elements)
//   huge  (huge_min .. giant_min)   the same, started first (the longest folds bound the phase)
//   giant [giant_min ..             segment ranges of one feature folded by several waves side by
//                                   side, partial sums joined afterwards (kernels_tile.h)
#ifndef FFM_HUGE_MIN
#define FFM_HUGE_MIN 128
#endif
constexpr int kHugeMin = FFM_HUGE_MIN;
// ... and from which one wave per (feature, 64 elements) would be the update phase's span: a
// feature with more than kRange occurrences is "giant"
#ifndef FFM_RANGE_SEGS
#define FFM_RANGE_SEGS 16
#endif
constexpr int kRangeSegs = FFM_RANGE_SEGS;
constexpr int kRange = kRangeSegs * kSeg;  // occurrences per range of a giant feature
constexpr int kGiantMin = kRange + 1;
constexpr int kFmRange = 64, kFmGiantMin = kFmRange + 1;
// FFM: from here on one workgroup per (feature, chunk) would be the phase's span again (a feature in
// every tenth row of a 65536-row block is 400 tiles): ranges all over the chip, two short launches more
#ifndef FFM_SUPER_MIN
#define FFM_SUPER_MIN 2048
#endif
constexpr int kSuperMin = FFM_SUPER_MIN;
constexpr int kChainMin = 64;  // the least giant_min an engine may choose (sizes Scratch::giant)
// (4 until the once-only features left the update phase; re-swept since: 8, then 10)
#ifndef FFM_SMALL_MAX
#define FFM_SMALL_MAX 10  // (round 6, re-swept with the regular-block fold: 6 / 8 / 10 / 12 -> C5 step 0.900 / 0.885 / 0.868 / 0.890 ms)
#endif
constexpr int kSmallMax = FFM_SMALL_MAX;  // occurrences per block up to which a feature takes the "small" path
enum { ERR_ROW_TOO_LONG = 1, ERR_FIELD_MAP = 2, ERR_SORT_BARRIER = 4 };

}  // namespace ftrl_dev
