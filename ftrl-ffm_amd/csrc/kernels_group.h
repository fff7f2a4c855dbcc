// kernels_group.h -- groups one block of rows by feature ("postings"): the mini-batch scheduler's
// device half.  Three steps: group_keys_kernel (one sort key per entry) -> a stable LSD radix sort
// of (feature id, entry index) pairs (rocPRIM device primitive) -> group_finish_kernel (group
// boundaries, owner lists, touched-slot masks).  No per-feature arrays and no same-address atomic
// chains: a feature present in a thousand rows costs what a thousand distinct features cost.  Replaces the per-feature std::mutex / std::shared_mutex arbitration of the
// reference (src/include/model/ftrl_model.h:49, ffm.h:32): instead of N threads racing for a
// feature's lock, every distinct feature of the block gets ONE owner that applies all of the
// block's touches to it in row order (kernels_update.h).  Integer work only; results are
// deterministic (group contents and order inside a group do not depend on scheduling).
#pragma once
#include "engine_types.h"

namespace ftrl_dev {

constexpr int kGroupThreads = 256;
#ifndef FFM_FINISH_THREADS
#define FFM_FINISH_THREADS 256
#endif
// group_finish_kernel: one atomic per list per workgroup.  (1024-thread workgroups need 16 free
// wave slots on one CU at once: beside the persistent update kernels they waited for that.)
constexpr int kFinishThreads = FFM_FINISH_THREADS;

// One atomic per wave instead of one per lane: lanes with pred get consecutive slots.
__device__ __forceinline__ int wave_append_slot(int *counter, bool pred) {
  const unsigned long long mask = __ballot(pred);
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll(static_cast<long long>(mask)) - 1;
  int base = 0;
  if (pred && lane == leader) base = atomicAdd(counter, __popcll(mask));
  base = __shfl(base, leader < 0 ? 0 : leader, 64);
  return base + __popcll(mask & ((1ull << lane) - 1ull));
}

// Same for a per-lane amount: wave-wide exclusive prefix sum, one atomic for the total.
__device__ __forceinline__ int wave_reserve(int *counter, int amount) {
  const int lane = threadIdx.x & 63;
  int incl = amount;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  int base = 0;
  if (lane == 63) base = atomicAdd(counter, total);
  base = __shfl(base, 63, 64);
  return base + incl - amount;
}

// Entry p: validate (remove_out_range, ftrl_model.cpp:36-42 / ffm.cpp:30-36), find its row, emit
// its sort key (the feature id; n_feats for erased entries, which therefore sort last).  FFM with
// n_fields <= 64 also collects, per row, the fields present once / more than once (s.rowmask: what
// group_finish_kernel turns into the touched-slot masks).
// A row longer than max_row_nnz (the row kernels' LDS capacity) cannot be trained: it raises
// ERR_ROW_TOO_LONG here, before anything has touched the model, and every later kernel of the
// block then does nothing (CNT_ERROR) -- the block is a no-op and the caller learns about it from
// the next ffm_engine_sync / check_errors / flush.
__global__ __launch_bounds__(kGroupThreads) void group_keys_kernel(ModelDev m, Rows rows, Scratch s,
                                                                   int max_row_nnz) {
  const int pp = blockIdx.x * blockDim.x + threadIdx.x;  // the grid covers whole waves
  const bool in = pp < rows.nnz;
  const int p = in ? pp : rows.nnz - 1;  // idle lanes of the last wave shadow the last entry
  // row_ptr[r] <= p < row_ptr[r+1]  (upper bound - 1; tolerates empty rows)
  int lo = 0, hi = rows.n_rows;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rows.row_ptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
  }
  const int i = rows.feat[p];
  const int f = rows.field ? rows.field[p] : 0;
  bool valid = i >= 0 && i < m.n_feats;
  if (m.type == 2) valid = valid && f >= 0 && f < m.n_fields;
  if (m.field_start && valid) {
    // compact shard: an id must lie in its field's id range (that is what makes field(i) a
    // function of i); an entry that does not is a caller error and voids the block like an
    // over-long row.  Columns this shard owns nothing of are dropped here.
    if (in && (i < m.field_start[f] || i >= m.field_start[f + 1])) {
      atomicOr(&s.counters[CNT_ERROR], ERR_FIELD_MAP);
      atomicOr(s.err, ERR_FIELD_MAP);
    }
    valid = keeps_field(m, f);
  }
  if (m.sort_start) {
    // The range sort's short cut: a block whose every row holds exactly one entry per field, in field
    // order, every id inside its field's range, has range f's entries at f, f + n_fields, ... -- no scan
    // needed to find them.  Anything else (a missing or repeated field, an erased entry, an id under
    // another field) takes the general path; only a wave that sees such an entry pays the atomic.
    const int b0 = rows.row_ptr[lo];
    bool regular = valid && rows.row_ptr[lo + 1] - b0 == m.n_fields && f == p - b0;
    if (regular) regular = i >= m.sort_start[f] && i < m.sort_start[f + 1];
    if (__any(in && !regular) && (threadIdx.x & 63) == 0) atomicOr(&s.counters[CNT_IRREGULAR], 1);
  }
  if (in) {
    s.row_of[p] = lo;
    s.key[p] = valid ? static_cast<unsigned>(i) : static_cast<unsigned>(m.n_feats);
    s.occpos[p] = OCC_FEW;
    s.uflag[p] = 0;
    if (s.gmask) { s.gmask[p] = 0ull; s.cmask[p] = 0ull; }
  }
  valid = valid && in;
  if (in && p == rows.row_ptr[lo] && rows.row_ptr[lo + 1] - p > max_row_nnz) {
    atomicOr(&s.counters[CNT_ERROR], ERR_ROW_TOO_LONG);
    atomicOr(s.err, ERR_ROW_TOO_LONG);
  }
  if (s.rowmask) {
    // fields present once / more than once per row: combined over the lanes of the wave that
    // share the row (entries of a row are consecutive), then one atomic per row piece
    const int lane = threadIdx.x & 63;
    unsigned long long once = valid ? 1ull << f : 0ull, twice = 0ull;
    const int prev_row = __shfl_up(lo, 1, 64);
    const unsigned long long starts = __ballot(lane == 0 || prev_row != lo);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const int seg0 = 63 - __clzll(static_cast<long long>(starts & upto));  // first lane of my row piece
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = __shfl_up(once, d, 64), w = __shfl_up(twice, d, 64);
      if (lane - d >= seg0) {
        twice |= w | (o & once);
        once |= o;
      }
    }
    const int next_row = __shfl_down(lo, 1, 64);
    if (in && (lane == 63 || next_row != lo || pp == rows.nnz - 1)) {  // last lane of the piece
      if (once) {
        const unsigned long long before = atomicOr(&s.rowmask[2 * lo], once);
        twice |= before & once;
      }
      if (twice) atomicOr(&s.rowmask[2 * lo + 1], twice);
    }
  }
}

// first index in skey[0, n) whose key is >= k (strict = false) or > k (strict = true)
__device__ __forceinline__ int sorted_bound(const unsigned *skey, int n, unsigned k, bool strict) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const unsigned v = skey[mid];
    if (strict ? v <= k : v < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Sorted position t (s.skey ascending, s.occ = the entries in that order; equal keys keep their
// entry order, i.e. row order then position in the row -- the order the reference's one-thread
// loop would touch them in).  Finds t's group [lower, upper): from the group heads inside the wave
// when the group starts / ends there, by binary search when it crosses the wave.  Group heads
// register the distinct feature (uniq / ustart / ucount) and sort it into the owner lists; every
// position publishes {entry, row}, its entry's occurrence class, and ORs its touched-slot mask
// into the group's (one atomic per group piece per wave).
__global__ __launch_bounds__(kFinishThreads) void group_finish_kernel(ModelDev m, Rows rows,
                                                                     Scratch s) {
  if (s.counters[CNT_ERROR]) return;  // untrainable block: no groups, no owners, nothing runs
  const int nnz = rows.nnz;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // the grid covers whole waves
  const bool in = t < nnz;
  const unsigned sentinel = static_cast<unsigned>(m.n_feats);
  const unsigned K = in ? s.skey[t] : sentinel;
  const bool valid = in && K < sentinel;
  const unsigned Kprev = (in && t > 0) ? s.skey[t - 1] : 0xffffffffu;
  const unsigned Knext = (t + 1 < nnz) ? s.skey[t + 1] : 0xffffffffu;
  const bool head = valid && (t == 0 || Kprev != K);
  const bool last = valid && Knext != K;
  if (last && Knext >= sentinel) s.counters[CNT_CURSOR] = t + 1;  // number of surviving entries
  // group bounds
  const unsigned long long heads = __ballot(head), lasts = __ballot(last);
  const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);  // lanes 0..lane
  const unsigned long long from = ~0ull << lane;                                   // lanes lane..63
  const unsigned long long hb = heads & upto, lb = lasts & from;
  const int hl = hb ? 63 - __clzll(static_cast<long long>(hb)) : -1;   // my group's head lane
  const int ll = lb ? __ffsll(static_cast<long long>(lb)) - 1 : -1;     // my group's last lane
  // a group that crosses the wave's edges: lane 0 / lane 63 search its bound for the whole piece
  int lower0 = 0, upper63 = 0;
  if (lane == 0 && valid && !head) lower0 = sorted_bound(s.skey, nnz, K, false);
  if (lane == 63 && valid && !last) upper63 = sorted_bound(s.skey, nnz, K, true);
  lower0 = __shfl(lower0, 0, 64);
  upper63 = __shfl(upper63, 63, 64);
  int lower = 0, upper = 0;
  if (valid) {
    lower = hl >= 0 ? t - (lane - hl) : lower0;
    upper = ll >= 0 ? t + (ll - lane) + 1 : upper63;
  }
  const int c = upper - lower;
  int p = 0;
  // dup: the feature occurs in this position's row a second time (equal keys are in entry order, so
  // two occurrences of one row are neighbours): UF_DUP for the group, and every slot this row
  // touches joins the feature's serial slots below (the checker under oracle/: mark_serial)
  bool dup = false;
  if (valid) {
    p = s.occ[t];
    const int r = s.row_of[p];
    s.occ2[t] = make_int2(p, r);
    s.occpos[p] = c > kSmallMax ? t : (c == 1 ? OCC_ONCE : OCC_FEW);
    dup = (!head && s.row_of[s.occ[t - 1]] == r) || (!last && s.row_of[s.occ[t + 1]] == r);
    if (dup) atomicOr(&s.uflag[lower], UF_DUP);
  }
  // distinct features and their owner lists: slots handed out per workgroup (one atomic per
  // list per workgroup -- per-wave atomics on five shared counters would be a serial chain)
  constexpr int kLists = 7;
  __shared__ int wave_cnt[kFinishThreads / 64][kLists];
  __shared__ int wave_base[kFinishThreads / 64][kLists];
  const int wv = threadIdx.x >> 6;
  const bool pred[kLists] = {head, head && c <= kSmallMax, head && c > 1 && c <= kSmallMax,
                             head && c > kSmallMax && c <= m.huge_min,
                             head && c > m.huge_min && c < m.giant_min, head && c == 1,
                             head && c > m.huge_min && c >= m.giant_min};
  const int which[kLists] = {CNT_NUNIQ, CNT_NSMALL, CNT_NFEW, CNT_NBIG, CNT_NHUGE, CNT_NSINGLE, CNT_NGIANT};
  unsigned long long pm[kLists];
#pragma unroll
  for (int q = 0; q < kLists; q++) {
    pm[q] = __ballot(pred[q]);
    if (lane == 0) wave_cnt[wv][q] = __popcll(pm[q]);
  }
  __syncthreads();
  if (threadIdx.x < kLists) {
    const int q = threadIdx.x;
    int total = 0;
    for (int w = 0; w < kFinishThreads / 64; w++) { wave_base[w][q] = total; total += wave_cnt[w][q]; }
    const int base = total ? atomicAdd(&s.counters[which[q]], total) : 0;
    for (int w = 0; w < kFinishThreads / 64; w++) wave_base[w][q] += base;
  }
  __syncthreads();
  int slot[kLists];
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < kLists; q++) slot[q] = wave_base[wv][q] + __popcll(pm[q] & below);
  const int u = slot[0], is = slot[1], iw = slot[2], ib = slot[3], ih = slot[4];
  if (head) {
    const int fld = rows.field ? rows.field[p] : 0;
    s.udesc[u] = make_int4(static_cast<int>(K), t, c, fld);
    if (c == 1) s.sdesc[slot[5]] = make_int4(static_cast<int>(K), p, s.row_of[p], fld);
    if (c > 1 && c <= kSmallMax) s.few[iw] = u;
    if (c <= kSmallMax) s.small[is] = u;
    else if (c <= m.huge_min) s.big[ib] = u;
    else if (c < m.giant_min) s.huge[ih] = u;
    else {
      // a giant feature: kSeg-occurrence segment slots for its partial sums, one list entry per
      // range of range_len occurrences (kernels_tile.h / kernels_update.h: the ranges are folded side by side)
      const int gi = slot[6];
      s.giant[gi] = u;
      // (the host only asks "any at all?": a plain system-scope store of 1 -- no read-modify-write on
      // host memory, which would need PCIe atomics to land)
      if (c >= m.super_min) __hip_atomic_store(s.n_super, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // segment slots and range entries only for the features that are folded as ranges: every FM
      // giant, FFM from super_min on (below it a workgroup folds the feature: ffm_coop_items)
      if (m.type != 2 || c >= m.super_min) {
        const int n_seg = (c + kSeg - 1) / kSeg, n_rng = (c + m.range_len - 1) / m.range_len;
        s.gseg[gi] = atomicAdd(&s.counters[CNT_NSEG], n_seg);
        const int rb = atomicAdd(&s.counters[CNT_NRANGE], n_rng);
        for (int r = 0; r < n_rng; r++) s.grange[rb + r] = make_int2(gi, r);
      } else {
        s.gseg[gi] = 0;
      }
    }
  }
  // slots of the feature that p's row touches: slot fp is touched when the row holds ANOTHER
  // entry of field fp (FFM::update_vector_w refreshes exactly those, ffm.cpp:72-88), within the
  // field pairs this shard owns; OR-ed over the group
  if (s.gmask) {
    unsigned long long tm = 0ull, tc = 0ull;
    if (valid) {
      const int r = s.row_of[p], f = rows.field[p];
      const unsigned long long once = s.rowmask[2 * r], twice = s.rowmask[2 * r + 1];
      const unsigned long long self = 1ull << f;
      tm = ((once & ~self) | (twice & self)) & m.ownmask[f];
      tc = twice & m.ownmask[f];  // partner fields that hold several entries in this row
      if (dup) tc |= tm;
    }
    const int seg0 = hl >= 0 ? hl : 0;  // first lane of my group's piece in this wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long v = __shfl_up(tm, d, 64), vc = __shfl_up(tc, d, 64);
      if (lane - d >= seg0) { tm |= v; tc |= vc; }
    }
    const bool piece_end = valid && (last || lane == 63 || t == nnz - 1);
    if (piece_end) {
      atomicOr(&s.gmask[lower], tm);
      if (tc) atomicOr(&s.cmask[lower], tc);
    }
  }
}

// create-time check that device stores to the mapped host words of Scratch::n_super land (engine.hip)
__global__ void host_word_probe_kernel(int *w, int n) {
  if (static_cast<int>(threadIdx.x) < n)
    __hip_atomic_store(w + threadIdx.x, 0x5eed + static_cast<int>(threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace ftrl_dev
