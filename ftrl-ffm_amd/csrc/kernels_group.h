// kernels_group.h -- groups one block of rows by feature ("postings"): the mini-batch scheduler's
// device half.  Replaces the per-feature std::mutex / std::shared_mutex arbitration of the
// reference (src/include/model/ftrl_model.h:49, ffm.h:32): instead of N threads racing for a
// feature's lock, every distinct feature of the block gets ONE owner that applies all of the
// block's touches to it in row order (kernels_update.h).  Integer work only; results are
// deterministic (group contents and order inside a group do not depend on scheduling).
#pragma once
#include "engine_types.h"

namespace ftrl_dev {

constexpr int kGroupThreads = 256;
constexpr int kSortCap = 16384;  // ints of LDS for the in-workgroup bitonic sort (64 KiB)

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

// Entry p: validate (remove_out_range, ftrl_model.cpp:36-42 / ffm.cpp:30-36), find its row,
// count its feature, register first-seen features.  FFM with n_fields <= 64 also collects, per
// row, the fields present once / more than once (s.rowmask: what group_scatter_kernel turns into
// the touched-slot masks) and decides whether the fused row kernel may take the block: it needs
// every row to hold at most max_nv entries and at most one per field, else CNT_NOFUSE.
__global__ __launch_bounds__(kGroupThreads) void group_count_kernel(ModelDev m, Rows rows,
                                                                    Scratch s, int max_nv) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = p < rows.nnz;
  bool valid = false;
  int i = 0;
  if (in) {
    // row_ptr[r] <= p < row_ptr[r+1]  (upper bound - 1; tolerates empty rows)
    int lo = 0, hi = rows.n_rows;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rows.row_ptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
    }
    s.row_of[p] = lo;
    i = rows.feat[p];
    const int f = rows.field ? rows.field[p] : 0;
    valid = i >= 0 && i < m.n_feats;
    if (m.type == 2) valid = valid && f >= 0 && f < m.n_fields;
    if (s.rowmask && valid) {
      const unsigned long long bit = 1ull << f;
      const unsigned long long before = atomicOr(&s.rowmask[2 * lo], bit);
      if (before & bit) {
        atomicOr(&s.rowmask[2 * lo + 1], bit);
        atomicOr(&s.counters[CNT_NOFUSE], 1);
      }
      if (p == rows.row_ptr[lo] && rows.row_ptr[lo + 1] - p > max_nv)
        atomicOr(&s.counters[CNT_NOFUSE], 1);
    }
  }
  if (in) { s.efeat[p] = valid ? i : -1; s.occpos[p] = -1; }
  const bool first = valid && atomicAdd(&s.cnt[i], 1) == 0;
  const int u = wave_append_slot(&s.counters[CNT_NUNIQ], first);
  if (first) s.uniq[u] = i;
}

// Distinct feature u: reserve its group in occ, remember where, re-zero its counter (reused as
// the fill cursor), and sort it into the lists the later passes walk: "multi" (more than one
// occurrence: needs ordering), "small" / "big" (which update path owns it).
__global__ __launch_bounds__(kGroupThreads) void group_alloc_kernel(Scratch s) {
  const int n_uniq = s.counters[CNT_NUNIQ];
  const int n_round = (n_uniq + 63) & ~63;  // whole waves stay converged for the wave-wide ops
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_round; u += gridDim.x * blockDim.x) {
    const bool live = u < n_uniq;
    const int i = live ? s.uniq[u] : 0;
    const int c = live ? s.cnt[i] : 0;
    const int start = wave_reserve(&s.counters[CNT_CURSOR], c);
    const int im = wave_append_slot(&s.counters[CNT_NMULTI], live && c > 1);
    const int is = wave_append_slot(&s.counters[CNT_NSMALL], live && c <= kSmallMax);
    const int ib = wave_append_slot(&s.counters[CNT_NBIG], live && c > kSmallMax && c <= kHugeMin);
    const int ih = wave_append_slot(&s.counters[CNT_NHUGE], live && c > kHugeMin);
    const int iw = wave_append_slot(&s.counters[CNT_NFEW], live && c > 1 && c <= kSmallMax);
    if (!live) continue;
    s.ustart[u] = start;
    s.ucount[u] = c;
    if (s.gmask) s.gmask[start] = 0ull;
    s.fstart[i] = start;
    s.cnt[i] = 0;
    if (c > 1) s.multi[im] = u;
    if (c > 1 && c <= kSmallMax) s.few[iw] = u;
    if (c <= kSmallMax) s.small[is] = u;
    else if (c <= kHugeMin) s.big[ib] = u;
    else s.huge[ih] = u;
  }
}

__global__ __launch_bounds__(kGroupThreads) void group_scatter_kernel(
    Rows rows, Scratch s, const unsigned long long *ownmask) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= rows.nnz) return;
  const int i = s.efeat[p];
  if (i < 0) return;
  const int start = s.fstart[i];
  s.occ[start + atomicAdd(&s.cnt[i], 1)] = p;
  if (s.gmask) {
    // slots of feature i that p's row touches: slot fp is touched when the row holds ANOTHER
    // entry of field fp (FFM::update_vector_w refreshes exactly those, ffm.cpp:72-88), within the
    // field pairs this shard owns
    const int r = s.row_of[p], f = rows.field[p];
    const unsigned long long once = s.rowmask[2 * r], twice = s.rowmask[2 * r + 1];
    const unsigned long long self = 1ull << f;
    atomicOr(&s.gmask[start], ((once & ~self) | (twice & self)) & ownmask[f]);
  }
}

// Groups with more than one entry: sort ascending by entry index (= row order, then position in
// the row), so the owner applies touches in the order the reference's single-thread loop would.
__global__ __launch_bounds__(kGroupThreads) void group_sort_kernel(Scratch s, int *tmp) {
  __shared__ int keys[kSortCap];
  const int n_multi = s.counters[CNT_NMULTI];
  for (int mi = blockIdx.x; mi < n_multi; mi += gridDim.x) {
    const int u = s.multi[mi];
    const int start = s.ustart[u], c = s.ucount[u];
    int *seg = s.occ + start;
    if (c <= 64) {
      // rank sort inside one wave: entries are distinct
      if (threadIdx.x < 64) {
        const int mine = threadIdx.x < c ? seg[threadIdx.x] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < c; j++) rank += (__shfl(mine, j, 64) < mine) ? 1 : 0;
        if (threadIdx.x < c) seg[rank] = mine;  // all lanes loaded before any lane stores
      }
    } else if (c <= kSortCap) {
      int n2 = 128;
      while (n2 < c) n2 <<= 1;
      for (int t = threadIdx.x; t < n2; t += blockDim.x) keys[t] = t < c ? seg[t] : 0x7fffffff;
      __syncthreads();
      for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
          for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
            const int lo = (t / stride) * (stride << 1) + (t % stride);
            const int hi = lo + stride;
            const bool up = ((lo & size) == 0);
            const int a = keys[lo], b = keys[hi];
            if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
          }
          __syncthreads();
        }
      for (int t = threadIdx.x; t < c; t += blockDim.x) seg[t] = keys[t];
    } else {
      // longer than the LDS buffer (a feature present far more often than once per row):
      // O(c^2) rank sort through a global temporary
      int *out = tmp + start;
      for (int t = threadIdx.x; t < c; t += blockDim.x) {
        const int mine = seg[t];
        int rank = 0;
        for (int j = 0; j < c; j++) rank += (seg[j] < mine) ? 1 : 0;
        out[rank] = mine;
      }
      __syncthreads();
      for (int t = threadIdx.x; t < c; t += blockDim.x) seg[t] = out[t];
    }
    __syncthreads();
  }
}

// Groups are final: publish them as {entry, row} pairs so the owners need one load per touch.
__global__ __launch_bounds__(kGroupThreads) void group_expand_kernel(int nnz_valid_cap, Scratch s) {
  const int n = s.counters[CNT_CURSOR];  // number of surviving entries
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int p = s.occ[t];
    s.occ2[t] = make_int2(p, s.row_of[p]);
    // after the scatter cnt[feature] holds its occurrence count again
    const int c = s.cnt[s.efeat[p]];
    s.occpos[p] = c > kSmallMax ? t : (c == 1 ? OCC_ONCE : OCC_FEW);
  }
}

// Once tmp_grad is known: per-occurrence facts of the hot features' entries, in occ order, so
// their owners read them as one contiguous stream.
__global__ __launch_bounds__(kGroupThreads) void hot_meta_kernel(Rows rows, Scratch s) {
  const int n = s.counters[CNT_CURSOR];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int2 pr = s.occ2[t];
    if (s.occpos[pr.x] < 0) continue;
    s.hmeta[t] = make_float2(s.tg[pr.y], rows.val[pr.x]);
  }
}

// Leave cnt[] all zero for the next block.
__global__ __launch_bounds__(kGroupThreads) void group_cleanup_kernel(Scratch s) {
  const int n_uniq = s.counters[CNT_NUNIQ];
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_uniq; u += gridDim.x * blockDim.x)
    s.cnt[s.uniq[u]] = 0;
}

}  // namespace ftrl_dev
