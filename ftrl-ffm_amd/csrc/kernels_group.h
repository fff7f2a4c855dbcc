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

// Entry p: validate (remove_out_range, ftrl_model.cpp:36-42 / ffm.cpp:30-36), find its row,
// count its feature, register first-seen features.
__global__ __launch_bounds__(kGroupThreads) void group_count_kernel(ModelDev m, Rows rows,
                                                                    Scratch s) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= rows.nnz) return;
  // row_ptr[r] <= p < row_ptr[r+1]  (upper bound - 1; tolerates empty rows)
  int lo = 0, hi = rows.n_rows;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rows.row_ptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
  }
  s.row_of[p] = lo;
  const int i = rows.feat[p];
  const int f = rows.field ? rows.field[p] : 0;
  bool valid = i >= 0 && i < m.n_feats;
  if (m.type == 2) valid = valid && f >= 0 && f < m.n_fields;
  s.efeat[p] = valid ? i : -1;
  if (!valid) return;
  if (atomicAdd(&s.cnt[i], 1) == 0) {
    const int u = atomicAdd(&s.counters[CNT_NUNIQ], 1);
    s.uniq[u] = i;
  }
}

// Distinct feature u: reserve its group in occ, remember where, re-zero its counter (reused as
// the fill cursor), list it for sorting when it occurs more than once.
__global__ __launch_bounds__(kGroupThreads) void group_alloc_kernel(Scratch s) {
  const int n_uniq = s.counters[CNT_NUNIQ];
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_uniq; u += gridDim.x * blockDim.x) {
    const int i = s.uniq[u];
    const int c = s.cnt[i];
    const int start = atomicAdd(&s.counters[CNT_CURSOR], c);
    s.ustart[u] = start;
    s.ucount[u] = c;
    s.fstart[i] = start;
    s.cnt[i] = 0;
    if (c > 1) s.multi[atomicAdd(&s.counters[CNT_NMULTI], 1)] = u;
  }
}

__global__ __launch_bounds__(kGroupThreads) void group_scatter_kernel(Rows rows, Scratch s) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= rows.nnz) return;
  const int i = s.efeat[p];
  if (i < 0) return;
  s.occ[s.fstart[i] + atomicAdd(&s.cnt[i], 1)] = p;
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

// Leave cnt[] all zero for the next block.
__global__ __launch_bounds__(kGroupThreads) void group_cleanup_kernel(Scratch s) {
  const int n_uniq = s.counters[CNT_NUNIQ];
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_uniq; u += gridDim.x * blockDim.x)
    s.cnt[s.uniq[u]] = 0;
}

}  // namespace ftrl_dev
