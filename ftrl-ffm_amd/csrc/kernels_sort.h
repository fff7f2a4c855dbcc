// kernels_sort.h -- the grouping's sort as ONE launch: a stable LSD radix sort of (feature id, entry
// index) pairs, 8 bits per pass, by a few co-resident 256-thread workgroups that meet at grid
// barriers between the phases of a pass.
//
// Why not the library primitive (rocPRIM Onesweep, rounds 1-4): for a block of 320 k entries it is
// ~14 launches (histogram, scan, and per pass two fills + twenty 1024-thread workgroups), each of which
// has to find room on a GPU that the row / update kernels keep full -- a 1024-thread workgroup needs
// 16 free wave slots on ONE CU, and even a one-workgroup fill waited 50-80 us for a slot
// (profiles/archive/r04_experiments.md).  The grouping is two blocks ahead of its use, but it is a CHAIN: at
// steps under ~0.5 ms (FM, FFM k = 4, small blocks) the look-ahead queue's ~17 dependent launches per
// block took as long as the step itself and the main stream waited for them one to one.  Here the
// sort's workgroups get on the machine once and stay until the block is sorted.
//
// Pass p (digit = bits [8p, 8p+8) of the key), workgroup w owning the contiguous chunk w of the
// pass's input:
//   A  digit histogram of the chunk (LDS atomics)              -> cnt[digit][w]
//      -- grid barrier --
//   B  base[digit] = sum of cnt over (smaller digits, every workgroup) + (this digit, workgroups < w):
//      every workgroup derives its own 256 bases from the table (no second barrier)
//   C  the chunk in tiles of 256 keys in order: lanes with the same digit find each other with eight
//      ballots (rank inside the wave = position among the peers), the four waves' counts go through
//      LDS (one workgroup barrier per tile), destination = running base of the digit + earlier waves'
//      count + rank.  Stable.
//      -- grid barrier (the next pass reads what every workgroup scattered) --
// Integer work only; the result does not depend on scheduling.
#pragma once
#include "engine_types.h"

namespace ftrl_dev {

constexpr int kSortThreads = 256;
constexpr int kSortMaxWgs = 64;
constexpr int kSortBatch = 4;  // tiles whose loads are in flight together

struct SortJob {
  const unsigned *key;  // [n] input keys (values are the indices 0 .. n-1)
  unsigned *okey;       // [n] sorted keys
  int *oval;            // [n] the indices in sorted order
  unsigned *tkey;       // [n] scratch
  int *tval;            // [n] scratch
  unsigned *cnt;        // [256][gridDim.x] scratch
  int *bar;             // grid barrier word, zero at launch (the grouping's counters are cleared per block)
  int *block_err;       // the block's CNT_ERROR word and the engine's sticky flags: where a barrier that
  int *err;             //   never completes reports ERR_SORT_BARRIER (the block is then skipped as a whole)
  int n, passes;
};

// What the workgroups hand each other (the ping-pong buffers, the count table) moves with
// agent-coherent accesses (sc1: past the XCD's L2) instead of release / acquire fences: a fence is a
// write-back or an invalidate of the whole L2 of the XCD it runs on, ~450 of each per block with 64
// workgroups and seven barriers -- beside a row kernel that lives on that L2 (measured: row kernel
// 518 -> 577 us, the step 0.95 -> 1.01 ms with fences).
// MEMORY-MODEL NOTE: relaxed agent-scope atomics order nothing in the HIP memory model; handing data
// over with them plus s_waitcnt is formally a data race.  It relies on what gfx942 / gfx950 do: an
// sc1 store is written through to the memory side before the wave's vmcnt for it retires, and an
// sc1 load does not hit in a non-coherent L2 line.  ffm_engine_create takes this sort only on the
// architecture it was validated on (gfx950: tests/test_gpu_parity.py, a 2.5 M-entry four-pass
// block against the library sort and the oracle); anything else keeps rocPRIM.
template <class T> __device__ __forceinline__ T coh_load(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ void coh_store(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The usual counter barrier.  It needs every workgroup of the launch on the machine at once: the
// grid is clamped to what the device can hold of this kernel (engine: sort_grid_cap, from
// hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs) -- other kernels on the device never wait for
// the sort, so its workgroups all arrive once those drain -- and the spin is BOUNDED: a workgroup
// that has waited two seconds (CU masks, a partitioned device, ...) raises ERR_SORT_BARRIER, the block
// is skipped as a whole (CNT_ERROR, like an over-long row) and the caller learns about it at the
// next sync.  Returns false then; every workgroup leaves within the same bound.
// __syncthreads() waits for the wave's own stores (vmcnt) before it arrives.
constexpr unsigned long long kSortSpinTicks = 200000000ull;  // of s_memrealtime (100 MHz): two seconds
__device__ __forceinline__ bool sort_grid_barrier(const SortJob &a, int n_wg, int &target) {
  __shared__ int ok;
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    target += n_wg;
    __hip_atomic_fetch_add(a.bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (bounded by the clock, not by an iteration count: an iteration is an agent-scope load whose
    // latency depends on what else the chip is doing -- ADVICE r05)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool arrived;
    while (!(arrived = __hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) &&
           __builtin_amdgcn_s_memrealtime() - t0 < kSortSpinTicks)
      __builtin_amdgcn_s_sleep(8);
    ok = arrived;
    if (!ok) {
      atomicOr(a.block_err, ERR_SORT_BARRIER);
      atomicOr(a.err, ERR_SORT_BARRIER);
    }
  }
  __syncthreads();
  return ok != 0;
}

// One tile of 256 keys of phase C.  `it` counts the tiles of the pass: the wave counts rotate over
// three buffers and the running destinations over two, so that a tile needs ONE workgroup barrier
// (the buffer written here was zeroed a tile ago, behind that tile's barrier).
struct SortLds {
  unsigned hist[2][256];     // A: hist[0] = the chunk's digit counts; C: running destination per digit
  unsigned wcnt[3][4][256];  // C: per wave, how many keys of the tile carry each digit
  unsigned wtot[4];
};
// d: the key's digit of this pass (0 .. 255).
__device__ __forceinline__ void sort_tile(SortLds &l, int it, bool in, unsigned key, int val, unsigned d,
                                          unsigned *kout, int *vout) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cb = it % 3, zb = (it + 2) % 3, hb = it & 1;
  unsigned long long peers = __ballot(in);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    peers &= bit ? bal : ~bal;
  }
  const unsigned rank = __popcll(peers & ((1ull << lane) - 1ull));
  if (in && rank == 0u) l.wcnt[cb][wave][d] = __popcll(peers);
  __syncthreads();
  unsigned dst = l.hist[hb][d] + rank;
  for (int v = 0; v < wave; v++) dst += l.wcnt[cb][v][d];
  if (in) {
    coh_store(kout + dst, key);
    coh_store(vout + dst, val);
  }
  l.hist[hb ^ 1][t] = l.hist[hb][t] + l.wcnt[cb][0][t] + l.wcnt[cb][1][t] + l.wcnt[cb][2][t] + l.wcnt[cb][3][t];
  l.wcnt[zb][0][t] = 0u; l.wcnt[zb][1][t] = 0u; l.wcnt[zb][2][t] = 0u; l.wcnt[zb][3][t] = 0u;
}

__global__ __launch_bounds__(kSortThreads) void group_sort_kernel(SortJob a) {
  __shared__ SortLds l;
  const int W = gridDim.x, w = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (((a.n + W - 1) / W) + kSortThreads - 1) & ~(kSortThreads - 1);
  const int c0 = min(a.n, w * per), c1 = min(a.n, c0 + per);
  int target = 0;
  for (int p = 0; p < a.passes; p++) {
    const bool to_out = ((a.passes - 1 - p) & 1) == 0;  // the last pass lands in (okey, oval)
    const unsigned *kin = p == 0 ? a.key : (to_out ? a.tkey : a.okey);
    const int *vin = p == 0 ? nullptr : (to_out ? a.tval : a.oval);
    unsigned *kout = to_out ? a.okey : a.tkey;
    int *vout = to_out ? a.oval : a.tval;
    const int shift = 8 * p;
    // ---- A
    l.hist[0][t] = 0u;
#pragma unroll
    for (int b = 0; b < 3; b++) { l.wcnt[b][0][t] = 0u; l.wcnt[b][1][t] = 0u; l.wcnt[b][2][t] = 0u; l.wcnt[b][3][t] = 0u; }
    __syncthreads();
    for (int i = c0 + t; i < c1; i += kSortBatch * kSortThreads) {
      unsigned k[kSortBatch];
#pragma unroll
      for (int j = 0; j < kSortBatch; j++)
        k[j] = i + j * kSortThreads < c1 ? (p == 0 ? kin[i + j * kSortThreads] : coh_load(kin + i + j * kSortThreads)) : 0u;
#pragma unroll
      for (int j = 0; j < kSortBatch; j++)
        if (i + j * kSortThreads < c1) atomicAdd(&l.hist[0][(k[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    coh_store(a.cnt + t * W + w, l.hist[0][t]);
    if (!sort_grid_barrier(a, W, target)) return;
    // ---- B: thread t owns digit t
    unsigned total = 0u, before = 0u;
    for (int v = 0; v < W; v++) {
      const unsigned c = coh_load(a.cnt + t * W + v);
      total += c;
      before += v < w ? c : 0u;
    }
    unsigned incl = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    if (lane == 63) l.wtot[wave] = incl;
    __syncthreads();
    unsigned prefix = incl - total;
    for (int v = 0; v < wave; v++) prefix += l.wtot[v];
    l.hist[0][t] = prefix + before;
    __syncthreads();
    // ---- C
    int it = 0;
    for (int base = c0; base < c1; base += kSortBatch * kSortThreads) {
      unsigned k[kSortBatch];
      int val[kSortBatch];
#pragma unroll
      for (int j = 0; j < kSortBatch; j++) {
        const int i = base + j * kSortThreads + t;
        k[j] = i < c1 ? (p == 0 ? kin[i] : coh_load(kin + i)) : 0u;
        val[j] = i < c1 ? (vin ? coh_load(vin + i) : i) : 0;
      }
#pragma unroll
      for (int j = 0; j < kSortBatch; j++) {
        if (base + j * kSortThreads >= c1) break;  // (uniform)
        sort_tile(l, it++, base + j * kSortThreads + t < c1, k[j], val[j], (k[j] >> shift) & 255u, kout, vout);
      }
    }
    if (p + 1 < a.passes && !sort_grid_barrier(a, W, target)) return;
  }
}

// ---- the same sort for ids that come in RANGES (round 6): one workgroup per range, no grid barrier ----
// libffm data lays a field's ids out as one contiguous range (python/generate_data.py:272-306, the
// bundled data, the synthetic generator); an engine that is told the ranges (ffm_engine_config::
// field_start) sorts every range by itself: the sorted block is the ranges' sorted entries one after
// the other.  Workgroup g
//   A1  scans ALL keys once, a contiguous quarter per wave, counting the keys below its range and
//       inside it (the first count is where its output starts: nobody waits for anybody);
//   A2  scans them again and appends its own (key, entry) pairs in entry order (stable);
//   B   LSD radix passes over its own pairs, 8 bits of (key - range start) per pass, as many as the
//       range's width needs (three for 846 k ids): histogram, 256 bases, then tiles of 256 keys through
//       sort_tile -- the ping-pong buffers are private to the workgroup.
// Nothing depends on which field an ENTRY claims to be of: an id that sits under the "wrong" field is
// simply sorted with the range it lies in, so the result is the stable sort by id whatever the data.
// 39 workgroups of four waves and 14 KB of LDS each take a slot like any row of the kernels they run
// beside; the library sort's ~14 launches of 1024-thread workgroups each waited for a whole CU to
// drain (a pass: 7 us alone, 100-450 us beside the row / update kernels), and the look-ahead queue --
// upload, keys, sort, finish, per block, on ONE queue -- was as long as a C5 step with the H2D and
// longer than a C3 step.  Erased entries (key = the sentinel n_feats) fall in no range; the last
// workgroup writes the sentinel keys behind the sorted ones.
struct RangeSortJob {
  const unsigned *key;  // [n] input keys (values are the indices 0 .. n-1)
  unsigned *okey;       // [n] sorted keys
  int *oval;            // [n] the indices in sorted order
  unsigned *tkey;       // [n] scratch
  int *tval;            // [n] scratch
  const int *start;     // [n_ranges + 1] ascending range boundaries, start[0] = 0, start[n_ranges] = sentinel
  int n, n_ranges;
  unsigned sentinel;
  const int *irregular;  // 0: range g's entries are g, g + n_ranges, ... (group_keys_kernel: CNT_IRREGULAR)
};
constexpr int kRangeScanBatch = 8;  // 16-byte loads per lane in flight while a wave scans
__global__ __launch_bounds__(kSortThreads) void group_sort_ranges_kernel(RangeSortJob a) {
  __shared__ SortLds l;
  __shared__ int w_below[4], w_mine[4];
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned lo = static_cast<unsigned>(a.start[g]), hi = static_cast<unsigned>(a.start[g + 1]);
  // ---- A1: this wave's quarter of the keys, 16 bytes per lane and kRangeScanBatch loads in flight
  // (every workgroup reads ALL keys, twice: what is in flight per wave is what the scans cost)
  const int per = ((((a.n + 3) >> 2) + 255) >> 8) << 8;  // a multiple of 256 keys: whole uint4 pieces
  const int q0 = min(a.n, wave * per), q1 = min(a.n, q0 + per);
  const uint4 *key4 = reinterpret_cast<const uint4 *>(a.key);  // (hipMalloc'ed: 16-byte aligned; q0 is a multiple of 4)
  auto load4 = [&](int i) {  // the keys i .. i + 3 (i a multiple of 4), 0xffffffff beyond the quarter
    uint4 k = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    if (i + 4 <= q1) k = key4[i >> 2];
    else if (i < q1) {
      k.x = a.key[i];
      if (i + 1 < q1) k.y = a.key[i + 1];
      if (i + 2 < q1) k.z = a.key[i + 2];
    }
    return k;
  };
  auto in_range = [&](unsigned k) { return k - lo < hi - lo; };  // lo <= k < hi (unsigned wrap; hi >= lo)
  // (the short cut: every row is one entry per field in field order, ids inside their fields' ranges --
  // range g holds exactly the entries g, g + n_ranges, ...: no scan)
  const bool regular = a.irregular && a.n % a.n_ranges == 0 && *a.irregular == 0;
  int below = 0, mine = 0;
  if (!regular) {
    for (int i = q0; i < q1; i += 256 * kRangeScanBatch) {
      uint4 k[kRangeScanBatch];
#pragma unroll
      for (int j = 0; j < kRangeScanBatch; j++) k[j] = load4(i + 256 * j + 4 * lane);
#pragma unroll
      for (int j = 0; j < kRangeScanBatch; j++) {
        below += (k[j].x < lo) + (k[j].y < lo) + (k[j].z < lo) + (k[j].w < lo);
        mine += in_range(k[j].x) + in_range(k[j].y) + in_range(k[j].z) + in_range(k[j].w);
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      below += __shfl_xor(below, d, 64);
      mine += __shfl_xor(mine, d, 64);
    }
  }
  if (lane == 0) { w_below[wave] = below; w_mine[wave] = mine; }
  __syncthreads();
  const int rows_reg = a.n / a.n_ranges;
  const int off = regular ? g * rows_reg : w_below[0] + w_below[1] + w_below[2] + w_below[3];
  const int n_g = regular ? rows_reg : w_mine[0] + w_mine[1] + w_mine[2] + w_mine[3];
  int at = off;
  for (int v = 0; v < wave; v++) at += w_mine[v];
  const unsigned width = hi > lo ? hi - lo : 1u;
  const int bits = width > 1u ? 32 - __clz(static_cast<int>(width - 1u)) : 0;
  const int passes = (bits + 7) >> 3;
  // (an even number of passes starts in the output buffers, an odd one in the scratch)
  unsigned *const k0 = (passes & 1) ? a.tkey : a.okey, *const k1 = (passes & 1) ? a.okey : a.tkey;
  int *const v0 = (passes & 1) ? a.tval : a.oval, *const v1 = (passes & 1) ? a.oval : a.tval;
  // ---- A2: append this range's pairs in entry order (a lane's four keys are four consecutive entries)
  if (regular) {
    for (int r0 = t; r0 < rows_reg; r0 += kSortThreads * kRangeScanBatch) {
      unsigned k[kRangeScanBatch];
#pragma unroll
      for (int j = 0; j < kRangeScanBatch; j++) {
        const int r = r0 + j * kSortThreads;
        k[j] = r < rows_reg ? a.key[static_cast<size_t>(r) * a.n_ranges + g] : 0u;
      }
#pragma unroll
      for (int j = 0; j < kRangeScanBatch; j++) {
        const int r = r0 + j * kSortThreads;
        if (r < rows_reg) {
          coh_store(k0 + off + r, k[j]);
          coh_store(v0 + off + r, r * a.n_ranges + g);
        }
      }
    }
  } else if (n_g > 0) {
    for (int i = q0; i < q1; i += 256 * kRangeScanBatch) {
      uint4 k[kRangeScanBatch];
#pragma unroll
      for (int j = 0; j < kRangeScanBatch; j++) k[j] = load4(i + 256 * j + 4 * lane);
#pragma unroll
      for (int j = 0; j < kRangeScanBatch; j++) {
        const bool mx = in_range(k[j].x), my = in_range(k[j].y), mz = in_range(k[j].z), mw = in_range(k[j].w);
        const int c = mx + my + mz + mw;  // 0 .. 4: its exclusive prefix over the lanes from three ballots
        const unsigned long long lower = (1ull << lane) - 1ull;
        const unsigned long long b0 = __ballot(c & 1), b1 = __ballot(c & 2), b2 = __ballot(c & 4);
        if (b0 | b1 | b2) {
          int dst = at + __popcll(b0 & lower) + 2 * __popcll(b1 & lower) + 4 * __popcll(b2 & lower);
          const int e0 = i + 256 * j + 4 * lane;
          if (mx) { coh_store(k0 + dst, k[j].x); coh_store(v0 + dst, e0); dst++; }
          if (my) { coh_store(k0 + dst, k[j].y); coh_store(v0 + dst, e0 + 1); dst++; }
          if (mz) { coh_store(k0 + dst, k[j].z); coh_store(v0 + dst, e0 + 2); dst++; }
          if (mw) { coh_store(k0 + dst, k[j].w); coh_store(v0 + dst, e0 + 3); }
          at += __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
        }
      }
    }
  }
  // the sentinel keys of the erased entries, behind every range's pairs
  if (g == a.n_ranges - 1)
    for (int i = off + n_g + t; i < a.n; i += kSortThreads) { a.okey[i] = a.sentinel; a.oval[i] = 0; }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  // ---- B: LSD radix passes over [off, off + n_g)
  const int c0 = off, c1 = off + n_g;
  for (int p = 0; p < passes; p++) {
    const unsigned *kin = (p & 1) ? k1 : k0;
    const int *vin = (p & 1) ? v1 : v0;
    unsigned *kout = (p & 1) ? k0 : k1;
    int *vout = (p & 1) ? v0 : v1;
    const int shift = 8 * p;
    l.hist[0][t] = 0u;
#pragma unroll
    for (int b = 0; b < 3; b++) { l.wcnt[b][0][t] = 0u; l.wcnt[b][1][t] = 0u; l.wcnt[b][2][t] = 0u; l.wcnt[b][3][t] = 0u; }
    __syncthreads();
    for (int i = c0 + t; i < c1; i += kSortBatch * kSortThreads) {
      unsigned k[kSortBatch];
#pragma unroll
      for (int j = 0; j < kSortBatch; j++) k[j] = i + j * kSortThreads < c1 ? coh_load(kin + i + j * kSortThreads) : 0u;
#pragma unroll
      for (int j = 0; j < kSortBatch; j++)
        if (i + j * kSortThreads < c1) atomicAdd(&l.hist[0][((k[j] - lo) >> shift) & 255u], 1u);
    }
    __syncthreads();
    // thread t owns digit t: where its keys start
    const unsigned total = l.hist[0][t];
    unsigned incl = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    if (lane == 63) l.wtot[wave] = incl;
    __syncthreads();
    unsigned prefix = incl - total;
    for (int v = 0; v < wave; v++) prefix += l.wtot[v];
    l.hist[0][t] = static_cast<unsigned>(c0) + prefix;
    __syncthreads();
    int it = 0;
    for (int base = c0; base < c1; base += kSortBatch * kSortThreads) {
      unsigned k[kSortBatch];
      int val[kSortBatch];
#pragma unroll
      for (int j = 0; j < kSortBatch; j++) {
        const int i = base + j * kSortThreads + t;
        k[j] = i < c1 ? coh_load(kin + i) : lo;
        val[j] = i < c1 ? coh_load(vin + i) : 0;
      }
#pragma unroll
      for (int j = 0; j < kSortBatch; j++) {
        if (base + j * kSortThreads >= c1) break;  // (uniform)
        sort_tile(l, it++, base + j * kSortThreads + t < c1, k[j], val[j], ((k[j] - lo) >> shift) & 255u, kout, vout);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
}

// workgroups for n keys: chunks of ~5000 keys (20 tiles), at most 64 (32 / 128 / 16 workgroups for a
// 320 k-entry block measured 8 / 13 / 35 % slower per FM step: every barrier needs them all resident)
static inline int sort_grid(int n, int cap) { return std::max(1, std::min(std::min(kSortMaxWgs, cap), (n + 4999) / 5000)); }
static inline size_t sort_scratch_bytes(size_t n) {
  return 8 * ((n + 63) & ~static_cast<size_t>(63)) + sizeof(unsigned) * 256 * kSortMaxWgs + 256;
}

}  // namespace ftrl_dev
