// engine.hip -- host side of the C ABI in include/ffm_engine.h: owns the model state in HBM and the
// per-block scratch, and schedules the gfx950 kernels of one block of rows on one HIP stream.
//
// Pipeline of one training block (DESIGN.md "Pipeline"):
//   group_keys -> radix sort (rocPRIM) -> group_finish              (kernels_group.h)
//   ffm_refresh           lazy refresh, once per distinct feature   (kernels_row.h)
//   {ffm,fm}_row<TRAIN>   lazy refresh + forward -> logit           (kernels_row.h)
//   [cross-shard sum of the logits when n_shards > 1 -- done by the caller, RCCL]
//   tmp_grad -> loss_sum
//   the (n, z) update of every touched accumulator, folded by        (kernels_fold.h, kernels_update.h,
//   reductions: bias, linear, {ffm,fm} latent                         kernels_tile.h)
// There is no CPU fallback anywhere in this library.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/ffm_engine.h"
#include "engine_types.h"
#include "init_rng.h"
#include "kernels_group.h"
#include "kernels_row.h"
#include "kernels_fold.h"
#include "kernels_update.h"
#include "kernels_tile.h"
#include "kernels_fm.h"
#include "kernels_predict.h"
#include "kernels_sort.h"

using namespace ftrl_dev;

// The grouping's sort: always the Onesweep radix sort (a few passes over the key bits), never
// rocPRIM's merge-sort variant (about twenty small launches at this size).
// (Its default tuning -- 1024-thread workgroups, twenty of them for a block -- only finds room when a
// kernel beside it drains; 256-thread configurations do find room and were measured slower for the
// step: the sort then takes wave slots from the kernels on the critical path.)
using GroupSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                  rocprim::default_config, 0>;

namespace {

thread_local std::string g_last_error;

// FFM_HOST_TIMING=1: wall time the submitting thread spends in labelled sections (debug aid)
struct HostTimers {
  bool on = std::getenv("FFM_HOST_TIMING") != nullptr;
  struct Acc { const char *name; double total = 0, worst = 0; long n = 0; } acc[16];
  int n_acc = 0;
  Acc &get(const char *name) {
    for (int i = 0; i < n_acc; i++) if (acc[i].name == name) return acc[i];
    acc[n_acc].name = name;
    return acc[n_acc < 15 ? n_acc++ : 15];
  }
  ~HostTimers() {
    if (!on) return;
    for (int i = 0; i < n_acc; i++)
      std::fprintf(stderr, "[host timing] %-22s n=%6ld total=%9.3f ms mean=%8.1f us worst=%9.1f us\n",
                   acc[i].name, acc[i].n, acc[i].total * 1e3, acc[i].total * 1e6 / std::max(1l, acc[i].n), acc[i].worst * 1e6);
  }
} g_timers;
struct ScopedTimer {
  const char *name;
  std::chrono::steady_clock::time_point t0;
  explicit ScopedTimer(const char *n) : name(n) { if (g_timers.on) t0 = std::chrono::steady_clock::now(); }
  ~ScopedTimer() {
    if (!g_timers.on) return;
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    static std::mutex mu;  // (the staging thread times its sections too)
    std::lock_guard<std::mutex> lock(mu);
    auto &a = g_timers.get(name);
    a.total += s; a.n++; if (s > a.worst) a.worst = s;
  }
};

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t err__ = (expr);                                                              \
    if (err__ != hipSuccess)                                                                \
      return fail(err__ == hipErrorOutOfMemory ? FFM_E_NOMEM : FFM_E_DEVICE,                \
                  std::string(#expr) + ": " + hipGetErrorString(err__));                    \
  } while (0)

enum KernelId {
  K_GROUP_KEYS, K_GROUP_SORT, K_GROUP_FINISH, K_ROW, K_TMP_GRAD,
  K_LOSS_SUM, K_LINEAR_UPDATE, K_BIAS_UPDATE, K_LATENT_UPDATE, K_LATENT_UPDATE_FEW, K_LATENT_UPDATE_WALK, K_LATENT_UPDATE_GIANT,
  K_PREDICT_ROW, K_REFRESH, K_LATENT_UPDATE_SINGLE,
  K_COUNT
};
const char *kKernelNames[K_COUNT] = {
    "group_keys_kernel", "group_radix_sort", "group_finish_kernel", "row_kernel<train>", "tmp_grad_kernel", "loss_sum_kernel",
    "linear_update_kernel", "bias_update_kernel", "update_kernel", "update_few_kernel", "update_walk_kernel", "update_giant_kernel",
    "row_kernel<predict>", "refresh_kernel", "update_single_kernel"};

struct ProfRec {
  int kid;
  hipEvent_t e0, e1;
};

// Where element e of the LOGICAL latent row of feature `feat` ([n_fields][n_factors] for FFM, the
// reference's layout, ffm.cpp:138-146) lives: a pointer to component `comp` of it, or null when
// this shard does not store it (compact storage keeps only the owned slots of the kept fields).
__device__ __forceinline__ float *logical_elem(const ModelDev &m, int64_t feat, int e, int comp) {
  if (!m.field_start)
    return m.lat + feat * 3 * m.row_len + static_cast<int64_t>(comp) * m.row_len + e;
  int lo = 0, hi = m.n_fields;  // field of the id: field_start[fa] <= feat < field_start[fa + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (m.field_start[mid] <= feat) lo = mid; else hi = mid;
  }
  const int fa = lo, fp = e / m.n_factors, kk = e - fp * m.n_factors;
  const int sl = fp - m.own_lo[fa];
  if (static_cast<unsigned>(sl) >= static_cast<unsigned>(m.own_n[fa])) return nullptr;
  return lat_row(m, static_cast<int>(feat), fa) + static_cast<int64_t>(comp) * m.row_len + sl * m.n_factors + kk;
}

// `logical_len` = n_fields * n_factors (FFM) or n_factors (FM): every draw is addressed by its
// logical coordinates, so a shard initialises its slots to the values the unsharded model holds.
__global__ void init_weights_kernel(ModelDev m, int logical_len, float mean, float stddev, uint64_t seed) {
  const int64_t n_lat = static_cast<int64_t>(m.n_feats) * logical_len;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < n_lat;
       idx += stride) {
    const int64_t feat = idx / logical_len;
    float *p = logical_elem(m, feat, static_cast<int>(idx - feat * logical_len), LAT_W);
    if (p) *p = ftrl_rng::init_weight(seed, 1, idx, mean, stddev);
  }
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < m.n_feats;
       i += stride)
    m.lin_w[i] = ftrl_rng::init_weight(seed, 0, i, mean, stddev);
}

__global__ void fill_state_kernel(ModelDev m, int logical_len, uint64_t seed, float n_lo, float n_hi, float z_sd) {
  const int64_t n_lat = static_cast<int64_t>(m.n_feats) * logical_len;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  auto unif = [&](uint64_t stream, int64_t idx) {
    return n_lo + (n_hi - n_lo) * ftrl_rng::uniform01(seed, stream, idx);
  };
  auto normal01 = [](uint64_t sd, uint64_t stream, int64_t idx) { return ftrl_rng::normal01(sd, stream, idx); };
  for (int64_t idx = t0; idx < n_lat; idx += stride) {
    const int64_t feat = idx / logical_len;
    const int e = static_cast<int>(idx - feat * logical_len);
    float *pn = logical_elem(m, feat, e, LAT_N);
    if (!pn) continue;
    *pn = unif(11, idx);
    pn[m.row_len] = z_sd * normal01(seed, 12, idx);  // the z row follows the n row
  }
  for (int64_t i = t0; i < m.n_feats; i += stride) {
    m.lin_n[i] = unif(13, i);
    m.lin_z[i] = z_sd * normal01(seed, 14, i);
  }
  if (t0 == 0) {
    m.bias3[1] = unif(15, 0);
    m.bias3[2] = z_sd * normal01(seed, 16, 0);
  }
}

// Proves div_alpha_fast == IEEE divide for this alpha over every float significand, both signs
// and two exponents (ftrl_math.h); any mismatch raises *bad.
__global__ void verify_div_alpha_kernel(Hyper h, int *bad) {
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;  // 2^25 threads
  const unsigned mant = idx & 0x7fffffu;
  const unsigned expo = (idx >> 23) & 1u ? 0x3f800000u : 0x40000000u;
  const unsigned sign = (idx >> 24) & 1u ? 0x80000000u : 0u;
  const float x = __uint_as_float(sign | expo | mant);
  const float a = div_alpha_fast(h, x), b = x / h.alpha;
  if (__float_as_uint(a) != __float_as_uint(b)) atomicOr(bad, 1);
}

// Compares sqrt_fast / sqrt_fast0 (ftrl_math.h) with sqrtf on EVERY float of their range,
// [2^-96, 2^96] and +0, on the device the engine runs on; any mismatch raises *bad.
__global__ void verify_sqrt_fast_kernel(int *bad) {
  const unsigned lo = __float_as_uint(0x1p-96f), hi = __float_as_uint(0x1p96f);
  const unsigned stride = gridDim.x * blockDim.x;
  bool differs = false;
  for (unsigned b = lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi && b >= lo; b += stride) {
    const float x = __uint_as_float(b);
    const unsigned want = __float_as_uint(sqrtf(x));
    differs = differs || __float_as_uint(sqrt_fast(x)) != want || __float_as_uint(sqrt_fast0(x)) != want;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) differs = differs || __float_as_uint(sqrt_fast0(0.0f)) != 0u;
  if (differs) atomicOr(bad, 1);
}

// Copies one component (n, z or w) of features [feat0, feat0+nf) between the stored records and a
// dense [feat][logical_len] staging buffer (the reference's save order).  Slots this shard does
// not store read as 0 and ignore writes.
__global__ void lat_component_copy_kernel(ModelDev m, int logical_len, int comp, float *dense,
                                          int64_t feat0, int64_t nf, int to_dense) {
  const int64_t total = nf * logical_len;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    const int64_t f = idx / logical_len;
    float *p = logical_elem(m, feat0 + f, static_cast<int>(idx - f * logical_len), comp);
    if (to_dense) dense[idx] = p ? *p : 0.0f; else if (p) *p = dense[idx];
  }
}

// The same for a list of features: component `comp` of ids[j] <-> dense[j][logical_len]
// (ffm_engine_get_rows / set_rows).
__global__ void lat_rows_copy_kernel(ModelDev m, int logical_len, int comp, float *dense,
                                     const int *ids, int64_t nf, int to_dense) {
  const int64_t total = nf * logical_len;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    const int64_t j = idx / logical_len;
    const int i = ids[j];
    float *p = (i < 0 || i >= m.n_feats) ? nullptr
                                          : logical_elem(m, i, static_cast<int>(idx - j * logical_len), comp);
    if (to_dense) dense[idx] = p ? *p : 0.0f; else if (p) *p = dense[idx];
  }
}
__global__ void lin_rows_copy_kernel(float *lin, int n_feats, float *dense, const int *ids, int nf,
                                     int to_dense) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const int i = ids[j];
  if (i < 0 || i >= n_feats) { if (to_dense) dense[j] = 0.0f; return; }
  if (to_dense) dense[j] = lin[i]; else lin[i] = dense[j];
}

#include "engine_plan.h"
}  // namespace

struct ffm_engine {
  ffm_engine_config cfg{};
  ModelDev m{};
  // kSets sets of grouping outputs, used in rotation: block t+1 is grouped on the prep stream into
  // the set block t-2 used, so it never waits for block t-1 or t to retire (with two sets the
  // grouping could only start when the previous block ended, and then sat starved of wave slots
  // behind that block's persistent update kernels).  What the row kernel writes is shared.
  static constexpr int kSets = 4;
  Scratch sc[kSets]{};
  int cur = 0;                 // set of the block being trained
  bool cur_prepared = false;   // ... was grouped ahead (ev_grouped[cur] marks the end of its grouping)
  int *h_super = nullptr;      // [kSets] page-locked: Scratch::n_super of every set, as the host reads it
  // evaluation block whose predict launch ffm_engine_predict_batch_async deferred by one call
  struct { bool on = false; int slot = 0; bool labelled = false; } eval_pending;
  bool eval_hold = false;       // inside predict_batch_async, before its upload is submitted
  bool eval_defer_off = false;  // FFM_EVAL_DEFER=0
  bool super_wait = true;       // wait for a block's grouping before deciding on its super launches (FFM_SUPER_WAIT=0: only ask)
  bool super_flag_ok = false;  // create proved that a device store to h_super reaches the host
  // groupings made ahead by ffm_engine_prepare_device, oldest first (at most kSets - 1)
  int n_prepared = 0;
  int prepared_set[kSets] = {};
  Rows prepared_rows[kSets] = {};
  int last_set = 0;            // set handed out most recently (to a prepare or to a training block)
  bool set_used[kSets] = {};
  hipStream_t prep = nullptr;  // side stream of ffm_engine_prepare_device
  // Uploads of staged host blocks (pull_block_kernel) go on the prep stream, ahead of the block's
  // grouping.  (A fifth stream for them shares a hardware queue with one of the four and
  // serialises with it: measured 1.45-1.70 ms per step instead of 1.3.)
  bool pull_after_row = false;  // a staged block's upload waits for the running block's row kernel to end (long steps)
  hipStream_t copy = nullptr;  // the upload kernel's stream: the prep stream (round 4 had it on aux3 for long steps)
  bool own_sort = false;        // the grouping's sort: kernels_sort.h (short steps) or rocPRIM Onesweep (ffm_engine_create)
  bool range_sort = false;      // ... or, when the id ranges of the fields are known, a workgroup per range (kernels_sort.h)
  int *d_sort_start = nullptr;  // [n_sort_ranges + 1] the ranges' boundaries, the last one = n_feats
  int n_sort_ranges = 0;
  int sort_grid_cap = 1;        // workgroups of the one-launch sort the device holds at once (its grid barrier needs them all)
  hipEvent_t ev_grouped[kSets] = {}, ev_set_free[kSets] = {};
  hipEvent_t ev_row_done[kSets] = {};  // the row kernel of the block trained from the set has ended
  bool prep_after_row = false;         // a look-ahead grouping starts beside an UPDATE phase (FFM_PREP_AFTER_ROW)
  // Scheduling of a look-ahead grouping: the block being
  // prepared will start training when its predecessor's update ends; its grouping is made to
  // START when the block before that one ends -- so it runs beside the predecessor's refresh and
  // row phases (memory-bound, ~400 us, the grouping takes ~300 us there) and never beside an
  // update phase, where the update's kernels starve it (a Onesweep pass then takes 250-300 us
  // instead of 7) and it slows them.  trained_set[0] / [1]: the scratch sets of the
  // training blocks enqueued last / before that (their ev_set_free marks the end of the update).
  bool prep_window = true;
  int trained_set[2] = {-1, -1};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // Streams: the runtime multiplexes streams onto few hardware queues (4 by default), and two
  // streams on one queue run one after the other -- so no more than stream + 2 side + prep here.
  hipStream_t aux3 = nullptr;  // side stream: a shard's once-only / few-occurrence launches beside its update
                               // launch; the few-occurrence launch of a side-by-side update
  hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_join = nullptr, ev_join2 = nullptr;
  hipStream_t aux4 = nullptr;  // second side stream of the update (FFM_UPDATE_SPLIT=2)
  int max_rows = 0, max_nnz = 0, max_row_nnz = 1024;
  // longest row of the block being staged from host memory (known there; 0 = unknown: device
  // callers).  The row kernels size their LDS by it, which decides how many rows a CU holds.
  int staged_row_cap = 0;
  // staging for the host-buffer entry points
  int *d_row_ptr = nullptr, *d_field = nullptr, *d_feat = nullptr, *d_label = nullptr;
  float *d_val = nullptr, *d_out = nullptr;
  double *d_loss_sum = nullptr;
  // pipelined host-buffer training (ffm_engine_train_batch_async): kSlots staging slots, each a
  // pinned host image and device arrays of one block; block t is copied + grouped on the prep
  // stream while block t-1 trains on the main stream and the caller builds block t+1
  static constexpr int kSlots = 4;
  struct Slot {
    char *pinned = nullptr;
    int *row_ptr = nullptr, *field = nullptr, *feat = nullptr, *label = nullptr;
    float *val = nullptr;
    hipEvent_t ev_copied = nullptr, ev_trained = nullptr;
    // what frees the slot's device arrays: the scratch set's ev_set_free of the block that trained from
    // it (no record of its own: one packet less on the main stream per step), ev_trained after a prediction
    hipEvent_t free_ev = nullptr;
    bool used = false, zero_copy = false;
    int64_t seq = 0;  // 1-based number of the block staged in it
    int n_rows = 0, nnz = 0, row_cap = 0;
    bool has_field = false;
  } slots[kSlots];
  bool slots_ready = false;
  int slot_next = 0;
  int staged[kSlots] = {};  // slots staged and not yet in training, oldest first
  int n_staged = 0;
  int cur_slot = -1;        // slot of the block between train_forward_staged and train_update
  int64_t n_staged_total = 0, n_pulled = 0;  // blocks staged so far / known to be uploaded
  // The upload kernel of block number b (1-based; uploads run in staging order) writes b into a
  // word of page-locked host memory when its last workgroup is done: ffm_engine_blocks_pulled is a
  // plain read.  (Per staging slot would not do -- a slot is refilled, on the HOST's timeline, while
  // the GPU may still be several blocks behind --, and an event per block costs the submitting
  // thread a query per poll and the runtime a signal per block.)
  long long *h_pulled = nullptr;       // hipHostMalloc
  long long *d_pulled = nullptr;       // the same word as the device sees it
  unsigned *d_pull_ticket = nullptr;   // workgroups of the running upload kernel that are done
  double *d_loss_acc = nullptr;  // [1] sum of the async blocks' losses since the last flush
  double *d_loss_part = nullptr; // [kLossParts + 1] loss_sum_kernel's partial sums and ticket
  void *d_sort_tmp[kSets] = {};  // rocPRIM radix sort workspace per scratch set
  size_t sort_tmp_bytes = 0;
  unsigned sort_bits = 32;
  float *d_stage = nullptr;  // dense staging for get/set
  int64_t stage_floats = 0;
  int *d_ids = nullptr;      // feature ids of one ffm_engine_get_rows / set_rows chunk
  static constexpr int kIdsCap = 1 << 20;
  int *d_err = nullptr;      // [1] sticky ERR_* flags of every block since the last report
  std::vector<void *> allocs;
  // split-phase bookkeeping
  Rows pending{};
  bool has_pending = false;
  bool whole_step = false;  // the call in flight is train_batch_device (forward + update in one)
  // train_batch_device on one shard: the row kernel has the whole logit, so it also produces
  // tmp_grad and the row losses (no tmp_grad pass)
  bool own_tg_cur = false;
  float *own_logit_out = nullptr;
  // the block's lazy refresh as one pass over its distinct features (ffm_refresh_kernel) instead
  // of per occurrence inside the row kernel; FFM_ENGINE_ROW_REFRESH=1 keeps it in the row kernel
  bool pre_refresh = false;
  // 1: every distinct feature in ffm_refresh_kernel; 2: the once-only ones by their row; 3: and
  // updated there too when the row kernel has the whole logit (FFM_ENGINE_ROW_REFRESH=0/2/3)
  int refresh_mode = 3;
  bool singles_in_row = false;  // the block in flight had its once-only features updated by the row kernel
  unsigned long long *d_ownmask = nullptr;
  bool lin_any = true;          // this shard owns the linear terms of at least one field
  int logical_len = 0;          // n_fields * n_factors (FFM), n_factors (FM), 0 (LR): the reference's row
  int64_t n_records = 0;        // stored latent records (n_feats, or fewer on a compact shard)
  std::vector<int> field_start; // copy of cfg->field_start (compact shards)
  // workgroups of the update launch's ranges: few-occurrence features, hot tiles, the row-order walk
  // of serial slots; and of the once-only kernel of a shard
  int grid_small = 768, grid_hot = 2048, grid_walk = 256, grid_single = 768, grid_giant = 1024;
  int update_split = -1;  // FFM_UPDATE_SPLIT: 0 the update's ranges as ONE launch; 2 as three launches side by side
                          // on three queues (hot + bias + linear | few + serial walk + loss | giant), each with the
                          // registers and LDS of its own path; 1 as launches one after another (timing aid);
                          // default: 2 for a whole model's large launches (nnz * k >= 4 M: the fork / join pays), else 0
  int wide_max_nnz = 100000;  // FFM_WIDE_NNZ: blocks below this many entries run the update launch with eight-wave workgroups (0: never)
  bool predict_waves = true;  // FFM_PREDICT_WAVE=0: evaluation rows through ffm_row_kernel (one workgroup per row)
  int update_order = 210;     // FFM_UPDATE_ORDER: the update launch's big ranges, first range = last digit (0 giant, 1 hot, 2 few)
  // workgroups of pull_block_kernel.  Few on purpose: its loads take microseconds
  // (PCIe) and every one in flight holds a miss entry of an L2; 128 workgroups (512 KB in flight)
  // slowed the HBM-bound kernels they ran beside up to 4x (refresh of a 65536-row block 70 -> 330 us),
  // 24 (96 KB, about the link's bandwidth-delay product) still move the block at link rate.
  int grid_pull = 24;
  bool single_kernel = true;  // (false: once-only features through the few-occurrence kernel)
  bool single_flat = true;    // (false: one wave per feature also for short stored records)
  int row_threads = kRowThreads;  // workgroup size of the FFM row kernel (FFM_ROW_THREADS)
  bool serial = false;  // FFM_ENGINE_SERIAL=1: no side streams (per-kernel timings without overlap)
  // ---- staging thread ------------------------------------------------------------------------
  // The GPU submissions of a staged block (its upload kernel and the ~10 launches of its grouping,
  // all on the prep stream) are issued by a thread of the engine's own, two blocks ahead of the
  // training launches that the caller's thread issues.  Why: rocPRIM's radix sort calls
  // hipMemsetAsync inside, which on this runtime can hold the submitting thread until earlier work
  // of the stream has retired -- measured 30 us per sort call on most boxes of the pool but 240 us
  // mean / 31 ms worst on others (profiles/archive/r03_host_leg_diag.txt), during which the caller's thread
  // did not enqueue the NEXT training block and the H2D-inclusive step grew from 1.13 to 1.32 ms
  // with the resident step unchanged.  All bookkeeping stays on the caller's thread; the worker only
  // replays closures in order.  staged_issued: ordinal of the last staged block whose launches are
  // out (a training call waits for ITS block only, never for the look-ahead's).
  bool stage_thread_on = true;  // FFM_STAGE_THREAD=0: the caller's thread submits everything
  std::thread worker;
  std::mutex wmu;
  std::condition_variable wcv_job, wcv_done;
  std::deque<std::function<int()>> wjobs;
  bool wstop = false, wbusy = false;
  int64_t staged_issued = 0;
  int worker_rc = 0;
  std::string worker_msg;
  void worker_main() {
    (void)hipSetDevice(cfg.device_id);
    for (;;) {
      std::function<int()> job;
      {
        std::unique_lock<std::mutex> lock(wmu);
        wcv_job.wait(lock, [&] { return wstop || !wjobs.empty(); });
        if (wjobs.empty()) return;
        job = std::move(wjobs.front());
        wjobs.pop_front();
        wbusy = true;
      }
      const int rc = job();
      {
        std::lock_guard<std::mutex> lock(wmu);
        if (rc != 0 && worker_rc == 0) { worker_rc = rc; worker_msg = g_last_error; }
        wbusy = false;
      }
      wcv_done.notify_all();
    }
  }
  // Runs `job` on the staging thread (or right here when it is off / while kernels are being timed).
  int submit(std::function<int()> job) {
    if (!stage_thread_on || prof_on) return job();
    {
      std::lock_guard<std::mutex> lock(wmu);
      if (!worker.joinable()) worker = std::thread([this] { worker_main(); });
      wjobs.push_back(std::move(job));
    }
    wcv_job.notify_one();
    return 0;
  }
  int worker_error() {  // an error of an earlier deferred submission surfaces at the next wait
    if (worker_rc == 0) return 0;
    const int rc = worker_rc;
    worker_rc = 0;
    return fail(rc, "staging thread: " + worker_msg);
  }
  // Until everything submitted so far is issued (before the caller's thread touches the prep stream).
  int drain() {
    std::unique_lock<std::mutex> lock(wmu);
    wcv_done.wait(lock, [&] { return wjobs.empty() && !wbusy; });
    return worker_error();
  }
  // Until the launches of staged block number `seq` are issued.
  int wait_issued(int64_t seq) {
    std::unique_lock<std::mutex> lock(wmu);
    wcv_done.wait(lock, [&] { return staged_issued >= seq || (wjobs.empty() && !wbusy); });
    return worker_error();
  }
  void stop_worker() {
    {
      std::lock_guard<std::mutex> lock(wmu);
      wstop = true;
    }
    wcv_job.notify_all();
    if (worker.joinable()) worker.join();
  }
  // profiling
  bool prof_on = false;
  int prof_only = -1;  // >= 0: record only this kernel id (ffm_engine_profile_focus)
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> event_pool;

  template <typename T>
  int alloc(T **p, size_t count) {
    void *q = nullptr;
    if (count == 0) count = 1;
    hipError_t err = hipMalloc(&q, count * sizeof(T));
    if (err != hipSuccess)
      return fail(FFM_E_NOMEM, "hipMalloc of " + std::to_string(count * sizeof(T)) +
                                   " bytes failed: " + hipGetErrorString(err));
    allocs.push_back(q);
    *p = static_cast<T *>(q);
    return FFM_OK;
  }

  hipEvent_t get_event() {
    if (!event_pool.empty()) {
      hipEvent_t e = event_pool.back();
      event_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  void prof_begin(int kid, hipStream_t st) {
    prof_skip = !prof_on || (prof_only >= 0 && kid != prof_only);
    if (prof_skip) return;
    ProfRec r{kid, get_event(), get_event()};
    if (!r.e0 || !r.e1) {  // out of events: this launch goes untimed
      if (r.e0) event_pool.push_back(r.e0);
      if (r.e1) event_pool.push_back(r.e1);
      prof_skip = true;
      return;
    }
    (void)hipEventRecord(r.e0, st);
    prof.push_back(r);
  }
  bool prof_skip = true;
  void prof_end(hipStream_t st) {
    if (prof_skip) return;
    (void)hipEventRecord(prof.back().e1, st);
  }
};

#define LAUNCH_ON(e, st, kid, kernel, grid, block, shmem, ...)                   \
  do {                                                                           \
    (e)->prof_begin(kid, st);                                                    \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, st, __VA_ARGS__); \
    (e)->prof_end(st);                                                           \
  } while (0)
#define LAUNCH(e, kid, kernel, grid, block, shmem, ...) \
  LAUNCH_ON(e, (e)->stream, kid, kernel, grid, block, shmem, __VA_ARGS__)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int loss_grid(int n_rows) { return std::max(1, std::min(kLossParts, cdiv(n_rows, 1024))); }

// The update launch (kernels_tile.h) is instantiated per number of facts a stager lane of the hot
// features' tiles carries: slots per 64-element chunk / 4.
static int tile_nf(const ffm_engine *e) {
  const int k = e->m.n_factors, spc = k <= 64 ? 64 / k : 1;
  return spc > 8 ? 4 : spc > 4 ? 2 : 1;
}
// The once-only / few-occurrence kernels over a flat (feature, vector) index space instead of a
// wave per feature: when a wave per feature would leave more than a sixth of its lanes idle on a
// compact shard's records (40 vectors at 8 shards, 80 at 2 or 4).  For full-length records (156
// vectors, 81 % lane use) the flat shape measured no better.
static bool flat_pays(const ffm_engine *e, int span4) {
  if (!e->single_flat || e->m.n_shards <= 1 || span4 <= 0) return false;
  const int lanes = (span4 + 63) / 64 * 64;
  return span4 * 6 < lanes * 5;
}

extern "C" {

int ffm_engine_abi_version(void) { return FFM_ENGINE_ABI_VERSION; }
int ffm_engine_block_segment(void) { return ftrl_dev::kSeg; }
const char *ffm_engine_last_error(void) { return g_last_error.c_str(); }

void ffm_engine_default_config(ffm_engine_config *cfg) {
  if (!cfg) return;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->model_type = FFM_MODEL_FFM;  // cmd_option.h:62
  cfg->n_feats = 10000;
  cfg->n_fields = 8;
  cfg->n_factors = 16;
  cfg->w_alpha = 1e-4f;
  cfg->w_beta = 1.0f;
  cfg->w_l1 = 0.1f;
  cfg->w_l2 = 5.0f;
  cfg->init_mean = 0.0f;
  cfg->init_stddev = 0.02f;
  cfg->seed = 42;
  cfg->max_batch_rows = 8192;
  cfg->max_batch_nnz = 8192 * 64;
  cfg->device_id = 0;
  cfg->n_shards = 1;
  cfg->shard_rank = 0;
}

int64_t ffm_engine_row_len(const ffm_engine *e) { return e ? e->logical_len : 0; }

void ffm_engine_destroy(ffm_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device_id);
  e->stop_worker();  // (issues what is still queued, then joins)
  // everything the engine has in flight -- uploads still reading the caller's page-locked arrays
  // (staged, never trained), look-ahead groupings, the side streams -- ends before anything is freed
  if (e->prep) (void)hipStreamSynchronize(e->prep);
  if (e->aux3) (void)hipStreamSynchronize(e->aux3);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_fork2) (void)hipEventDestroy(e->ev_fork2);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_join2) (void)hipEventDestroy(e->ev_join2);
  if (e->aux4) (void)hipStreamDestroy(e->aux4);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (auto &r : e->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto ev : e->event_pool) (void)hipEventDestroy(ev);
  for (auto &sl : e->slots) {
    if (sl.pinned) (void)hipHostFree(sl.pinned);
    if (sl.ev_copied) (void)hipEventDestroy(sl.ev_copied);
    if (sl.ev_trained) (void)hipEventDestroy(sl.ev_trained);
  }
  for (void *p : e->allocs) (void)hipFree(p);
  for (int i = 0; i < ffm_engine::kSets; i++) {
    if (e->ev_grouped[i]) (void)hipEventDestroy(e->ev_grouped[i]);
    if (e->ev_set_free[i]) (void)hipEventDestroy(e->ev_set_free[i]);
    if (e->ev_row_done[i]) (void)hipEventDestroy(e->ev_row_done[i]);
  }
  if (e->h_pulled) (void)hipHostFree(e->h_pulled);
  if (e->h_super) (void)hipHostFree(e->h_super);
  if (e->prep) (void)hipStreamDestroy(e->prep);

  if (e->aux3) (void)hipStreamDestroy(e->aux3);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int ffm_engine_create(const ffm_engine_config *cfg, ffm_engine **out) {
  if (!cfg || !out) return fail(FFM_E_INVALID, "null config or output pointer");
  *out = nullptr;
  if (cfg->model_type < FFM_MODEL_LR || cfg->model_type > FFM_MODEL_FFM)
    return fail(FFM_E_INVALID, "invalid model_type, expect LR(0), FM(1) or FFM(2)");
  if (cfg->n_feats <= 0) return fail(FFM_E_INVALID, "n_feats must be positive");
  if (cfg->model_type == FFM_MODEL_FFM && cfg->n_fields <= 0)
    return fail(FFM_E_INVALID, "n_fields must be positive for FFM");
  if (cfg->model_type != FFM_MODEL_LR && cfg->n_factors <= 0)
    return fail(FFM_E_INVALID, "n_factors must be positive for FM/FFM");
  if (cfg->model_type == FFM_MODEL_FM && cfg->n_factors > kTermsCap)
    return fail(FFM_E_UNSUPPORTED, "FM n_factors above 2048 is not supported");
  if (cfg->max_batch_rows <= 0 || cfg->max_batch_nnz <= 0)
    return fail(FFM_E_INVALID, "max_batch_rows / max_batch_nnz must be positive");
  if (cfg->n_shards < 1 || cfg->shard_rank < 0 || cfg->shard_rank >= cfg->n_shards)
    return fail(FFM_E_INVALID, "invalid n_shards / shard_rank");
  if (cfg->n_shards > 1 && cfg->model_type != FFM_MODEL_FFM)
    return fail(FFM_E_UNSUPPORTED, "field-pair sharding applies to FFM only");
  if (cfg->n_shards > 1 && cfg->n_fields > 64)
    return fail(FFM_E_UNSUPPORTED, "field-pair sharding supports up to 64 fields");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(FFM_E_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (cfg->device_id < 0 || cfg->device_id >= n_dev)
    return fail(FFM_E_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(cfg->device_id));

  ffm_engine *e = new (std::nothrow) ffm_engine;
  if (!e) return fail(FFM_E_NOMEM, "host allocation failed");
  e->cfg = *cfg;
  e->max_rows = cfg->max_batch_rows;
  e->max_nnz = cfg->max_batch_nnz;
  if (cfg->max_row_nnz > 0) e->max_row_nnz = cfg->max_row_nnz;
  if (const char *sv = std::getenv("FFM_ENGINE_SERIAL")) e->serial = sv[0] == '1';
  if (const char *sv = std::getenv("FFM_UPDATE_SPLIT")) e->update_split = std::atoi(sv);
  if (const char *sv = std::getenv("FFM_UPDATE_ORDER")) {
    // a permutation of the digits {0, 1, 2} (first range last) or nothing: any other value would run
    // one range of the update launch twice and drop another
    const int o = std::atoi(sv);
    const int d0 = o % 10, d1 = o / 10 % 10, d2 = o / 100;
    if (o >= 0 && d2 <= 2 && d0 <= 2 && d1 <= 2 && d0 != d1 && d0 != d2 && d1 != d2) e->update_order = o;
  }
  if (const char *sv = std::getenv("FFM_SUPER_WAIT")) e->super_wait = sv[0] != '0';
  if (const char *sv = std::getenv("FFM_PREP_AFTER_ROW")) e->prep_after_row = sv[0] == '1';
  if (const char *sv = std::getenv("FFM_EVAL_DEFER")) e->eval_defer_off = sv[0] == '0';
  if (const char *sv = std::getenv("FFM_PREDICT_WAVE")) e->predict_waves = std::atoi(sv) != 0;
  if (const char *sv = std::getenv("FFM_WIDE_NNZ")) e->wide_max_nnz = std::atoi(sv);
  if (const char *sv = std::getenv("FFM_GRID_PULL")) e->grid_pull = std::max(1, std::atoi(sv));
  if (const char *sv = std::getenv("FFM_GRID_HOT")) e->grid_hot = std::max(1, std::atoi(sv));
  if (const char *sv = std::getenv("FFM_GRID_SMALL")) e->grid_small = std::max(1, std::atoi(sv));
  if (const char *sv = std::getenv("FFM_GRID_WALK")) e->grid_walk = std::max(1, std::atoi(sv));
  if (const char *sv = std::getenv("FFM_GRID_GIANT")) e->grid_giant = std::max(1, std::atoi(sv));
  // (a sharded rank's row holds 1/n_shards of the pairs and facts, walked directly; 64-thread row
  // workgroups measured the same as 256)
  // a shard's row holds ~1/n_shards of the pairs: one wave per row, so that (with the small LDS
  // footprint of short rows) many more rows are in flight per CU
  if (cfg->n_shards > 1) e->row_threads = 64;
  if (const char *sv = std::getenv("FFM_ROW_THREADS")) e->row_threads = std::max(64, std::min(kRowMaxThreads, std::atoi(sv) / 64 * 64));
  // (the lean once-only kernel of a compact shard holds six waves per SIMD: 1152 workgroups
  // measured 2.5 % per step better than 768 on an 8-GPU rank's blocks)
  if (cfg->n_shards > 1) e->grid_single = 1152;
  // (... and its flat few-occurrence kernel, 126 VGPRs = four waves per SIMD, fills the chip at 1024
  // workgroups: an emulated rank's step 1.615 -> 1.605 ms)
  if (cfg->n_shards > 1 && !std::getenv("FFM_GRID_SMALL")) e->grid_small = 1024;

  {
    const char *rr = std::getenv("FFM_ENGINE_ROW_REFRESH");
    const int64_t per = cfg->n_factors % 4 == 0 ? static_cast<int64_t>(cfg->n_fields) * cfg->n_factors / 4
                                                : static_cast<int64_t>(cfg->n_fields) * cfg->n_factors;
    e->pre_refresh = cfg->model_type == FFM_MODEL_FFM && cfg->n_fields <= 64 &&
                     static_cast<int64_t>(e->max_nnz) * per < (1ll << 31) && !(rr && rr[0] == '1');
    if (rr && rr[0] == '0') e->refresh_mode = 1;
    if (rr && rr[0] == '2') e->refresh_mode = 2;
    if (rr && rr[0] == '3') e->refresh_mode = 3;
  }
  ModelDev &m = e->m;
  m.type = cfg->model_type;
  m.n_feats = cfg->n_feats;
  m.n_fields = cfg->model_type == FFM_MODEL_FFM ? cfg->n_fields : 1;
  m.n_factors = cfg->model_type == FFM_MODEL_LR ? 0 : cfg->n_factors;
  e->logical_len = cfg->model_type == FFM_MODEL_FFM ? cfg->n_fields * cfg->n_factors
                   : cfg->model_type == FFM_MODEL_FM ? cfg->n_factors : 0;
  m.n_shards = cfg->n_shards;
  m.shard_rank = cfg->shard_rank;
  m.bias_own = 1;
  m.huge_min = kHugeMin;
  m.giant_min = cfg->model_type == FFM_MODEL_FM ? kFmGiantMin : kGiantMin;
  m.range_len = cfg->model_type == FFM_MODEL_FM ? kFmRange : kRange;
  m.huge_min = std::min(m.huge_min, m.giant_min - 1);  // (the lists: big <= huge_min < huge < giant_min <= giant)
  m.super_min = kSuperMin;
  if (const char *sv = std::getenv("FFM_SUPER_MIN")) m.super_min = std::max(kGiantMin, std::atoi(sv));
  m.rec_slots = m.n_fields;
  e->n_records = cfg->n_feats;
  // field-pair partition: this shard's ranges, and (with per-field id ranges) compact storage
  ShardPlan plan;
  const bool compact = cfg->n_shards > 1 && cfg->field_start != nullptr;
  if (cfg->field_start && cfg->model_type == FFM_MODEL_FFM) {
    e->field_start.assign(cfg->field_start, cfg->field_start + cfg->n_fields + 1);
    bool ok = e->field_start[0] == 0 && e->field_start[cfg->n_fields] == cfg->n_feats;
    for (int f = 0; f < cfg->n_fields; f++) ok = ok && e->field_start[f] <= e->field_start[f + 1];
    if (!ok) { delete e; return fail(FFM_E_INVALID, "field_start must ascend from 0 to n_feats"); }
  }
  if (cfg->n_shards > 1) {
    plan = make_shard_plan(cfg->n_fields, cfg->n_shards, compact);
    m.bias_own = plan.bias_owner == cfg->shard_rank ? 1 : 0;
    if (compact) {
      int widest = 1;
      for (int f = 0; f < cfg->n_fields; f++) widest = std::max(widest, plan.n(cfg->shard_rank, f));
      m.rec_slots = widest;
      e->n_records = 0;
      for (int f = 0; f < cfg->n_fields; f++)
        if (plan.n(cfg->shard_rank, f) > 0) e->n_records += e->field_start[f + 1] - e->field_start[f];
      if (e->n_records == 0) e->n_records = 1;
    }
  }
  m.row_len = cfg->model_type == FFM_MODEL_FFM ? m.rec_slots * cfg->n_factors
              : cfg->model_type == FFM_MODEL_FM ? cfg->n_factors : 0;
  m.h = Hyper{cfg->w_alpha, cfg->w_beta, cfg->w_l1, cfg->w_l2, 1.0f / cfg->w_alpha, 0,
              (cfg->flags & FFM_FLAG_LEARN) ? 1 : 0};
  if (static_cast<int64_t>(cfg->n_fields) * cfg->n_factors > (1 << 24))
    { delete e; return fail(FFM_E_UNSUPPORTED, "n_fields*n_factors too large"); }
  if (static_cast<int64_t>(cfg->max_batch_nnz) / (kSmallMax + 1) * ((m.row_len + 7) / 8 + 1) >= (1ll << 31))
    { delete e; return fail(FFM_E_UNSUPPORTED, "max_batch_nnz * row_len too large for 32-bit work-item ids"); }

  int rc = FFM_OK;
#define TRY_ALLOC(call) do { rc = (call); if (rc != FFM_OK) { ffm_engine_destroy(e); return rc; } } while (0)
#define TRY_HIP(expr) do { hipError_t err__ = (expr); if (err__ != hipSuccess) { \
    rc = fail(FFM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(err__)); \
    ffm_engine_destroy(e); return rc; } } while (0)
  if (cfg->stream) {
    e->stream = static_cast<hipStream_t>(cfg->stream);
  } else {
    TRY_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
  }
  // (priority streams for the hot / very hot update: measured +45 % per step; for the look-ahead
  // grouping: the same)
  TRY_HIP(hipStreamCreateWithFlags(&e->aux3, hipStreamNonBlocking));
  TRY_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  TRY_HIP(hipEventCreateWithFlags(&e->ev_fork2, hipEventDisableTiming));
  TRY_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
  TRY_HIP(hipEventCreateWithFlags(&e->ev_join2, hipEventDisableTiming));
  TRY_HIP(hipStreamCreateWithFlags(&e->aux4, hipStreamNonBlocking));
  const size_t nf = static_cast<size_t>(cfg->n_feats);
  const size_t n_lat = static_cast<size_t>(e->n_records) * 3 * static_cast<size_t>(m.row_len);
  TRY_ALLOC(e->alloc(&m.bias3, 4));
  TRY_ALLOC(e->alloc(&m.lin_n, nf));
  TRY_ALLOC(e->alloc(&m.lin_z, nf));
  TRY_ALLOC(e->alloc(&m.lin_w, nf));
  TRY_ALLOC(e->alloc(&m.lat, n_lat));
  const size_t R = static_cast<size_t>(e->max_rows), E = static_cast<size_t>(e->max_nnz);
  Scratch &s = e->sc[0];
  TRY_ALLOC(e->alloc(&s.key, E));
  TRY_ALLOC(e->alloc(&s.skey, E));
  TRY_ALLOC(e->alloc(&s.row_of, E));
  TRY_ALLOC(e->alloc(&s.occ, E));
  TRY_ALLOC(e->alloc(&s.occ2, E));
  TRY_ALLOC(e->alloc(&s.udesc, E));
  TRY_ALLOC(e->alloc(&s.small, E));
  TRY_ALLOC(e->alloc(&s.few, E));
  TRY_ALLOC(e->alloc(&s.sdesc, E));
  TRY_ALLOC(e->alloc(&s.big, E));
  TRY_ALLOC(e->alloc(&s.huge, E));
  TRY_ALLOC(e->alloc(&s.giant, E / kChainMin + 1));
  TRY_ALLOC(e->alloc(&s.gseg, E / kChainMin + 1));
  TRY_ALLOC(e->alloc(&s.grange, E / kFmRange + E / kFmGiantMin + 2));
  {
    // partial sums of the giant features' folds (shared by the scratch sets)
    const size_t max_segs = E / kSeg + E / kFmGiantMin + 2, rl = static_cast<size_t>(std::max(1, m.row_len));
    const bool lat_model = cfg->model_type != FFM_MODEL_LR, ffm_m = cfg->model_type == FFM_MODEL_FFM;
    const size_t chunks = !ffm_m ? 1 : cfg->n_factors <= 64
                              ? (static_cast<size_t>(m.rec_slots) + 64 / cfg->n_factors - 1) / (64 / cfg->n_factors)
                              : static_cast<size_t>(m.rec_slots) * ((cfg->n_factors + 63) / 64);
    TRY_ALLOC(e->alloc(&s.segP, lat_model ? max_segs * rl : 1));
    TRY_ALLOC(e->alloc(&s.segG, lat_model ? max_segs * rl : 1));
    TRY_ALLOC(e->alloc(&s.segD, ffm_m ? max_segs * rl : 1));
    TRY_ALLOC(e->alloc(&s.segF, ffm_m ? max_segs * chunks * 3 : 1));
    TRY_ALLOC(e->alloc(&s.gcap, ffm_m ? (E / kGiantMin + 2) * rl : 1));
  }
  TRY_ALLOC(e->alloc(&s.counters, kNumCounters));
  TRY_ALLOC(e->alloc(&e->d_err, 1));
  TRY_HIP(hipMemsetAsync(e->d_err, 0, sizeof(int), e->stream));
  s.err = e->d_err;
  TRY_ALLOC(e->alloc(&e->d_ids, static_cast<size_t>(ffm_engine::kIdsCap)));
  TRY_ALLOC(e->alloc(&s.head, R * static_cast<size_t>(m.n_fields)));
  TRY_ALLOC(e->alloc(&s.next, E));
  TRY_ALLOC(e->alloc(&s.rowtab, R * static_cast<size_t>(m.n_fields)));
  TRY_ALLOC(e->alloc(&s.occpos, E));
  TRY_ALLOC(e->alloc(&s.uflag, E));
  const bool ffm_model = m.type == FFM_MODEL_FFM;
  const bool masks = ffm_model && m.n_fields <= 64;
  if (masks) {
    TRY_ALLOC(e->alloc(&s.rowmask, 2 * R));
    TRY_ALLOC(e->alloc(&s.gmask, E));
    TRY_ALLOC(e->alloc(&s.cmask, E));
    std::vector<unsigned long long> own(m.n_fields, 0ull);
    for (int fa = 0; fa < m.n_fields; fa++)
      for (int fb = 0; fb < m.n_fields; fb++)
        if (m.n_shards <= 1 || plan.owns(m.shard_rank, fa, fb)) own[fa] |= 1ull << fb;
    TRY_ALLOC(e->alloc(&e->d_ownmask, static_cast<size_t>(m.n_fields)));
    TRY_HIP(hipMemcpy(e->d_ownmask, own.data(), own.size() * sizeof(own[0]), hipMemcpyHostToDevice));
    m.ownmask = e->d_ownmask;
  }
  if (m.n_shards > 1) {
    // this shard's ranges (and, compact, the record index of every kept field) on the device
    const int F = m.n_fields, r = m.shard_rank;
    std::vector<int> lo(F), cnt(F), lin(F);
    std::vector<long long> base(F, -1);
    long long next = 0;
    for (int f = 0; f < F; f++) {
      lo[f] = plan.lo(r, f);
      cnt[f] = plan.n(r, f);
      lin[f] = plan.lin_owner[f] == r ? 1 : 0;
      if (compact && cnt[f] > 0) { base[f] = next; next += e->field_start[f + 1] - e->field_start[f]; }
    }
    int *d_lo = nullptr, *d_n = nullptr, *d_lin = nullptr;
    TRY_ALLOC(e->alloc(&d_lo, lo.size()));
    TRY_ALLOC(e->alloc(&d_n, cnt.size()));
    TRY_ALLOC(e->alloc(&d_lin, lin.size()));
    TRY_HIP(hipMemcpy(d_lo, lo.data(), lo.size() * sizeof(int), hipMemcpyHostToDevice));
    TRY_HIP(hipMemcpy(d_n, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
    TRY_HIP(hipMemcpy(d_lin, lin.data(), lin.size() * sizeof(int), hipMemcpyHostToDevice));
    m.own_lo = d_lo;
    m.own_n = d_n;
    m.lin_own = d_lin;
    e->lin_any = std::any_of(lin.begin(), lin.end(), [](int v) { return v != 0; });
    if (compact) {
      int *d_fs = nullptr;
      long long *d_base = nullptr;
      TRY_ALLOC(e->alloc(&d_fs, e->field_start.size()));
      TRY_ALLOC(e->alloc(&d_base, base.size()));
      TRY_HIP(hipMemcpy(d_fs, e->field_start.data(), e->field_start.size() * sizeof(int), hipMemcpyHostToDevice));
      TRY_HIP(hipMemcpy(d_base, base.data(), base.size() * sizeof(long long), hipMemcpyHostToDevice));
      m.field_start = d_fs;
      m.rec_base = d_base;
    }
  }
  TRY_ALLOC(e->alloc(&s.logit, R));
  TRY_ALLOC(e->alloc(&s.tg, R));
  TRY_ALLOC(e->alloc(&s.loss, R));
  TRY_ALLOC(e->alloc(&s.svx, R * static_cast<size_t>(m.type == FFM_MODEL_FM ? m.n_factors : 1)));
  {
    // stable LSD radix sort of (feature id, entry) pairs: key bits = those of the sentinel n_feats
    e->sort_bits = 1;
    while (e->sort_bits < 32 && (static_cast<uint64_t>(cfg->n_feats) >> e->sort_bits) != 0) e->sort_bits++;
    TRY_HIP(rocprim::radix_sort_pairs<GroupSortConfig>(nullptr, e->sort_tmp_bytes, s.key, s.skey,
                                      rocprim::counting_iterator<int>(0), s.occ, E, 0u, e->sort_bits,
                                      e->stream));
    e->sort_tmp_bytes = std::max(e->sort_tmp_bytes, sort_scratch_bytes(E));
    unsigned char *tmp = nullptr;
    TRY_ALLOC(e->alloc(&tmp, e->sort_tmp_bytes));
    e->d_sort_tmp[0] = tmp;
  }
  {
    TRY_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_super), sizeof(int) * ffm_engine::kSets, hipHostMallocMapped));
    std::memset(e->h_super, 0, sizeof(int) * ffm_engine::kSets);
    int *d_super = nullptr;
    TRY_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_super), e->h_super, 0));
    s.n_super = d_super;
    // The host skips the longest features' two extra launches when it reads 0 here -- which is only
    // safe if a device store to this mapping actually lands: prove it once, else the
    // launches are always made.
    hipLaunchKernelGGL(host_word_probe_kernel, dim3(1), dim3(64), 0, e->stream, d_super, ffm_engine::kSets);
    TRY_HIP(hipStreamSynchronize(e->stream));
    e->super_flag_ok = true;
    for (int si = 0; si < ffm_engine::kSets; si++) e->super_flag_ok = e->super_flag_ok && e->h_super[si] == 0x5eed + si;
    std::memset(e->h_super, 0, sizeof(int) * ffm_engine::kSets);
  }
  for (int si = 1; si < ffm_engine::kSets; si++) {
    Scratch &t = e->sc[si];
    t = s;  // shared: head/next/rowtab/logit/tg/loss/svx
    t.n_super = s.n_super + si;
    TRY_ALLOC(e->alloc(&t.key, E));
    TRY_ALLOC(e->alloc(&t.skey, E));
    TRY_ALLOC(e->alloc(&t.row_of, E));
    TRY_ALLOC(e->alloc(&t.occ, E));
    TRY_ALLOC(e->alloc(&t.occ2, E));
    TRY_ALLOC(e->alloc(&t.udesc, E));
    TRY_ALLOC(e->alloc(&t.small, E));
    TRY_ALLOC(e->alloc(&t.few, E));
    TRY_ALLOC(e->alloc(&t.sdesc, E));
    TRY_ALLOC(e->alloc(&t.big, E));
    TRY_ALLOC(e->alloc(&t.huge, E));
    TRY_ALLOC(e->alloc(&t.giant, E / kChainMin + 1));
    TRY_ALLOC(e->alloc(&t.gseg, E / kChainMin + 1));
    TRY_ALLOC(e->alloc(&t.grange, E / kFmRange + E / kFmGiantMin + 2));
    TRY_ALLOC(e->alloc(&t.counters, kNumCounters));
    TRY_ALLOC(e->alloc(&t.occpos, E));
    TRY_ALLOC(e->alloc(&t.uflag, E));
    if (masks) {
      TRY_ALLOC(e->alloc(&t.rowmask, 2 * R));
      TRY_ALLOC(e->alloc(&t.gmask, E));
      TRY_ALLOC(e->alloc(&t.cmask, E));
    }
    unsigned char *tmp = nullptr;
    TRY_ALLOC(e->alloc(&tmp, e->sort_tmp_bytes));
    e->d_sort_tmp[si] = tmp;
    TRY_HIP(hipMemsetAsync(t.counters, 0, kNumCounters * sizeof(int), e->stream));
  }
  TRY_HIP(hipStreamCreateWithFlags(&e->prep, hipStreamNonBlocking));
  // Which sort the grouping uses depends on how long a step of this engine is (estimated from the
  // largest block the engine was sized for).  Short steps wait for the look-ahead queue's chain of ~17
  // launches one to one (FM k = 64: 0.32 ms per step of which 0.23 ms are its two kernels) and take the
  // one-launch sort of kernels_sort.h (FM 0.323 -> 0.294 ms, FFM 8 x 16 at 4096 rows 0.139 -> 0.122); a
  // long FFM step is saturated by its row kernel, a sort that actually runs beside it slows it (r05:
  // 1.048 -> 1.094 ms), and the library sort's 1024-thread workgroups -- which only find room in the
  // gaps -- are the better neighbour.  FFM 39 x 4 is the crossover (no difference).
  // The upload kernel runs on the prep queue, ahead of the block's grouping.  (Round 4 moved it, for
  // long steps, behind the update's chain kernel on that kernel's stream; the update is one launch on
  // the main stream now and that stream is otherwise empty: 1.060 ms there, 1.048 ms on the prep
  // queue.)
  {
    const double per_row = e->max_rows > 0 ? static_cast<double>(e->max_nnz) / e->max_rows : 0.0;
    const double phase_us = static_cast<double>(e->max_nnz) * std::max(0.0, per_row - 1.0) * m.n_factors / 0.44e6;
    const bool ffm4 = m.type == FFM_MODEL_FFM && m.n_factors % 4 == 0;
    e->copy = e->prep;
    e->own_sort = !(ffm4 && phase_us / std::max(1, m.n_shards) >= 100.0);  // (a shard does 1/n_shards of the pairs)
    if (const char *sv = std::getenv("FFM_OWN_SORT")) e->own_sort = std::atoi(sv) != 0;
    // The fields' id ranges are known (cfg->field_start): every range is sorted by a workgroup of its
    // own, one launch, no grid barrier (kernels_sort.h: group_sort_ranges_kernel).  FFM_RANGE_SORT=0/1
    // overrides; 1 without field_start cuts the id space into n_fields equal ranges (a block whose ids
    // crowd into one of them is still sorted correctly, by one workgroup).
    {
      const char *sv = std::getenv("FFM_RANGE_SORT");
      const bool have = !e->field_start.empty();
      // (by default only where a workgroup per range is quick: a whole model -- a compact shard's dropped
      // columns make every block irregular, so every workgroup would scan all keys twice -- and at most
      // 16 k entries per range -- a rank's 65 536-row blocks are 256 tiles per pass for ONE workgroup:
      // an emulated 8-GPU rank went 1.72 -> 2.25 ms per step with it)
      const bool quick = cfg->n_shards == 1 && static_cast<int64_t>(e->max_nnz) <= 16384ll * std::max(1, cfg->n_fields);
      const bool want = sv ? std::atoi(sv) != 0 : (have && quick);
      if (want && (m.type == FFM_MODEL_FFM || sv)) {
        const int nr = have ? cfg->n_fields : (sv && std::atoi(sv) > 1 ? std::min(1024, std::atoi(sv)) : std::max(1, std::min(64, cfg->n_fields)));
        std::vector<int> st(static_cast<size_t>(nr) + 1);
        for (int f = 0; f <= nr; f++)
          st[f] = have ? e->field_start[f] : static_cast<int>(static_cast<int64_t>(cfg->n_feats) * f / nr);
        TRY_ALLOC(e->alloc(&e->d_sort_start, st.size()));
        TRY_HIP(hipMemcpy(e->d_sort_start, st.data(), st.size() * sizeof(int), hipMemcpyHostToDevice));
        e->n_sort_ranges = nr;
        e->range_sort = true;
        // (the regular-block short cut needs range f to be field f's: only with the caller's field_start)
        if (have && m.type == FFM_MODEL_FFM) m.sort_start = e->d_sort_start;
        // ... and its look-ahead starts when the ROW kernel of the block enqueued last has ended, beside
        // that block's update launches (half of their wave slots are free, and they are not bound by
        // bandwidth) instead of at the block's end, beside the next block's saturated row kernel -- where
        // this sort, unlike the library's (whose 1024-thread workgroups wait for a CU to drain and so only
        // ever fill gaps), costs the row kernel 5 %: C5 driver shape 0.970 -> 0.949 ms, C3 0.531 -> 0.521
        // (profiles/r06_experiments.md).  FFM_PREP_AFTER_ROW=0/1 overrides.
        // (long steps only: a 4096 x 8 block's update phase is 70 us, shorter than the chain)
        if (!std::getenv("FFM_PREP_AFTER_ROW")) e->prep_after_row = !e->own_sort;
        // ... and from 4 M slot-factors per block on the block's upload waits for that event as well (C5: the
        // driver's 20-step shape 0.887-0.896 -> 0.877-0.884 ms, 200 steps 0.870 -> 0.868; C3, 1.2 M: 0.472 ->
        // 0.478, not taken).  FFM_PULL_AFTER_ROW=0/1 overrides.
        e->pull_after_row = e->prep_after_row && static_cast<int64_t>(e->max_nnz) * m.n_factors >= (1ll << 22);
        if (const char *pv = std::getenv("FFM_PULL_AFTER_ROW")) e->pull_after_row = pv[0] == '1';
      }
    }
    // the one-launch sort meets at a grid barrier: never more workgroups than the device holds of it,
    // and only on the architecture its hand-over of data between workgroups was validated on
    {
      hipDeviceProp_t prop;
      TRY_HIP(hipGetDeviceProperties(&prop, cfg->device_id));
      int per_cu = 0;
      TRY_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, group_sort_kernel, kSortThreads, 0));
      e->sort_grid_cap = std::max(1, per_cu * prop.multiProcessorCount);
      if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && std::strncmp(prop.gcnArchName, "gfx942", 6) != 0) e->own_sort = false;
    }
  }
  for (int i = 0; i < ffm_engine::kSets; i++) {
    TRY_HIP(hipEventCreateWithFlags(&e->ev_grouped[i], hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&e->ev_set_free[i], hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&e->ev_row_done[i], hipEventDisableTiming));
  }
  TRY_ALLOC(e->alloc(&e->d_row_ptr, R + 1));
  TRY_ALLOC(e->alloc(&e->d_field, E));
  TRY_ALLOC(e->alloc(&e->d_feat, E));
  TRY_ALLOC(e->alloc(&e->d_val, E));
  TRY_ALLOC(e->alloc(&e->d_label, R));
  TRY_ALLOC(e->alloc(&e->d_out, R));
  TRY_ALLOC(e->alloc(&e->d_loss_sum, 2));
  TRY_ALLOC(e->alloc(&e->d_loss_part, kLossParts + 1));
  TRY_HIP(hipMemsetAsync(e->d_loss_part, 0, sizeof(double) * (kLossParts + 1), e->stream));
  e->stage_floats = 16 << 20;  // 64 MiB dense staging for get/set
  if (static_cast<int64_t>(e->logical_len) > e->stage_floats) e->stage_floats = e->logical_len;
  TRY_ALLOC(e->alloc(&e->d_stage, static_cast<size_t>(e->stage_floats)));

  TRY_HIP(hipMemsetAsync(m.bias3, 0, 4 * sizeof(float), e->stream));
  TRY_HIP(hipMemsetAsync(m.lin_n, 0, nf * sizeof(float), e->stream));
  TRY_HIP(hipMemsetAsync(m.lin_z, 0, nf * sizeof(float), e->stream));
  TRY_HIP(hipMemsetAsync(m.lin_w, 0, nf * sizeof(float), e->stream));
  if (n_lat) TRY_HIP(hipMemsetAsync(m.lat, 0, n_lat * sizeof(float), e->stream));
  TRY_HIP(hipMemsetAsync(s.counters, 0, kNumCounters * sizeof(int), e->stream));
  {
    // the five-instruction square root rests on this device's v_sqrt_f32 / v_rsq_f32: compare it
    // with sqrtf over its whole range (1.6e9 inputs, a few milliseconds) before any kernel uses it
    // (once per device and process: the answer is a property of the hardware)
    static std::atomic<int> verdict[64];  // 0 unknown, 1 exact, -1 differs
    const int dev = cfg->device_id & 63;
    int bad = verdict[dev].load() < 0 ? 1 : 0;
    if (verdict[dev].load() == 0) {
      hipLaunchKernelGGL(verify_sqrt_fast_kernel, dim3(16384), dim3(256), 0, e->stream, s.counters);
      TRY_HIP(hipMemcpyAsync(&bad, s.counters, sizeof(int), hipMemcpyDeviceToHost, e->stream));
      TRY_HIP(hipStreamSynchronize(e->stream));
      TRY_HIP(hipMemsetAsync(s.counters, 0, kNumCounters * sizeof(int), e->stream));
      verdict[dev].store(bad ? -1 : 1);
    }
    if (bad) {
      ffm_engine_destroy(e);
      return fail(FFM_E_UNSUPPORTED, "this device's v_sqrt_f32 / v_rsq_f32 do not give the correctly rounded "
                                     "square root in the short sequence the kernels use (built and proven for gfx950)");
    }
  }
  if (cfg->w_alpha >= 0x1p-30f && cfg->w_alpha <= 0x1p30f) {
    // prove the short x/alpha sequence exact for this alpha before any kernel may use it
    int bad = 0;
    Hyper probe = m.h;
    probe.fast_div = 1;
    hipLaunchKernelGGL(verify_div_alpha_kernel, dim3(1u << 17), dim3(256), 0, e->stream, probe, s.counters);
    TRY_HIP(hipMemcpyAsync(&bad, s.counters, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    TRY_HIP(hipStreamSynchronize(e->stream));
    TRY_HIP(hipMemsetAsync(s.counters, 0, kNumCounters * sizeof(int), e->stream));
    m.h.fast_div = bad ? 0 : 1;
    m.h.fast_w = m.h.fast_div && cfg->w_beta >= 0x1p-40f && cfg->w_beta <= 0x1p40f ? 1 : 0;
  }
  {
    // the row kernels stage one row in dynamic LDS: opt in to what the longest admissible row needs
    const size_t lds = row_lds_bytes(e->max_row_nnz, m.n_fields, kTermsCap);
    if (lds > 150 * 1024) {
      ffm_engine_destroy(e);
      return fail(FFM_E_UNSUPPORTED, "max_row_nnz too large for the 160 KB of LDS per workgroup");
    }
    const size_t park_budget = std::getenv("FFM_ROW_PARK_BUDGET") ? static_cast<size_t>(std::atoi(std::getenv("FFM_ROW_PARK_BUDGET"))) : 0;
    if (lds > 32 * 1024 || park_budget > 32 * 1024) {
      const int bytes = static_cast<int>(std::max(lds, park_budget));
      TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_row_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_row_kernel<true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_row_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_row_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_row_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fm_row_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fm_row_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    // the FFM update launch: the tile pipelines' per-wave LDS regions of its (up to 16-wave) workgroups
    TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_update_all_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(tile_lds_bytes(1))));
    TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_update_all_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(tile_lds_bytes(2))));
    TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_update_all_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(tile_lds_bytes(4))));
    TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_update_all_kernel<1, UPD_ALL & ~UPD_FEW>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(tile_lds_bytes(1))));
    TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_update_all_kernel<2, UPD_ALL & ~UPD_FEW>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(tile_lds_bytes(2))));
    TRY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffm_update_all_kernel<4, UPD_ALL & ~UPD_FEW>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(tile_lds_bytes(4))));
  }
  if (!(cfg->flags & FFM_FLAG_SKIP_INIT))
    hipLaunchKernelGGL(init_weights_kernel, dim3(2048), dim3(256), 0, e->stream, m, e->logical_len,
                       cfg->init_mean, cfg->init_stddev, cfg->seed);
  TRY_HIP(hipGetLastError());
  TRY_HIP(hipStreamSynchronize(e->stream));
#undef TRY_ALLOC
#undef TRY_HIP
  *out = e;
  return FFM_OK;
}

// What init_weights_kernel stored, computed on the host (csrc/init_rng.h: same bits).
static int eval_launch_pending(ffm_engine *e);  // (engine_step.h)

int ffm_engine_init_weights_host(uint64_t seed, float init_mean, float init_stddev, int32_t latent,
                                 int64_t first, int64_t count, float *out) {
  if (first < 0 || count < 0 || (count > 0 && !out)) return fail(FFM_E_INVALID, "bad range or null output");
  for (int64_t j = 0; j < count; j++)
    out[j] = ftrl_rng::init_weight(seed, latent ? 1 : 0, static_cast<uint64_t>(first + j), init_mean, init_stddev);
  return FFM_OK;
}

int ffm_engine_fill_state(ffm_engine *e, uint64_t seed, float n_lo, float n_hi, float z_stddev) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (!(n_lo >= 0.0f) || !(n_hi >= n_lo)) return fail(FFM_E_INVALID, "need 0 <= n_lo <= n_hi");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  if (int rc_e = eval_launch_pending(e)) return rc_e;
  hipLaunchKernelGGL(fill_state_kernel, dim3(2048), dim3(256), 0, e->stream, e->m, e->logical_len, seed, n_lo, n_hi, z_stddev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  return FFM_OK;
}

int ffm_engine_eval_sigmoid(ffm_engine *e, int32_t n, const float *x, float *y) {
  if (!e || n < 0 || (n > 0 && (!x || !y))) return fail(FFM_E_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  const int chunk = static_cast<int>(std::min<int64_t>(e->stage_floats / 2, 1 << 22));
  for (int off = 0; off < n; off += chunk) {
    const int c = std::min(chunk, n - off);
    HIP_TRY(hipMemcpyAsync(e->d_stage, x + off, sizeof(float) * c, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(sigmoid_eval_kernel, dim3(cdiv(c, 256)), dim3(256), 0, e->stream, c, e->d_stage, e->d_stage + chunk);
    HIP_TRY(hipMemcpyAsync(y + off, e->d_stage + chunk, sizeof(float) * c, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return FFM_OK;
}

// Waits for the engine's stream, then reports (and clears) what the kernels flagged since the
// last report: the device entry points are asynchronous, so this is where their callers learn
// that a block could not be trained.
static int check_device_errors(ffm_engine *e) {
  int flags = 0;
  if (int rc_e = eval_launch_pending(e)) return rc_e;
  if (int rc_w = e->drain()) return rc_w;
  // the uploads of blocks that are staged but not trained yet run on the prep stream: the
  // zero-copy contract (include/ffm_engine.h) lets the caller reuse its page-locked arrays once
  // this returns, so they must have been pulled too
  for (int i = 0; i < e->n_staged; i++) HIP_TRY(hipEventSynchronize(e->slots[e->staged[i]].ev_copied));
  HIP_TRY(hipMemcpyAsync(&flags, e->d_err, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (flags) HIP_TRY(hipMemsetAsync(e->d_err, 0, sizeof(int), e->stream));
  if (flags & ERR_ROW_TOO_LONG)
    return fail(FFM_E_CAPACITY, "a row has more entries than max_row_nnz: its block was not trained "
                                "(outputs of that block are NaN)");
  if (flags & ERR_FIELD_MAP)
    return fail(FFM_E_INVALID, "an entry's feature id lies outside its field's id range (field_start): "
                               "its block was not trained");
  if (flags & ERR_SORT_BARRIER)
    return fail(FFM_E_DEVICE, "the grouping's one-launch sort could not get its workgroups on the device "
                              "together (grid barrier timed out): its block was not trained; FFM_OWN_SORT=0 "
                              "selects the library sort");
  if (flags) return fail(FFM_E_DEVICE, "device error flags " + std::to_string(flags));
  return FFM_OK;
}

int ffm_engine_sync(ffm_engine *e) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  return check_device_errors(e);
}

int ffm_engine_check_errors(ffm_engine *e) { return ffm_engine_sync(e); }

void *ffm_engine_stream(ffm_engine *e) { return e ? static_cast<void *>(e->stream) : nullptr; }

// Who owns what under field-pair sharding (host arithmetic only, no device needed).
int32_t ffm_engine_default_batch_ramp(float w_alpha) {
  if (!(w_alpha > 1e-3f)) return 32;
  const double r = 32.0 * std::pow(8.0, std::log10(static_cast<double>(w_alpha) / 1e-3));  // x 8 per decade
  return static_cast<int32_t>(std::min(r, 1048576.0) + 0.5);
}

int ffm_engine_shard_plan(int32_t n_fields, int32_t n_shards, int32_t field_map, int32_t *pair_owner,
                          int32_t *lin_owner, int32_t *bias_owner) {
  if (n_fields <= 0 || n_shards <= 0) return fail(FFM_E_INVALID, "n_fields and n_shards must be positive");
  const ShardPlan p = make_shard_plan(n_fields, n_shards, field_map != 0);
  // self-check: every unordered pair has exactly one owner, symmetrically
  for (int fa = 0; fa < n_fields; fa++)
    for (int fb = 0; fb < n_fields; fb++) {
      int owner = -1, count = 0;
      for (int r = 0; r < n_shards; r++)
        if (p.owns(r, fa, fb)) { owner = r; count++; }
      if (count != 1 || !p.owns(owner, fb, fa)) return fail(FFM_E_UNSUPPORTED, "no field-pair partition for this shape");
      if (pair_owner) pair_owner[static_cast<size_t>(fa) * n_fields + fb] = owner;
    }
  if (lin_owner) for (int f = 0; f < n_fields; f++) lin_owner[f] = p.lin_owner[f];
  if (bias_owner) *bias_owner = p.bias_owner;
  return FFM_OK;
}

#include "engine_state.h"
#include "engine_step.h"
#include "engine_stage.h"
#include "engine_profile.h"
}  // extern "C"

#include "engine_group.h"
