// engine_group.h -- several GPUs in one process: a group of field-pair shard engines and the one
// collective of the path (include/ffm_engine.h: ffm_group_*).  Included at the end of engine.hip.
//
// The reference has no counterpart (single process, shared-memory threads: SURVEY.md 5); the caller
// this serves is FtrlOffline::one_epoch / FtrlOnline::run_task (src/task/ftrl_offline.cpp:63-103,
// src/task/ftrl_online.cpp:70-80) when the model is sharded (BASELINE.json config 5).
//
// Per block: every engine stages the block (its GPU pulls the rows from the caller's page-locked
// arrays over its own PCIe link), computes the partial logits of the field pairs it owns
// (ffm_engine_train_forward_staged), ONE all-reduce sums the n_rows floats, every engine updates
// its slots (ffm_engine_train_update_device).  The all-reduce is RCCL's ncclAllReduce, one call per
// device between ncclGroupStart / ncclGroupEnd, enqueued on each engine's OWN stream -- so it is
// ordered behind that engine's forward and ahead of its update without any event, and the host
// never waits.  librccl.so is bound at run time (dlopen): the host side needs no ROCm headers.
// 32 KB per 8192-row block: the collective is latency-bound (a few tens of microseconds over
// xGMI), not link-bound.
#pragma once

#include <dlfcn.h>

namespace {

// The six RCCL entry points used, by their C signatures (rccl.h); ncclFloat32 = 7, ncclSum = 0.
struct Rccl {
  using comm_t = void *;
  int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
  Rccl() {
    // a copy the process already holds (a host that also uses torch.distributed has one) is reused:
    // two RCCL images in one process corrupt each other's teardown
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    if (!h)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return;
#define BIND(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(h, sym))
    BIND(CommInitAll, "ncclCommInitAll");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(AllReduce, "ncclAllReduce");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce;
  }
};
Rccl &rccl() {
  static Rccl r;
  return r;
}
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

// Engines that share a device: out[r][i] = sum_s part[s][i] for every r (shard order, float32).
struct SumJob { float *part[16]; int n; };
__global__ void group_sum_kernel(SumJob job, int n_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  float acc = job.part[0][i];
  for (int s = 1; s < job.n; s++) acc += job.part[s][i];
  for (int s = 0; s < job.n; s++) job.part[s][i] = acc;
}

}  // namespace

struct ffm_group {
  std::vector<ffm_engine *> eng;
  std::vector<int> dev;
  std::vector<float *> logit;  // [n] n_rows floats on each engine's device
  std::vector<Rccl::comm_t> comm;
  bool use_rccl = false;
  hipEvent_t ev_part[16] = {}, ev_sum = nullptr;  // device-local sum: partials done / sum done
  int max_rows = 0;
  int64_t handed = 0;  // blocks handed over so far
  int n_staged = 0;    // staged on every engine, not trained yet
  bool poisoned = false;  // a block was staged on SOME engines only: the shards no longer hold the same queue
};

static int group_allreduce(ffm_group *g, int n_rows) {
  const int n = static_cast<int>(g->eng.size());
  if (n == 1 && !g->use_rccl) return FFM_OK;
  if (g->use_rccl) {
    int rc = rccl().GroupStart();
    hipError_t dev_err = hipSuccess;
    for (int r = 0; r < n && rc == 0 && dev_err == hipSuccess; r++) {
      // (no early return between GroupStart and GroupEnd: an open RCCL group would swallow every
      // later collective of the process)
      dev_err = hipSetDevice(g->dev[r]);
      if (dev_err != hipSuccess) break;
      rc = rccl().AllReduce(g->logit[r], g->logit[r], static_cast<size_t>(n_rows), kNcclFloat32, kNcclSum,
                            g->comm[r], g->eng[r]->stream);
    }
    const int rc2 = rccl().GroupEnd();
    if (dev_err != hipSuccess)
      return fail(FFM_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(dev_err));
    if (rc || rc2)
      return fail(FFM_E_DEVICE, std::string("ncclAllReduce: ") +
                                    (rccl().GetErrorString ? rccl().GetErrorString(rc ? rc : rc2) : "error"));
    return FFM_OK;
  }
  // engines on one device: engine 0's stream sums once every partial is there; the others wait for it
  HIP_TRY(hipSetDevice(g->dev[0]));
  SumJob job{};
  job.n = n;
  for (int r = 0; r < n; r++) {
    job.part[r] = g->logit[r];
    if (r > 0) {
      HIP_TRY(hipEventRecord(g->ev_part[r], g->eng[r]->stream));
      HIP_TRY(hipStreamWaitEvent(g->eng[0]->stream, g->ev_part[r], 0));
    }
  }
  hipLaunchKernelGGL(group_sum_kernel, dim3(cdiv(n_rows, 256)), dim3(256), 0, g->eng[0]->stream, job, n_rows);
  HIP_TRY(hipEventRecord(g->ev_sum, g->eng[0]->stream));
  for (int r = 1; r < n; r++) HIP_TRY(hipStreamWaitEvent(g->eng[r]->stream, g->ev_sum, 0));
  return FFM_OK;
}

// forward -> all-reduce -> update of the oldest block staged on every engine; its loss goes into
// engine 0's running sum (every engine computes the same tmp_grad; one reports the loss)
static int group_train_one_staged(ffm_group *g, float *logit_host) {
  const int n = static_cast<int>(g->eng.size());
  const int n_rows = g->eng[0]->slots[g->eng[0]->staged[0]].n_rows;
  int rc;
  for (int r = 0; r < n; r++)
    if ((rc = ffm_engine_train_forward_staged(g->eng[r], g->logit[r]))) return rc;
  if ((rc = group_allreduce(g, n_rows))) return rc;
  for (int r = 0; r < n; r++)
    if ((rc = ffm_engine_train_update_device(g->eng[r], g->logit[r], nullptr, r == 0 ? g->eng[0]->d_loss_sum : nullptr)))
      return rc;
  HIP_TRY(hipSetDevice(g->dev[0]));
  hipLaunchKernelGGL(loss_accumulate_kernel, dim3(1), dim3(1), 0, g->eng[0]->stream, g->eng[0]->d_loss_acc,
                     g->eng[0]->d_loss_sum);
  if (logit_host && n_rows > 0)
    HIP_TRY(hipMemcpyAsync(logit_host, g->logit[0], sizeof(float) * n_rows, hipMemcpyDeviceToHost, g->eng[0]->stream));
  g->n_staged--;
  return FFM_OK;
}

extern "C" {

int ffm_group_create(const ffm_engine_config *cfg, int32_t n, const int32_t *device_ids, ffm_group **out) {
  if (!cfg || !out || n < 1 || n > 16 || !device_ids) return fail(FFM_E_INVALID, "bad group arguments (1 .. 16 engines)");
  *out = nullptr;
  auto *g = new (std::nothrow) ffm_group();
  if (!g) return fail(FFM_E_NOMEM, "out of host memory");
  bool distinct = true;
  for (int a = 0; a < n; a++)
    for (int b = a + 1; b < n; b++) distinct = distinct && device_ids[a] != device_ids[b];
  // one engine alone needs no exchange; FFM_GROUP_RCCL=1 runs the collective anyway (a sum over one
  // rank: lets a one-GPU box exercise the librccl binding and the stream ordering)
  const char *force = std::getenv("FFM_GROUP_RCCL");
  g->use_rccl = distinct && (n > 1 || (force && force[0] == '1'));
  int rc = FFM_OK;
  for (int r = 0; r < n && rc == FFM_OK; r++) {
    ffm_engine_config c = *cfg;
    c.n_shards = n;
    c.shard_rank = r;
    c.device_id = device_ids[r];
    c.stream = nullptr;  // every engine on a stream of its own
    ffm_engine *e = nullptr;
    rc = ffm_engine_create(&c, &e);
    if (rc) break;
    g->eng.push_back(e);
    g->dev.push_back(device_ids[r]);
    float *lg = nullptr;
    if ((rc = e->alloc(&lg, static_cast<size_t>(std::max(1, c.max_batch_rows))))) break;
    g->logit.push_back(lg);
  }
  if (rc == FFM_OK && g->use_rccl) {
    if (!rccl().ok) {
      rc = fail(FFM_E_UNSUPPORTED, "librccl.so not found: a group on distinct devices needs it");
    } else {
      g->comm.assign(static_cast<size_t>(n), nullptr);
      const int nrc = rccl().CommInitAll(g->comm.data(), n, g->dev.data());
      if (nrc) rc = fail(FFM_E_DEVICE, std::string("ncclCommInitAll: ") + (rccl().GetErrorString ? rccl().GetErrorString(nrc) : "error"));
    }
  }
  if (rc == FFM_OK && !g->use_rccl && n > 1) {
    if (hipSetDevice(g->dev[0]) != hipSuccess) rc = fail(FFM_E_DEVICE, "hipSetDevice");
    for (int r = 0; r < n && rc == FFM_OK; r++)
      if (hipEventCreateWithFlags(&g->ev_part[r], hipEventDisableTiming) != hipSuccess) rc = fail(FFM_E_DEVICE, "hipEventCreate");
    if (rc == FFM_OK && hipEventCreateWithFlags(&g->ev_sum, hipEventDisableTiming) != hipSuccess) rc = fail(FFM_E_DEVICE, "hipEventCreate");
  }
  if (rc != FFM_OK) {
    const std::string keep = g_last_error;
    ffm_group_destroy(g);
    g_last_error = keep;
    return rc;
  }
  g->max_rows = cfg->max_batch_rows;
  *out = g;
  return FFM_OK;
}

void ffm_group_destroy(ffm_group *g) {
  if (!g) return;
  for (size_t r = 0; r < g->eng.size(); r++) {
    (void)hipSetDevice(g->dev[r]);
    (void)hipStreamSynchronize(g->eng[r]->stream);
  }
  for (size_t r = 0; r < g->comm.size(); r++)
    if (g->comm[r]) {
      (void)hipSetDevice(g->dev[r]);  // a communicator is torn down on its own device
      (void)rccl().CommDestroy(g->comm[r]);
    }
  for (auto &ev : g->ev_part) if (ev) (void)hipEventDestroy(ev);
  if (g->ev_sum) (void)hipEventDestroy(g->ev_sum);
  for (auto *e : g->eng) ffm_engine_destroy(e);
  delete g;
}

int32_t ffm_group_size(const ffm_group *g) { return g ? static_cast<int32_t>(g->eng.size()) : 0; }
ffm_engine *ffm_group_engine(ffm_group *g, int32_t rank) {
  return (g && rank >= 0 && rank < static_cast<int32_t>(g->eng.size())) ? g->eng[rank] : nullptr;
}
const char *ffm_group_collective(const ffm_group *g) {
  return !g ? "" : g->use_rccl ? "rccl" : g->eng.size() > 1 ? "device-local sum" : "none";
}

int ffm_group_train_batch_async(ffm_group *g, int32_t n_rows, const int32_t *row_ptr,
                                const int32_t *field, const int32_t *feat, const float *val,
                                const int32_t *label, int32_t zero_copy) {
  if (!g) return fail(FFM_E_INVALID, "null group");
  if (g->poisoned) return fail(FFM_E_INVALID, "an earlier block reached only some of the group's engines: destroy the group");
  int rc;
  for (size_t r = 0; r < g->eng.size(); r++)
    if ((rc = ffm_engine_stage_batch(g->eng[r], n_rows, row_ptr, field, feat, val, label, zero_copy))) {
      // (engine 0 refuses what any engine would refuse -- the checks are the block's -- so r > 0 means
      // a device error; the engines before r hold one block more than the rest)
      if (r > 0) g->poisoned = true;
      return rc;
    }
  g->n_staged++;
  g->handed++;
  // copying path: two blocks in flight (the one staged just now keeps uploading and grouping while
  // the previous one trains); page-locked path: three, as on one engine
  const int keep = zero_copy ? 2 : 1;
  while (g->n_staged > keep)
    if ((rc = group_train_one_staged(g, nullptr))) return rc;
  return FFM_OK;
}

int ffm_group_train_flush(ffm_group *g, double *loss_sum_out) {
  if (!g) return fail(FFM_E_INVALID, "null group");
  int rc;
  while (g->n_staged > 0)
    if ((rc = group_train_one_staged(g, nullptr))) return rc;
  double total = 0.0;
  for (size_t r = 0; r < g->eng.size(); r++) {
    double part = 0.0;
    if ((rc = ffm_engine_train_flush(g->eng[r], &part))) return rc;  // waits; engine 0 holds the sum
    if (r == 0) total = part;
  }
  if (loss_sum_out) *loss_sum_out = total;
  return FFM_OK;
}

int ffm_group_train_batch(ffm_group *g, int32_t n_rows, const int32_t *row_ptr, const int32_t *field,
                          const int32_t *feat, const float *val, const int32_t *label,
                          float *logit_out, double *loss_sum_out) {
  if (!g) return fail(FFM_E_INVALID, "null group");
  if (g->n_staged > 0) return fail(FFM_E_INVALID, "pipelined blocks are still waiting: flush first");
  if (g->poisoned) return fail(FFM_E_INVALID, "an earlier block reached only some of the group's engines: destroy the group");
  int rc;
  for (size_t r = 0; r < g->eng.size(); r++)
    if ((rc = ffm_engine_stage_batch(g->eng[r], n_rows, row_ptr, field, feat, val, label, 0))) {
      if (r > 0) g->poisoned = true;
      return rc;
    }
  g->n_staged++;
  g->handed++;
  if ((rc = group_train_one_staged(g, logit_out))) return rc;
  return ffm_group_train_flush(g, loss_sum_out);
}

int64_t ffm_group_blocks_pulled(ffm_group *g) {
  if (!g) return 0;
  int64_t lo = INT64_MAX;
  for (auto *e : g->eng) lo = std::min<int64_t>(lo, ffm_engine_blocks_pulled(e));
  return lo == INT64_MAX ? 0 : lo;
}

int ffm_group_predict_batch(ffm_group *g, int32_t n_rows, const int32_t *row_ptr, const int32_t *field,
                            const int32_t *feat, const float *val, const int32_t *label,
                            int32_t output_prob, float *out, double *loss_sum_out) {
  if (!g) return fail(FFM_E_INVALID, "null group");
  if (g->n_staged > 0) return fail(FFM_E_INVALID, "pipelined blocks are still waiting: flush first");
  const int n = static_cast<int>(g->eng.size());
  int rc;
  int32_t nnz = 0;
  for (int r = 0; r < n; r++) {
    ffm_engine *e = g->eng[r];
    if ((rc = stage_block(e, n_rows, row_ptr, field, feat, val, label, &nnz))) return rc;
    if ((rc = ffm_engine_predict_batch_device(e, n_rows, nnz, e->d_row_ptr, field ? e->d_field : nullptr, e->d_feat,
                                              e->d_val, nullptr, 0, g->logit[r], nullptr)))
      return rc;
  }
  if ((rc = group_allreduce(g, n_rows))) return rc;
  ffm_engine *e0 = g->eng[0];
  if ((rc = ffm_engine_predict_finish_device(e0, n_rows, g->logit[0], label ? e0->d_label : nullptr, output_prob,
                                             e0->d_out, label ? e0->d_loss_sum : nullptr)))
    return rc;
  HIP_TRY(hipSetDevice(g->dev[0]));
  if (out && n_rows > 0) HIP_TRY(hipMemcpyAsync(out, e0->d_out, sizeof(float) * n_rows, hipMemcpyDeviceToHost, e0->stream));
  if (loss_sum_out) {
    if (label) HIP_TRY(hipMemcpyAsync(loss_sum_out, e0->d_loss_sum, sizeof(double), hipMemcpyDeviceToHost, e0->stream));
    else *loss_sum_out = 0.0;
  }
  for (int r = 0; r < n; r++)
    if ((rc = ffm_engine_sync(g->eng[r]))) return rc;
  return FFM_OK;
}

}  // extern "C"
