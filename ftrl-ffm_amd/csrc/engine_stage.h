// engine_stage.h -- rows streaming from host memory inside the training loop: staging slots, the
// upload kernel, stage_batch / train_staged / train_batch_async[_pinned] / predict_batch_async.
// Part of engine.hip's translation unit (included inside its extern "C" block).

// ---- pipelined host-buffer training ---------------------------------------------------------

// Upload of one staged block by a kernel: the five CSR arrays are read straight out of page-locked
// (device-mapped) host memory, 16 bytes per lane, and written to the staging slot's device arrays.
// A hipMemcpyAsync here makes the SUBMITTING THREAD wait until the stream's earlier kernels have
// finished (measured: mean 0.38 ms, up to 16 ms per call) -- the host then cannot run ahead of the
// GPU and every other step starts ~230 us late; a kernel launch never blocks.
// Every wave starts with a system-scope acquire (it drops the non-coherent lines of its L2): a
// caller that refills a block buffer it has used before (the trainers' ring) must not be served
// lines of the previous block that are still on-die, should the runtime map its page-locked memory
// cacheable.  (System-scope LOADS instead -- 8 bytes per lane -- halved the upload rate.)
struct PullJob {
  const char *src[5]; char *dst[5]; unsigned bytes[5];
  long long ordinal; long long *pulled; unsigned *ticket;  // completion word (host memory), see h_pulled
  // rows handed over without a field array (one entry per field in field order): the slot's field
  // array is written here -- entry j of the block is field j mod n_fields -- instead of crossing PCIe
  int *gen_field; unsigned gen_n; int gen_fields;
};
__global__ __launch_bounds__(256) void pull_block_kernel(PullJob job) {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
#pragma unroll
  for (int a = 0; a < 5; a++) {
    const unsigned n16 = job.bytes[a] >> 4;
    const int4 *s = reinterpret_cast<const int4 *>(job.src[a]);
    int4 *d = reinterpret_cast<int4 *>(job.dst[a]);
    // (one load per lane in flight: two, four and eight measured the same -- 42 GB/s is what a kernel reads over PCIe here)
    for (unsigned i = tid; i < n16; i += stride) d[i] = s[i];
    const unsigned tail = job.bytes[a] & 15u;  // sizes are multiples of 4
    if (tid < (tail >> 2))
      reinterpret_cast<int *>(job.dst[a])[(n16 << 2) + tid] = reinterpret_cast<const int *>(job.src[a])[(n16 << 2) + tid];
  }
  if (job.gen_n) {
    const unsigned n4 = job.gen_n >> 2, F = static_cast<unsigned>(job.gen_fields);
    for (unsigned i = tid; i < n4; i += stride) {
      const unsigned f0 = (4u * i) % F, f1 = f0 + 1u < F ? f0 + 1u : 0u, f2 = f1 + 1u < F ? f1 + 1u : 0u;
      reinterpret_cast<int4 *>(job.gen_field)[i] = make_int4(static_cast<int>(f0), static_cast<int>(f1), static_cast<int>(f2),
                                                             static_cast<int>(f2 + 1u < F ? f2 + 1u : 0u));
    }
    if (tid < (job.gen_n & 3u)) job.gen_field[(n4 << 2) + tid] = static_cast<int>(((n4 << 2) + tid) % F);
  }
  // the workgroup that finishes last publishes the block's number to the host
  __syncthreads();  // (every load of this workgroup has returned: its stores were issued after them)
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(job.ticket, 1u) == gridDim.x - 1) {
      *job.ticket = 0u;
      __hip_atomic_store(job.pulled, job.ordinal, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static int slots_init(ffm_engine *e) {
  if (e->slots_ready) return FFM_OK;
  const size_t R = static_cast<size_t>(e->max_rows), E = static_cast<size_t>(e->max_nnz);
  const size_t bytes = 4 * (R + 1) + 4 * E * 3 + 4 * R + 5 * 16;  // each of the 5 arrays is padded to 16 B
  for (auto &sl : e->slots) {
    // (portable: page-locked for EVERY device of the process -- an ffm_group's engines sit on
    // different GPUs and all read host blocks)
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sl.pinned), bytes, hipHostMallocPortable | hipHostMallocMapped));
    int rc;
    if ((rc = e->alloc(&sl.row_ptr, R + 1)) || (rc = e->alloc(&sl.field, E)) || (rc = e->alloc(&sl.feat, E)) ||
        (rc = e->alloc(&sl.val, E)) || (rc = e->alloc(&sl.label, R)))
      return rc;
    HIP_TRY(hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&sl.ev_trained, hipEventDisableTiming));
  }
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_pulled), 64, hipHostMallocPortable | hipHostMallocMapped));
  *e->h_pulled = 0;
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&e->d_pulled), e->h_pulled, 0));
  if (int rc_t = e->alloc(&e->d_pull_ticket, 1)) return rc_t;
  HIP_TRY(hipMemsetAsync(e->d_pull_ticket, 0, sizeof(unsigned), e->copy));
  int rc = e->alloc(&e->d_loss_acc, 1);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(e->d_loss_acc, 0, sizeof(double), e->stream));
  e->slots_ready = true;
  return FFM_OK;
}

int ffm_engine_pin_host(void *p, size_t bytes) {
  if (!p || !bytes) return fail(FFM_E_INVALID, "null range");
  // portable: one registration serves every GPU of the process (an ffm_group stages the same host
  // block on all of its engines; without the flag only the current device may read it)
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
  return FFM_OK;
}
int ffm_engine_unpin_host(void *p) {
  if (!p) return fail(FFM_E_INVALID, "null pointer");
  HIP_TRY(hipHostUnregister(p));
  return FFM_OK;
}

// The next staging slot for a host block: waits until the slot's own pinned image is free, copies
// the caller's arrays into it (unless zero_copy) and describes the upload (pull_block_kernel's
// argument; the block's ordinal is n_staged_total + 1).  Bookkeeping of the slot is the caller's.
static int claim_slot(ffm_engine *e, int32_t n_rows, int32_t nnz, const int32_t *row_ptr,
                      const int32_t *field, const int32_t *feat, const float *val, const int32_t *label,
                      int32_t zero_copy, int *slot_out, bool *was_used, PullJob *job_out) {
  int rc;
  if ((rc = slots_init(e))) return rc;
  ffm_engine::Slot &sl = e->slots[e->slot_next];
  *slot_out = e->slot_next;
  *was_used = sl.used;
  if (sl.used && !sl.zero_copy) {
    ScopedTimer tm("stage:slot_wait");
    // the slot's own pinned image must have been pulled before it is overwritten.  (Not so for a
    // zero_copy block, whose image is the caller's: blocking the submitting thread here costs
    // ~0.2 ms per step; such callers ask ffm_engine_blocks_pulled before reusing their memory.)
    if ((rc = e->wait_issued(sl.seq))) return rc;
    HIP_TRY(hipEventSynchronize(sl.ev_copied));
  }
  // host arrays -> pinned image (the caller may reuse its buffers on return) -> device, prep stream
  const size_t R1 = static_cast<size_t>(n_rows) + 1, E = static_cast<size_t>(nnz);
  char *p = sl.pinned;
  PullJob &job = *job_out;
  int n_job = 0;
  // page-locked source of each array: the caller's own (zero_copy: untouched until the block has
  // trained) or its image in the slot's pinned buffer; the device then pulls it (pull_block_kernel)
  auto put = [&](const void *src, size_t bytes, void *dst) -> hipError_t {
    if (!bytes || !src) return hipSuccess;
    const void *host = src;
    if (!zero_copy) {
      std::memcpy(p, src, bytes);
      host = p;
      p += (bytes + 15) & ~static_cast<size_t>(15);
    }
    void *mapped = nullptr;
    hipError_t err = hipHostGetDevicePointer(&mapped, const_cast<void *>(host), 0);
    if (err != hipSuccess) return err;
    if ((reinterpret_cast<uintptr_t>(mapped) & 15u) != 0) return hipErrorInvalidValue;  // 16-byte aligned arrays only
    job.src[n_job] = static_cast<const char *>(mapped);
    job.dst[n_job] = static_cast<char *>(dst);
    job.bytes[n_job] = static_cast<unsigned>(bytes);
    n_job++;
    return hipSuccess;
  };
  ScopedTimer tm("stage:copies");
  HIP_TRY(put(row_ptr, 4 * R1, sl.row_ptr));
  if (field) HIP_TRY(put(field, 4 * E, sl.field));
  else if (e->m.type == FFM_MODEL_FFM && E > 0) {  // (validated: one entry per field in field order)
    job.gen_field = sl.field;
    job.gen_n = static_cast<unsigned>(E);
    job.gen_fields = e->m.n_fields;
  }
  HIP_TRY(put(feat, 4 * E, sl.feat));
  HIP_TRY(put(val, 4 * E, sl.val));
  HIP_TRY(put(label, 4 * static_cast<size_t>(n_rows), sl.label));
  job.ordinal = e->n_staged_total + 1;
  job.pulled = e->d_pulled;
  job.ticket = e->d_pull_ticket;
  return FFM_OK;
}

// Stage one block of host rows: (pinned image ->) HBM -> grouping, all on the prep stream.
int ffm_engine_stage_batch(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                           const int32_t *field, const int32_t *feat, const float *val,
                           const int32_t *label, int32_t zero_copy) {
  int32_t nnz = 0;
  int longest = 1;
  // (LR / FM rows have no fields -- libsvm: src/data/parser.cpp:20 gives every entry field 0 -- so a
  // field array the caller passes along is not uploaded: a third of the block's bytes over PCIe)
  if (e && e->m.type != FFM_MODEL_FFM) field = nullptr;
  int rc = validate_host_block(e, n_rows, row_ptr, field, feat, val, &nnz, &longest, true);
  if (rc) return rc;
  const bool has_field = field != nullptr || (e->m.type == FFM_MODEL_FFM && nnz > 0);  // (uploaded or written by the upload kernel)
  if (n_rows > 0 && !label) return fail(FFM_E_INVALID, "training needs labels");
  if ((rc = eval_launch_pending(e))) return rc;
  if (e->n_staged >= ffm_engine::kSlots - 1) return fail(FFM_E_CAPACITY, "three staged blocks are already waiting");
  if (e->has_pending) return fail(FFM_E_INVALID, "stage between train_forward and train_update");
  // everything ffm_engine_prepare_device can refuse is refused HERE, before the upload kernel is
  // launched: that kernel publishes the block's ordinal to ffm_engine_blocks_pulled, and a block
  // that then failed to stage would leave the count one ahead for good
  if (e->n_prepared >= ffm_engine::kSets - 1) return fail(FFM_E_CAPACITY, "three prepared blocks are already waiting");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  int this_slot = 0;
  bool slot_was_used = false;
  PullJob job{};
  if ((rc = claim_slot(e, n_rows, nnz, row_ptr, field, feat, val, label, zero_copy, &this_slot, &slot_was_used, &job)))
    return rc;
  ffm_engine::Slot &sl = e->slots[this_slot];
  // its grouping, behind its own upload on the prep stream: planned here, submitted with the upload
  PrepPlan plan;
  if ((rc = prepare_plan(e, Rows{n_rows, nnz, sl.row_ptr, has_field ? sl.field : nullptr, sl.feat, sl.val, nullptr}, &plan)))
    return rc;
  const int64_t seq = e->n_staged_total + 1;
  const int grid_pull = e->grid_pull;
  const bool timed = !e->stage_thread_on || e->prof_on;
  hipEvent_t free_ev = slot_was_used ? sl.free_ev : nullptr;
  rc = e->submit([e, this_slot, free_ev, job, plan, seq, grid_pull, timed]() -> int {
    ScopedTimer tm("stage:submit");
    ffm_engine::Slot &s2 = e->slots[this_slot];
    int rc2 = FFM_OK;
    auto body = [&]() -> int {
      HIP_TRY(hipSetDevice(e->cfg.device_id));
      if (free_ev) HIP_TRY(hipStreamWaitEvent(e->copy, free_ev, 0));  // nothing reads its device arrays
      // (long steps: the upload too waits for the running block's row kernel to end -- beside the update
      // launches it costs nothing, beside the row kernel, which is bound by the bytes it moves, it does)
      if (e->pull_after_row && plan.ws >= 0) HIP_TRY(hipStreamWaitEvent(e->copy, e->prep_after_row ? e->ev_row_done[plan.ws] : e->ev_set_free[plan.ws], 0));
      hipLaunchKernelGGL(pull_block_kernel, dim3(grid_pull), dim3(256), 0, e->copy, job);
      HIP_TRY(hipEventRecord(s2.ev_copied, e->copy));
      if (e->copy != e->prep) HIP_TRY(hipStreamWaitEvent(e->prep, s2.ev_copied, 0));  // the grouping reads the slot
      return prepare_submit(e, plan, timed);
    };
    rc2 = body();
    {
      std::lock_guard<std::mutex> lock(e->wmu);
      e->staged_issued = seq;  // (also after a failure: nobody may wait for it forever)
    }
    return rc2;
  });
  if (rc) return rc;
  sl.used = true;
  sl.zero_copy = zero_copy != 0;
  sl.n_rows = n_rows;
  sl.nnz = nnz;
  sl.row_cap = longest;
  sl.has_field = has_field;
  sl.seq = ++e->n_staged_total;
  e->slot_next = (e->slot_next + 1) % ffm_engine::kSlots;
  e->staged[e->n_staged++] = this_slot;
  return FFM_OK;
}

// How many of the blocks staged so far have been uploaded (their host arrays are free again).
int64_t ffm_engine_blocks_pulled(ffm_engine *e) {
  if (!e || !e->slots_ready) return 0;
  e->n_pulled = std::max<int64_t>(e->n_pulled, __atomic_load_n(e->h_pulled, __ATOMIC_ACQUIRE));
  return e->n_pulled;
}

// Phase 1 (grouping is done: refresh + forward) on the oldest staged block.
int ffm_engine_train_forward_staged(ffm_engine *e, float *partial_logit) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (e->n_staged == 0) return fail(FFM_E_INVALID, "no staged block");
  if (e->has_pending) return fail(FFM_E_INVALID, "the previous block still awaits train_update");
  const int slot = e->staged[0];
  ffm_engine::Slot &sl = e->slots[slot];
  e->staged_row_cap = sl.row_cap;
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  if (int rc_w = e->wait_issued(sl.seq)) return rc_w;  // its upload + grouping launches are out
  // its grouping (behind the upload on the prep stream, or waiting for it there) implies the upload:
  // the main stream waits for ev_copied itself only when that grouping was discarded
  const Rows staged_rows{sl.n_rows, sl.nnz, sl.row_ptr, sl.has_field ? sl.field : nullptr, sl.feat, sl.val, nullptr};
  if (!(e->n_prepared > 0 && same_block(e->prepared_rows[0], staged_rows)))
    HIP_TRY(hipStreamWaitEvent(e->stream, sl.ev_copied, 0));
  int rc = ffm_engine_train_forward_device(e, sl.n_rows, sl.nnz, sl.row_ptr, sl.has_field ? sl.field : nullptr,
                                           sl.feat, sl.val, sl.label, partial_logit);
  if (rc) return rc;
  for (int i = 1; i < e->n_staged; i++) e->staged[i - 1] = e->staged[i];
  e->n_staged--;
  e->cur_slot = slot;  // released (ev_trained) by the train_update that follows
  return FFM_OK;
}

// Whole step (ffm_engine_train_batch_device) on the oldest staged block.
int ffm_engine_train_staged(ffm_engine *e, float *logit_out, double *loss_sum_out) {
  if (e && e->m.n_shards > 1)
    return fail(FFM_E_INVALID, "sharded engines train with train_forward_staged + all-reduce + train_update");
  if (e) { e->whole_step = true; e->own_logit_out = logit_out; }
  int rc = ffm_engine_train_forward_staged(e, nullptr);
  if (e) e->whole_step = false;
  if (rc) return rc;
  return ffm_engine_train_update_device(e, nullptr, logit_out, loss_sum_out);
}

// ... with its loss going into the running sum of the flush.
static int train_one_staged(ffm_engine *e) {
  int rc = ffm_engine_train_staged(e, nullptr, e->d_loss_sum);
  if (rc) return rc;
  hipLaunchKernelGGL(loss_accumulate_kernel, dim3(1), dim3(1), 0, e->stream, e->d_loss_acc, e->d_loss_sum);
  return FFM_OK;
}

int ffm_engine_train_batch_async(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                                 const int32_t *field, const int32_t *feat, const float *val,
                                 const int32_t *label) {
  if (e && e->m.n_shards > 1)
    return fail(FFM_E_INVALID, "sharded engines train with stage_batch + train_forward_staged + all-reduce + train_update");
  int rc = ffm_engine_stage_batch(e, n_rows, row_ptr, field, feat, val, label, 0);
  if (rc) return rc;
  // train what the previous call staged; the block staged just now keeps uploading and grouping
  // beside it (and beside the caller's preparation of the next one)
  while (e->n_staged > 1)
    if ((rc = train_one_staged(e))) return rc;
  return FFM_OK;
}

int ffm_engine_train_batch_async_pinned(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                                        const int32_t *field, const int32_t *feat, const float *val,
                                        const int32_t *label) {
  if (e && e->m.n_shards > 1)
    return fail(FFM_E_INVALID, "sharded engines train with stage_batch + train_forward_staged + all-reduce + train_update");
  int rc = ffm_engine_stage_batch(e, n_rows, row_ptr, field, feat, val, label, 1);
  if (rc) return rc;
  // block t+2 is staged: train block t (bench.py's schedule; the grouping of t+2 then has its
  // window beside block t and until block t+1 ends)
  while (e->n_staged > 2)
    if ((rc = train_one_staged(e))) return rc;
  return FFM_OK;
}

// Pipelined evaluation: upload through a staging slot on the side stream, predict on the main one.
// The predict launch of a block is DEFERRED by one call (the caller reads nothing before
// ffm_engine_train_flush): call t submits the upload of block t and then the predict of block t-1, so
// that the upload kernel -- 24 workgroups that must find room on the chip -- is in its queue BEFORE
// the predict kernel that fills every wave slot, not behind it (round 5: 51.9 M rows/s with the H2D
// against 68.6 M resident).  The deferred launch is made by the next call of any entry point that
// puts work on the main stream (check_block), by ffm_engine_sync / check_errors and by the flush.
int ffm_engine_predict_batch_async(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                                   const int32_t *field, const int32_t *feat, const float *val,
                                   const int32_t *label, int32_t zero_copy) {
  int32_t nnz = 0;
  int longest = 1;
  if (e && e->m.type != FFM_MODEL_FFM) field = nullptr;  // (LR / FM: no fields to upload)
  if (e) e->eval_hold = true;  // (the deferred block waits until this one's upload is submitted)
  int rc = validate_host_block(e, n_rows, row_ptr, field, feat, val, &nnz, &longest, true);
  if (e) e->eval_hold = false;
  if (rc) return rc;
  const bool has_field = field != nullptr || (e->m.type == FFM_MODEL_FFM && nnz > 0);  // (uploaded or written by the upload kernel)
  if (e->m.n_shards > 1) return fail(FFM_E_INVALID, "a sharded engine predicts through predict_batch_device + predict_finish_device");
  if (e->n_staged > 0 || e->has_pending) return fail(FFM_E_INVALID, "staged training blocks are still waiting");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  if ((rc = e->drain())) return rc;
  int this_slot = 0;
  bool slot_was_used = false;
  PullJob job{};
  if ((rc = claim_slot(e, n_rows, nnz, row_ptr, field, feat, val, label, zero_copy, &this_slot, &slot_was_used, &job)))
    return rc;
  ffm_engine::Slot &sl = e->slots[this_slot];
  if (slot_was_used && sl.free_ev) HIP_TRY(hipStreamWaitEvent(e->copy, sl.free_ev, 0));  // nothing reads its device arrays
  hipLaunchKernelGGL(pull_block_kernel, dim3(e->grid_pull), dim3(256), 0, e->copy, job);
  HIP_TRY(hipEventRecord(sl.ev_copied, e->copy));
  sl.used = true;
  sl.zero_copy = zero_copy != 0;
  sl.n_rows = n_rows;
  sl.nnz = nnz;
  sl.row_cap = longest;
  sl.has_field = has_field;
  sl.seq = ++e->n_staged_total;
  {
    std::lock_guard<std::mutex> lock(e->wmu);
    e->staged_issued = sl.seq;
  }
  e->slot_next = (e->slot_next + 1) % ffm_engine::kSlots;
  HIP_TRY(hipGetLastError());
  if ((rc = eval_launch_pending(e))) return rc;  // the block of the call before, behind this upload
  e->eval_pending.on = true;
  e->eval_pending.slot = this_slot;
  e->eval_pending.labelled = label != nullptr;
  if (e->eval_defer_off) return eval_launch_pending(e);  // (FFM_EVAL_DEFER=0: as rounds 3-5)
  return FFM_OK;
}

int ffm_engine_train_flush(ffm_engine *e, double *loss_sum_out) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  int rc;
  if ((rc = eval_launch_pending(e))) return rc;
  if (e->m.n_shards == 1)
    while (e->n_staged > 0)
      if ((rc = train_one_staged(e))) return rc;
  double total = 0.0;
  if (e->slots_ready) {
    HIP_TRY(hipMemcpyAsync(&total, e->d_loss_acc, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_loss_acc, 0, sizeof(double), e->stream));
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (loss_sum_out) *loss_sum_out = total;
  return check_device_errors(e);
}
