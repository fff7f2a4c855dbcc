// engine_step.h -- one block of rows on the device: grouping, look-ahead, forward, update, predict
// (the *_device entry points and the copying train_batch / predict_batch).
// Part of engine.hip's translation unit (included inside its extern "C" block).

// ---- one block of rows ---------------------------------------------------------------------

static int eval_launch_pending(ffm_engine *e);
__global__ void loss_accumulate_kernel(double *acc, const double *one) { *acc += *one; }

static int check_block(ffm_engine *e, int32_t n_rows, int32_t nnz, const void *row_ptr,
                       const void *field, const void *feat, const void *val, bool implicit_fields = false) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  // an evaluation block whose predict launch predict_batch_async deferred goes first (every entry
  // point that puts work on the main stream passes here)
  if (e->eval_pending.on && !e->eval_hold)
    if (int rc_e = eval_launch_pending(e)) return rc_e;
  if (n_rows < 0 || nnz < 0) return fail(FFM_E_INVALID, "negative n_rows / nnz");
  if (n_rows > e->max_rows || nnz > e->max_nnz)
    return fail(FFM_E_CAPACITY, "block exceeds max_batch_rows / max_batch_nnz");
  if (!row_ptr || (nnz > 0 && (!feat || !val))) return fail(FFM_E_INVALID, "null CSR array");
  if (e->m.type == FFM_MODEL_FFM && nnz > 0 && !field && !implicit_fields)
    return fail(FFM_E_INVALID, "FFM requires the field array (libffm rows)");
  return FFM_OK;
}


// FFM evaluation rows one wave each (kernels_predict.h) where that kernel applies: a whole model on
// this device, k in {4, 8, 16, 32, 64}, the rows' entries within 64 KB of LDS per workgroup.
#ifndef FFM_PRED_LPP
#define FFM_PRED_LPP 4  // k = 16: lanes per pair (1, 2, 4) ...
#endif
#ifndef FFM_PRED_U
#define FFM_PRED_U 4    // ... and steps of loads in flight together
#endif
static bool launch_predict_waves(ffm_engine *e, const Rows &rows, int row_cap, float *out, int output_prob) {
  const ModelDev &m = e->m;
  if (!e->predict_waves || m.type != FFM_MODEL_FFM || m.n_shards > 1 || m.field_start || m.own_n || m.lin_own) return false;
  const int lds_cap = std::min(row_cap, kPredLdsCap);
  const size_t shmem = pred_lds_bytes(lds_cap);
  const int grid = cdiv(rows.n_rows, kPredRows), threads = 64 * kPredRows;
#define PRED_LAUNCH(LPP, VPL, U) \
  LAUNCH(e, K_PREDICT_ROW, (ffm_predict_wave_kernel<LPP, VPL, U>), grid, threads, shmem, m, rows, e->sc[e->cur], row_cap, lds_cap, out, output_prob)
  switch (m.n_factors) {
    case 4: PRED_LAUNCH(1, 1, 4); return true;
    case 8: PRED_LAUNCH(2, 1, 4); return true;
    case 16: PRED_LAUNCH(FFM_PRED_LPP, 4 / FFM_PRED_LPP, FFM_PRED_U); return true;
    case 32: PRED_LAUNCH(8, 1, 2); return true;
    case 64: PRED_LAUNCH(16, 1, 1); return true;
    default: return false;
  }
#undef PRED_LAUNCH
}

static void launch_row_kernel(ffm_engine *e, const Rows &rows, bool train, float *out, int output_prob, int own_tg = 0) {
  const int row_cap = e->staged_row_cap > 0 ? e->staged_row_cap : e->max_row_nnz;
  e->staged_row_cap = 0;
  if (rows.n_rows == 0) return;
  // (the kernels recompute the same terms capacity from the same arguments)
  const int terms_cap = e->m.type == FFM_MODEL_FM ? row_terms_cap(2, 0, e->m.n_factors)
                        : row_terms_cap(row_cap, e->m.n_shards > 1 ? e->m.rec_slots : 0, 0);
  const size_t shmem = row_lds_bytes(row_cap, e->m.n_fields, terms_cap);
  const int kid = train ? K_ROW : K_PREDICT_ROW;
  if (e->m.type == FFM_MODEL_FM && e->m.n_factors <= 64) {
    // one wave per row, lane = factor (kernels_fm.h)
    if (train) e->singles_in_row = own_tg != 0;
    const int grid = cdiv(rows.n_rows, kFmRowsPerBlock);
    if (train) LAUNCH(e, kid, fm_row_wave_kernel<true>, grid, 64 * kFmRowsPerBlock, 0, e->m, rows, e->sc[e->cur], row_cap, out, output_prob, own_tg);
    else LAUNCH(e, kid, fm_row_wave_kernel<false>, grid, 64 * kFmRowsPerBlock, 0, e->m, rows, e->sc[e->cur], row_cap, out, output_prob, 0);
  } else if (e->m.type == FFM_MODEL_FM) {
    if (train) e->singles_in_row = false;
    if (train) LAUNCH(e, kid, fm_row_kernel<true>, rows.n_rows, kRowThreads, shmem, e->m, rows, e->sc[e->cur], row_cap, out, output_prob);
    else LAUNCH(e, kid, fm_row_kernel<false>, rows.n_rows, kRowThreads, shmem, e->m, rows, e->sc[e->cur], row_cap, out, output_prob);
  } else {
    const bool vec4 = e->m.n_factors > 0 && e->m.n_factors % 4 == 0;
    const int mr = row_cap;
    int refreshed = train && e->pre_refresh ? e->refresh_mode : 0;
    if (refreshed == 3 && !(own_tg && vec4 && e->single_kernel)) refreshed = 2;
    if (train) e->singles_in_row = refreshed == 3;
    if (refreshed && rows.nnz > 0) {
      const int per = vec4 ? e->m.row_len / 4 : e->m.row_len;
      const int64_t items = static_cast<int64_t>(std::min(rows.nnz, e->max_nnz)) * per;
      const int grid = static_cast<int>(std::min<int64_t>((items + 255) / 256, 8192));
      if (vec4) LAUNCH(e, K_REFRESH, ffm_refresh_kernel<true>, grid, 256, 0, e->m, e->sc[e->cur], refreshed >= 2);
      else LAUNCH(e, K_REFRESH, ffm_refresh_kernel<false>, grid, 256, 0, e->m, e->sc[e->cur], refreshed >= 2);
    }
    // LDS parking (round 6: of w; round 4 parked (n, z)): the first vectors of w that a row's refresh
    // computes for its once-only features stay in LDS, where the row's pair phase and its in-row update
    // read them instead of fetching them back through an L2 they have left by then (128 rows in flight
    // per XCD x 31 KB of once-only w is the whole 4 MB L2).  As many as fit beside the staging arrays in
    // a 24 KB LDS budget (six rows per CU, which the kernel's 80 VGPRs allow: C5 row kernel 477 us with
    // nothing parked, 450 with 24 KB, 478 with 30 KB = five rows, 535 with 32 KB = four;
    // profiles/r06_experiments.md), at most what the longest row of the block can use.  FFM_ROW_PARK=bytes overrides
    // (0: off; test_row_kernel_lds_parking_is_bit_identical pins none / partial / default).
    int park = 0;
    size_t shmem_park = shmem;
    if (refreshed == 3) {
      static const int park_env = std::getenv("FFM_ROW_PARK") ? std::atoi(std::getenv("FFM_ROW_PARK")) : -1;
      static const int budget_env = std::getenv("FFM_ROW_PARK_BUDGET") ? std::atoi(std::getenv("FFM_ROW_PARK_BUDGET")) : 0;
      const size_t base = (shmem + 15) & ~static_cast<size_t>(15), budget = budget_env > 0 ? budget_env : 24 * 1024;
      long long bytes = park_env >= 0 ? park_env : (base < budget ? static_cast<long long>(budget - base) : 0);
      bytes = std::min<long long>(bytes, 16ll * row_cap * (e->m.row_len / 4));
      bytes = std::min<long long>(bytes, static_cast<long long>(budget) - static_cast<long long>(std::min(base, budget)));
      park = static_cast<int>(std::max<long long>(0, bytes) / 16);
      if (park > 0) shmem_park = base + 16 * static_cast<size_t>(park);
    }
    if (train && vec4) {
      if (own_tg)
        LAUNCH(e, kid, (ffm_row_kernel<true, true, true>), rows.n_rows, e->row_threads, shmem_park, e->m, rows, e->sc[e->cur], mr, out, output_prob, refreshed, own_tg, 0, park);
      else  // a shard: the logit is whole only after the all-reduce
        LAUNCH(e, kid, (ffm_row_kernel<true, true, false>), rows.n_rows, e->row_threads, shmem, e->m, rows, e->sc[e->cur], mr, out, output_prob, refreshed, 0, 0, 0);
    }
    else if (train) LAUNCH(e, kid, (ffm_row_kernel<true, false>), rows.n_rows, e->row_threads, shmem, e->m, rows, e->sc[e->cur], mr, out, output_prob, refreshed, own_tg, 0, 0);
    else if (launch_predict_waves(e, rows, row_cap, out, output_prob)) {
      // rows that may be longer than a wave stages (kernels_predict.h): the workgroup-per-row kernel
      // for those alone (its last argument: rows of at most that many entries are not its)
      if (row_cap > kPredLdsCap)
        LAUNCH(e, kid, (ffm_row_kernel<false, true>), rows.n_rows, e->row_threads, shmem, e->m, rows, e->sc[e->cur], mr, out, output_prob, 0, 0, 0, kPredLdsCap);
    }
    else if (vec4) LAUNCH(e, kid, (ffm_row_kernel<false, true>), rows.n_rows, e->row_threads, shmem, e->m, rows, e->sc[e->cur], mr, out, output_prob, 0, 0, 0, 0);
    else LAUNCH(e, kid, (ffm_row_kernel<false, false>), rows.n_rows, e->row_threads, shmem, e->m, rows, e->sc[e->cur], mr, out, output_prob, 0, 0, 0, 0);
  }
}

static bool same_block(const Rows &a, const Rows &b) {
  return a.n_rows == b.n_rows && a.nnz == b.nnz && a.row_ptr == b.row_ptr && a.field == b.field &&
         a.feat == b.feat && a.val == b.val;
}

// Zeroes the grouping's counters and per-row field masks (a kernel: hipMemsetAsync costs the
// submitting thread ~100 us per call here, a launch ~5).
__global__ __launch_bounds__(256) void group_clear_kernel(int *counters, int n_counters,
                                                          unsigned long long *rowmask, int n_mask, int *n_super) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (tid == 0) __hip_atomic_store(n_super, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int i = tid; i < n_counters; i += stride) counters[i] = 0;
  for (int i = tid; i < n_mask; i += stride) rowmask[i] = 0ull;
}

// Groups `rows` by feature into scratch set `set` on stream `st`.
// timed = false: from the staging thread (no HIP-event bookkeeping of the profiler there).
static int launch_grouping(ffm_engine *e, int set, const Rows &rows, hipStream_t st, bool timed = true) {
  Scratch &sc = e->sc[set];
  ScopedTimer tm_all("grouping:all");
  {
    ScopedTimer tm_clear("grouping:clear");
    const int n_mask = rows.nnz > 0 && sc.rowmask ? 2 * rows.n_rows : 0;
    hipLaunchKernelGGL(group_clear_kernel, dim3(std::max(1, std::min(64, cdiv(n_mask, 1024)))), dim3(256), 0, st,
                       sc.counters, kNumCounters, sc.rowmask, n_mask, sc.n_super);
  }
  if (rows.nnz > 0) {
    const int nnz = rows.nnz;
    if (timed) LAUNCH_ON(e, st, K_GROUP_KEYS, group_keys_kernel, cdiv(nnz, kGroupThreads), kGroupThreads, 0, e->m, rows, sc, e->max_row_nnz);
    else hipLaunchKernelGGL(group_keys_kernel, dim3(cdiv(nnz, kGroupThreads)), dim3(kGroupThreads), 0, st, e->m, rows, sc, e->max_row_nnz);
    if (timed) e->prof_begin(K_GROUP_SORT, st);
    {
      ScopedTimer tm_sort("grouping:sort");
      size_t bytes = e->sort_tmp_bytes;
      if (e->range_sort) {
        const size_t n_al = (static_cast<size_t>(nnz) + 63) & ~static_cast<size_t>(63);
        unsigned char *tmp = static_cast<unsigned char *>(e->d_sort_tmp[set]);
        RangeSortJob job{sc.key, sc.skey, sc.occ, reinterpret_cast<unsigned *>(tmp), reinterpret_cast<int *>(tmp + 4 * n_al),
                         e->d_sort_start, nnz, e->n_sort_ranges, static_cast<unsigned>(e->m.n_feats),
                         e->m.sort_start ? sc.counters + CNT_IRREGULAR : nullptr};
        hipLaunchKernelGGL(group_sort_ranges_kernel, dim3(e->n_sort_ranges), dim3(kSortThreads), 0, st, job);
      } else if (e->own_sort) {
        const size_t n_al = (static_cast<size_t>(nnz) + 63) & ~static_cast<size_t>(63);
        unsigned char *tmp = static_cast<unsigned char *>(e->d_sort_tmp[set]);
        SortJob job{sc.key, sc.skey, sc.occ, reinterpret_cast<unsigned *>(tmp), reinterpret_cast<int *>(tmp + 4 * n_al),
                    reinterpret_cast<unsigned *>(tmp + 8 * n_al), sc.counters + CNT_SORT_BAR,
                    sc.counters + CNT_ERROR, sc.err, nnz, static_cast<int>((e->sort_bits + 7) / 8)};
        hipLaunchKernelGGL(group_sort_kernel, dim3(sort_grid(nnz, e->sort_grid_cap)), dim3(kSortThreads), 0, st, job);
      } else
      HIP_TRY(rocprim::radix_sort_pairs<GroupSortConfig>(e->d_sort_tmp[set], bytes, sc.key, sc.skey,
                                        rocprim::counting_iterator<int>(0), sc.occ,
                                        static_cast<size_t>(nnz), 0u, e->sort_bits, st));
    }
    if (timed) e->prof_end(st);
    if (timed) LAUNCH_ON(e, st, K_GROUP_FINISH, group_finish_kernel, cdiv(nnz, kFinishThreads), kFinishThreads, 0, e->m, rows, sc);
    else hipLaunchKernelGGL(group_finish_kernel, dim3(cdiv(nnz, kFinishThreads)), dim3(kFinishThreads), 0, st, e->m, rows, sc);
  }
  HIP_TRY(hipGetLastError());
  return FFM_OK;
}

// A look-ahead grouping in two halves: the bookkeeping (caller's thread, in call order) and the
// stream operations (whoever submits: the caller's thread or the staging thread, in the same order).
struct PrepPlan {
  int set = 0;
  bool wait_free = false;  // the set carried an earlier block: wait for its ev_set_free
  int ws = -1;             // prep_window: start when this set's block has trained (-1: at once)
  Rows rows{};
};
static int prepare_plan(ffm_engine *e, const Rows &rows, PrepPlan *pl) {
  if (e->has_pending) return fail(FFM_E_INVALID, "prepare between train_forward and train_update");
  if (e->n_prepared >= ffm_engine::kSets - 1) return fail(FFM_E_CAPACITY, "three prepared blocks are already waiting");
  const int set = (e->last_set + 1) % ffm_engine::kSets;
  pl->set = set;
  pl->rows = rows;
  pl->wait_free = e->set_used[set];
  pl->ws = -1;
  if (e->prep_window) {
    // n_prepared == 1: the predecessor is prepared but not enqueued yet -> wait for the block
    // enqueued last; n_prepared == 0: the predecessor IS the block enqueued last -> the one before
    const int ws = e->trained_set[e->n_prepared >= 1 ? 0 : 1];
    if (ws >= 0 && ws != set) pl->ws = ws;
  }
  // the set is in use from now on, also when this look-ahead ends up discarded: whoever takes the
  // set next must wait for ev_set_free (recorded when the block trains or the look-ahead is dropped)
  e->set_used[set] = true;
  e->last_set = set;
  e->prepared_set[e->n_prepared] = set;
  e->prepared_rows[e->n_prepared] = rows;
  e->n_prepared++;
  return FFM_OK;
}
static int prepare_submit(ffm_engine *e, const PrepPlan &pl, bool timed) {
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  if (pl.wait_free) HIP_TRY(hipStreamWaitEvent(e->prep, e->ev_set_free[pl.set], 0));
  if (pl.ws >= 0) HIP_TRY(hipStreamWaitEvent(e->prep, e->prep_after_row ? e->ev_row_done[pl.ws] : e->ev_set_free[pl.ws], 0));
  int rc = launch_grouping(e, pl.set, pl.rows, e->prep, timed);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(e->ev_grouped[pl.set], e->prep));
  return FFM_OK;
}

int ffm_engine_prepare_device(ffm_engine *e, int32_t n_rows, int32_t nnz, const int32_t *row_ptr,
                              const int32_t *field, const int32_t *feat, const float *val) {
  int rc = check_block(e, n_rows, nnz, row_ptr, field, feat, val);
  if (rc) return rc;
  if ((rc = e->drain())) return rc;  // (staged blocks' submissions come first on the prep stream)
  PrepPlan pl;
  if ((rc = prepare_plan(e, Rows{n_rows, nnz, row_ptr, field, feat, val, nullptr}, &pl))) return rc;
  return prepare_submit(e, pl, true);
}

int ffm_engine_train_forward_device(ffm_engine *e, int32_t n_rows, int32_t nnz,
                                    const int32_t *row_ptr, const int32_t *field,
                                    const int32_t *feat, const float *val, const int32_t *label,
                                    float *partial_logit) {
  ScopedTimer tm_fwd("train:forward");
  int rc = check_block(e, n_rows, nnz, row_ptr, field, feat, val);
  if (rc) return rc;
  if (n_rows > 0 && !label) return fail(FFM_E_INVALID, "training needs labels");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  Rows rows{n_rows, nnz, row_ptr, field, feat, val, label};
  e->pending = rows;
  e->has_pending = true;
  // only train_batch_device has the whole logit in its row kernel (one shard, FFM / LR)
  e->own_tg_cur = e->whole_step && e->m.n_shards == 1 &&
                  (e->m.type != FFM_MODEL_FM || e->m.n_factors <= 64);  // (FM: fm_row_wave_kernel)
  e->whole_step = false;
  const bool use_prepared = e->n_prepared > 0 && same_block(e->prepared_rows[0], rows);
  if (e->n_prepared > 0 && !use_prepared) {
    // groupings made ahead for some other block: forget them all
    if ((rc = e->drain())) return rc;
    for (int i = 0; i < e->n_prepared; i++) HIP_TRY(hipEventRecord(e->ev_set_free[e->prepared_set[i]], e->prep));
    e->n_prepared = 0;
  }
  e->cur_prepared = use_prepared;
  if (use_prepared) {
    e->cur = e->prepared_set[0];
    for (int i = 1; i < e->n_prepared; i++) {
      e->prepared_set[i - 1] = e->prepared_set[i];
      e->prepared_rows[i - 1] = e->prepared_rows[i];
    }
    e->n_prepared--;
    HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_grouped[e->cur], 0));
  } else {
    e->cur = (e->last_set + 1) % ffm_engine::kSets;
    e->last_set = e->cur;
    if (e->set_used[e->cur]) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_set_free[e->cur], 0));
    rc = launch_grouping(e, e->cur, rows, e->stream);
    if (rc) return rc;
  }
  e->set_used[e->cur] = true;
  launch_row_kernel(e, rows, true, e->own_tg_cur ? e->own_logit_out : nullptr, 0, e->own_tg_cur ? 1 : 0);
  if (e->prep_after_row) HIP_TRY(hipEventRecord(e->ev_row_done[e->cur], e->stream));
  if (partial_logit && n_rows > 0)
    HIP_TRY(hipMemcpyAsync(partial_logit, e->sc[e->cur].logit, sizeof(float) * n_rows, hipMemcpyDeviceToDevice, e->stream));
  HIP_TRY(hipGetLastError());
  return FFM_OK;
}

int ffm_engine_train_update_device(ffm_engine *e, const float *logit, float *logit_out,
                                   double *loss_sum_out) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (!e->has_pending) return fail(FFM_E_INVALID, "train_update without a preceding train_forward");
  ScopedTimer tm_upd("train:update");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  const Rows rows = e->pending;
  e->has_pending = false;
  const float *lg = logit ? logit : e->sc[e->cur].logit;
  const bool own_tg = e->own_tg_cur && !logit;
  if (rows.n_rows > 0 && !own_tg)
    LAUNCH(e, K_TMP_GRAD, tmp_grad_kernel, cdiv(rows.n_rows, 256), 256, 0, rows.n_rows, lg, rows.label, e->sc[e->cur].tg, e->sc[e->cur].loss, logit_out);
  // Everything below runs on the main stream: the update has no long dependent chains (every
  // accumulator is folded by reductions, kernels_fold.h), so nothing needs a queue of its own.
  // FM, whole step: fm_row_wave_kernel has applied the touches of the once-only features itself
  const int fm_in_row = e->m.type == FFM_MODEL_FM && own_tg && e->singles_in_row ? 1 : 0;
  // this shard folds the bias / runs a linear update when it owns the bias / any field's linear terms
  const bool lin_owner = e->m.bias_own != 0 || e->lin_any;
  const bool ffm = e->m.type == FFM_MODEL_FFM && rows.nnz > 0;
  const bool vec4 = e->m.n_factors % 4 == 0;
  const bool masks = e->sc[e->cur].cmask != nullptr;  // (n_fields <= 64)
  const int lin_blocks = rows.nnz > 0 ? std::min(cdiv(rows.nnz, kUpdThreads), 1024) : 0;
  const int side_blocks = lin_owner && rows.n_rows > 0 ? 1 + lin_blocks : 0;
  bool loss_done = false;
  if (ffm && vec4 && masks) {
    const bool single = e->single_kernel;
    const int span4 = e->m.rec_slots * (e->m.n_factors / 4);  // 16-byte vectors of a stored record
    const bool flat = flat_pays(e, span4);
    // A shard's once-only features (their logit was only whole after the all-reduce) and, on compact
    // storage, its few-occurrence features have launches of their own: on the side stream, beside the
    // update launch (disjoint features; a rank's step is long, the two queue hops are not)
    const bool side_launches = (single && !e->singles_in_row) || flat;
    hipStream_t sst = side_launches && !e->serial ? e->aux3 : e->stream;
    if (sst != e->stream) {
      HIP_TRY(hipEventRecord(e->ev_fork, e->stream));
      HIP_TRY(hipStreamWaitEvent(sst, e->ev_fork, 0));
    }
    if (single && !e->singles_in_row) {  // (else: already applied by the row kernel)
      if (flat) LAUNCH_ON(e, sst, K_LATENT_UPDATE_SINGLE, ffm_update_single_flat_kernel, e->grid_single, kUpdThreads, 0, e->m, rows, e->sc[e->cur]);
      else if (span4 <= 64) LAUNCH_ON(e, sst, K_LATENT_UPDATE_SINGLE, ffm_update_single_kernel<1>, e->grid_single, kUpdThreads, 0, e->m, rows, e->sc[e->cur]);
      else if (span4 <= 128) LAUNCH_ON(e, sst, K_LATENT_UPDATE_SINGLE, ffm_update_single_kernel<2>, e->grid_single, kUpdThreads, 0, e->m, rows, e->sc[e->cur]);
      else LAUNCH_ON(e, sst, K_LATENT_UPDATE_SINGLE, ffm_update_single_kernel<3>, e->grid_single, kUpdThreads, 0, e->m, rows, e->sc[e->cur]);
    }
    // (the few-occurrence launch further down: beside the longest features' second pass and join,
    // which are chains of dependent loads on an otherwise idle chip)
    // (the grouping counted the block's longest features into page-locked host memory; it ran blocks
    // ahead, so its event has usually completed and the count can be read: none -> no launches)
    bool supers = rows.nnz >= e->m.super_min;
    if (supers && e->super_flag_ok && e->cur_prepared) {
      // The count is there once the block's grouping has run -- blocks ago in device time, so this wait
      // is over at once unless the caller is several steps ahead of the device, which it then stops
      // being (two steps stay queued).  Rounds 4-5 only QUERIED the event: a caller that ran ahead (a
      // zero-copy loop never blocks) found it pending in most steps and paid the two empty launches
      // and their gaps, ~30 us per C5 step (profiles/r06_experiments.md section 9).
      if (e->super_wait) HIP_TRY(hipEventSynchronize(e->ev_grouped[e->cur]));
      if (hipEventQuery(e->ev_grouped[e->cur]) == hipSuccess && e->h_super[e->cur] == 0) supers = false;
    }
    // (... and only then is the few-occurrence kernel worth deferring: a block without such features -- the
    // strong-scaling leg's 8192 rows -- has no second pass to run it beside: 0.349 -> 0.341 ms)
    const bool few_late = flat && sst != e->stream && supers;
    if (flat && !few_late) LAUNCH_ON(e, sst, K_LATENT_UPDATE_FEW, ffm_update_small_flat_kernel, e->grid_small, kUpdThreads, 0, e->m, rows, e->sc[e->cur], single ? 1 : 0);
    if (sst != e->stream && !few_late) HIP_TRY(hipEventRecord(e->ev_join, sst));
    // (the ranges' sizes are kept in 256-thread units and scaled to the launch's workgroup size)
    // One launch, or three side by side: the one launch runs every range at the register / LDS footprint
    // of the hungriest (3 waves per SIMD at k = 16); side by side the few-occurrence range gets 4 and
    // the ranges overlap as the dispatcher finds room -- C5 resident step 1.005 -> 0.95-0.97 ms, but a
    // 4096 x 8 block 0.15 -> 0.21 ms (two more launches and a fork / join on a 100 us phase), and a
    // shard's rank (its once-only and few-occurrence launches already sit on the side queue) 1.71 ->
    // 1.77 ms.
    const int split = e->update_split >= 0 ? e->update_split
                      : (!side_launches && static_cast<int64_t>(rows.nnz) * e->m.n_factors >= (1ll << 20) ? 2 : 0);
    // (small blocks of a whole model, k >= 16: eight waves per workgroup -- kernels_tile.h kWideWaves)
    const int nf = tile_nf(e);
    const bool wide = nf == 1 && !side_launches && split == 0 && rows.nnz < e->wide_max_nnz;
    const int wpb = wide ? kWideWaves : tile_waves(nf), scale = wpb / kUpdWaves, threads = 64 * wpb;
    const size_t lds = tile_lds_bytes(nf, wpb);
    // (... and to the block: a workgroup that finds its range's lists empty still costs a dispatch and
    // a few dependent loads -- 4100 of them were a fifth of a 4096 x 8 block's update)
    auto sized = [&](int grid, int per_wg, int least) { return std::max(least, std::min(grid, cdiv(rows.nnz, per_wg))); };
    const int nt = cdiv(sized(e->grid_hot, 32, 64), scale), ns = flat ? 0 : cdiv(sized(e->grid_small, 128, 64), scale);
    const int nw = cdiv(sized(e->grid_walk, 1024, 16), scale);
    const int ng = rows.nnz >= kGiantMin ? sized(e->grid_giant, 256, 32) : 0;  // workgroups that fold giant features together
    const int order = e->update_order;  // (which of the three big ranges the dispatcher sees first)
    const int lb = loss_sum_out ? loss_grid(rows.n_rows) : 0;
    const int side = side_blocks > 0 ? 1 + cdiv(lin_blocks, scale) : 0;
    const int grid = side + ng + nt + ns + nw + lb;
#define FTRL_LAUNCH_ALL(NF)                                                                                   \
    do {                                                                                                      \
      if (split == 2 && !e->serial) { /* side by side: few on aux3, giant on aux4 */                        \
        HIP_TRY(hipEventRecord(e->ev_fork2, e->stream));                                                      \
        if (ns + nw + lb > 0) {                                                                               \
          HIP_TRY(hipStreamWaitEvent(e->aux3, e->ev_fork2, 0));                                               \
          LAUNCH_ON(e, e->aux3, K_LATENT_UPDATE_FEW, (ffm_update_all_kernel<NF, UPD_FEW | UPD_REST>), ns + nw + lb, threads, 0, e->m, rows, e->sc[e->cur], \
                 0, 0, 0, ns, single ? 1 : 0, nw, lb, loss_sum_out, e->d_loss_part, order);                       \
          HIP_TRY(hipEventRecord(e->ev_join, e->aux3));                                                       \
        }                                                                                                     \
        if (ng > 0) {                                                                                         \
          HIP_TRY(hipStreamWaitEvent(e->aux4, e->ev_fork2, 0));                                               \
          LAUNCH_ON(e, e->aux4, K_LATENT_UPDATE_GIANT, (ffm_update_all_kernel<NF, UPD_GIANT>), ng, threads, lds, e->m, rows, e->sc[e->cur], \
                 0, ng, 0, 0, single ? 1 : 0, 0, 0, loss_sum_out, e->d_loss_part, order);                         \
          HIP_TRY(hipEventRecord(e->ev_join2, e->aux4));                                                      \
        }                                                                                                     \
        LAUNCH(e, K_LATENT_UPDATE, (ffm_update_all_kernel<NF, UPD_HOT | UPD_SIDE>), side + nt, threads, lds, e->m, rows, e->sc[e->cur], \
               side, 0, nt, 0, single ? 1 : 0, 0, 0, loss_sum_out, e->d_loss_part, 1);                            \
        if (ns + nw + lb > 0) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_join, 0));                          \
        if (ng > 0) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_join2, 0));                                   \
      } else if (split) { /* (timing aid: one launch per range, each under a name of its own) */              \
        if (ng > 0)                                                                                           \
          LAUNCH(e, K_LATENT_UPDATE_GIANT, (ffm_update_all_kernel<NF, UPD_GIANT>), ng, threads, lds, e->m, rows, e->sc[e->cur], \
                 0, ng, 0, 0, single ? 1 : 0, 0, 0, loss_sum_out, e->d_loss_part, order);                         \
        LAUNCH(e, K_LATENT_UPDATE, (ffm_update_all_kernel<NF, UPD_HOT | UPD_SIDE>), side + nt, threads, lds, e->m, rows, e->sc[e->cur], \
               side, 0, nt, 0, single ? 1 : 0, 0, 0, loss_sum_out, e->d_loss_part, order);                        \
        if (ns > 0)                                                                                           \
          LAUNCH(e, K_LATENT_UPDATE_FEW, (ffm_update_all_kernel<NF, UPD_FEW>), ns, threads, lds, e->m, rows, e->sc[e->cur], \
                 0, 0, 0, ns, single ? 1 : 0, 0, 0, loss_sum_out, e->d_loss_part, order);                         \
        LAUNCH(e, K_LATENT_UPDATE_WALK, (ffm_update_all_kernel<NF, UPD_REST>), nw + lb, threads, lds, e->m, rows, e->sc[e->cur], \
               0, 0, 0, 0, single ? 1 : 0, nw, lb, loss_sum_out, e->d_loss_part, order);                          \
      } else if (ns == 0) { /* a shard: its few-occurrence features have a kernel of their own, and without   \
           that range's registers the launch holds four waves per SIMD (128 VGPRs, 11 spilled) instead of    \
           three at 143: an emulated 8-GPU rank's step 1.658 -> 1.619 ms */                                   \
        LAUNCH(e, K_LATENT_UPDATE, (ffm_update_all_kernel<NF, UPD_ALL & ~UPD_FEW>), grid, threads, lds, e->m, rows, e->sc[e->cur], \
               side, ng, nt, ns, single ? 1 : 0, nw, lb, loss_sum_out, e->d_loss_part, order);                  \
      } else {                                                                                                \
        LAUNCH(e, K_LATENT_UPDATE, (ffm_update_all_kernel<NF>), grid, threads, lds, e->m, rows, e->sc[e->cur], \
               side, ng, nt, ns, single ? 1 : 0, nw, lb, loss_sum_out, e->d_loss_part, order);                  \
      }                                                                                                       \
    } while (0)
    if (wide)
      LAUNCH(e, K_LATENT_UPDATE, (ffm_update_all_kernel<1, UPD_ALL, kWideWaves>), grid, threads, lds, e->m, rows, e->sc[e->cur],
             side, ng, nt, ns, single ? 1 : 0, nw, lb, loss_sum_out, e->d_loss_part, order);
    else if (nf == 1) FTRL_LAUNCH_ALL(1);  // k >= 16
    else if (nf == 2) FTRL_LAUNCH_ALL(2);  // k = 8 / 12
    else FTRL_LAUNCH_ALL(4);               // k = 4
    if (few_late) {
      HIP_TRY(hipEventRecord(e->ev_fork2, e->stream));
      HIP_TRY(hipStreamWaitEvent(sst, e->ev_fork2, 0));
      LAUNCH_ON(e, sst, K_LATENT_UPDATE_FEW, ffm_update_small_flat_kernel, e->grid_small, kUpdThreads, 0, e->m, rows, e->sc[e->cur], single ? 1 : 0);
      HIP_TRY(hipEventRecord(e->ev_join, sst));
    }
    if (supers) {
      // the longest features' ranges: second pass (root differences) and the join of their tiles
      const int gg = e->grid_giant;
      e->prof_begin(K_LATENT_UPDATE_GIANT, e->stream);
      if (nf == 1) hipLaunchKernelGGL(ffm_update_super_b_kernel<1>, dim3(gg), dim3(kUpdThreads), 0, e->stream, e->m, rows, e->sc[e->cur]);
      else if (nf == 2) hipLaunchKernelGGL(ffm_update_super_b_kernel<2>, dim3(gg), dim3(kUpdThreads), 0, e->stream, e->m, rows, e->sc[e->cur]);
      else hipLaunchKernelGGL(ffm_update_super_b_kernel<4>, dim3(gg), dim3(kUpdThreads), 0, e->stream, e->m, rows, e->sc[e->cur]);
      hipLaunchKernelGGL(ffm_update_super_join_kernel, dim3(gg), dim3(kUpdThreads), 0, e->stream, e->m, e->sc[e->cur]);
      e->prof_end(e->stream);
    }
#undef FTRL_LAUNCH_ALL
    if (sst != e->stream) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    loss_done = loss_sum_out != nullptr;
  } else {
    if (side_blocks > 0 && e->m.type != FFM_MODEL_FM) {
      LAUNCH(e, K_BIAS_UPDATE, bias_update_kernel, 1, kUpdThreads, 0, e->m, rows.n_rows, e->sc[e->cur]);
      if (rows.nnz > 0)
        LAUNCH(e, K_LINEAR_UPDATE, linear_update_kernel, lin_blocks, kUpdThreads, 0, e->m, rows, e->sc[e->cur], 0);
    }
    if (ffm) {
      // n_factors % 4 != 0, or more than 64 fields (no field masks: every slot keeps the row-order
      // walk): the general owner for every feature, the once-only ones included
      LAUNCH(e, K_LATENT_UPDATE, ffm_update_generic_kernel, 2048, kUpdThreads, 0, e->m, rows, e->sc[e->cur], 0);
    } else if (e->m.type == FFM_MODEL_FM) {
      if (rows.nnz > 0) {
        LAUNCH(e, K_LATENT_UPDATE, fm_update_kernel, side_blocks + 2048, kUpdThreads, 0, e->m, rows, e->sc[e->cur], fm_in_row, side_blocks);
        if (rows.nnz >= e->m.giant_min)  // the giant features' segments joined
          LAUNCH(e, K_LATENT_UPDATE_GIANT, fm_update_join_kernel, 256, kUpdThreads, 0, e->m, e->sc[e->cur]);
      }
      else if (side_blocks > 0)
        LAUNCH(e, K_BIAS_UPDATE, bias_update_kernel, 1, kUpdThreads, 0, e->m, rows.n_rows, e->sc[e->cur]);
    }
  }
  if (loss_sum_out && !loss_done)
    LAUNCH(e, K_LOSS_SUM, loss_sum_kernel, loss_grid(rows.n_rows), 256, 0, rows.n_rows, e->sc[e->cur].loss, loss_sum_out, e->d_loss_part);
  HIP_TRY(hipEventRecord(e->ev_set_free[e->cur], e->stream));
  e->trained_set[1] = e->trained_set[0];
  e->trained_set[0] = e->cur;
  if (e->cur_slot >= 0) {  // a staged block: its staging slot may be refilled from here on
    e->slots[e->cur_slot].free_ev = e->ev_set_free[e->cur];  // (not recorded again before the slot's next turn: kSets blocks later)
    e->cur_slot = -1;
  }
  HIP_TRY(hipGetLastError());
  return FFM_OK;
}

int ffm_engine_train_batch_device(ffm_engine *e, int32_t n_rows, int32_t nnz,
                                  const int32_t *row_ptr, const int32_t *field,
                                  const int32_t *feat, const float *val, const int32_t *label,
                                  float *logit_out, double *loss_sum_out) {
  if (e && e->m.n_shards > 1)
    return fail(FFM_E_INVALID, "sharded engines train with train_forward + all-reduce + train_update");
  if (e) { e->whole_step = true; e->own_logit_out = logit_out; }
  int rc = ffm_engine_train_forward_device(e, n_rows, nnz, row_ptr, field, feat, val, label, nullptr);
  if (e) e->whole_step = false;
  if (rc) return rc;
  return ffm_engine_train_update_device(e, nullptr, logit_out, loss_sum_out);
}

int ffm_engine_predict_batch_device(ffm_engine *e, int32_t n_rows, int32_t nnz,
                                    const int32_t *row_ptr, const int32_t *field,
                                    const int32_t *feat, const float *val, const int32_t *label,
                                    int32_t output_prob, float *out, double *loss_sum_out) {
  int rc = check_block(e, n_rows, nnz, row_ptr, field, feat, val);
  if (rc) return rc;
  if (e->m.n_shards > 1 && (label || output_prob || loss_sum_out))
    return fail(FFM_E_INVALID, "a sharded engine predicts partial logits only (label = NULL, output_prob = 0, "
                               "no loss): sum them across shards, then ffm_engine_predict_finish_device");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  Rows rows{n_rows, nnz, row_ptr, field, feat, val, label};
  launch_row_kernel(e, rows, false, out ? out : e->d_out, output_prob);
  if (loss_sum_out && label)
    LAUNCH(e, K_LOSS_SUM, loss_sum_kernel, loss_grid(n_rows), 256, 0, n_rows, e->sc[e->cur].loss, loss_sum_out, e->d_loss_part);
  HIP_TRY(hipGetLastError());
  return FFM_OK;
}

// The predict launch of the block ffm_engine_predict_batch_async uploaded one call ago (engine_stage.h).
static int eval_launch_pending(ffm_engine *e) {
  if (!e->eval_pending.on) return FFM_OK;
  e->eval_pending.on = false;  // (first: the launch below passes check_block)
  ffm_engine::Slot &sl = e->slots[e->eval_pending.slot];
  const bool labelled = e->eval_pending.labelled;
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  HIP_TRY(hipStreamWaitEvent(e->stream, sl.ev_copied, 0));
  e->staged_row_cap = sl.row_cap;
  int rc = ffm_engine_predict_batch_device(e, sl.n_rows, sl.nnz, sl.row_ptr, sl.has_field ? sl.field : nullptr, sl.feat,
                                           sl.val, labelled ? sl.label : nullptr, 0, e->d_out,
                                           labelled ? e->d_loss_sum : nullptr);
  if (rc) return rc;
  if (labelled) hipLaunchKernelGGL(loss_accumulate_kernel, dim3(1), dim3(1), 0, e->stream, e->d_loss_acc, e->d_loss_sum);
  HIP_TRY(hipEventRecord(sl.ev_trained, e->stream));  // the slot's device arrays are free again
  sl.free_ev = sl.ev_trained;
  HIP_TRY(hipGetLastError());
  return FFM_OK;
}

int ffm_engine_predict_finish_device(ffm_engine *e, int32_t n_rows, const float *logit,
                                     const int32_t *label, int32_t output_prob, float *out,
                                     double *loss_sum_out) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (n_rows < 0 || n_rows > e->max_rows) return fail(FFM_E_CAPACITY, "n_rows out of range");
  if (n_rows > 0 && !logit) return fail(FFM_E_INVALID, "null logit array");
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  if (n_rows > 0)
    LAUNCH(e, K_TMP_GRAD, predict_finish_kernel, cdiv(n_rows, 256), 256, 0, n_rows, logit, label, output_prob, out, e->sc[e->cur].loss);
  if (loss_sum_out) {
    if (label) LAUNCH(e, K_LOSS_SUM, loss_sum_kernel, loss_grid(n_rows), 256, 0, n_rows, e->sc[e->cur].loss, loss_sum_out, e->d_loss_part);
    else HIP_TRY(hipMemsetAsync(loss_sum_out, 0, sizeof(double), e->stream));
  }
  HIP_TRY(hipGetLastError());
  return FFM_OK;
}

// Checks one block of host arrays; returns its nnz and its longest row.
// implicit_fields (the staged entry points, FFM, field == NULL): every row must then hold exactly one
// entry per field, entry j of a row being field j's -- the upload kernel writes the field array itself.
static int validate_host_block(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                               const int32_t *field, const int32_t *feat, const float *val,
                               int32_t *nnz_out, int *longest_out, bool implicit_fields = false) {
  if (!e) return fail(FFM_E_INVALID, "null engine");
  if (n_rows < 0) return fail(FFM_E_INVALID, "negative n_rows");
  if (!row_ptr) return fail(FFM_E_INVALID, "null row_ptr");
  if (n_rows > e->max_rows) return fail(FFM_E_CAPACITY, "block exceeds max_batch_rows");
  if (row_ptr[0] != 0) return fail(FFM_E_INVALID, "row_ptr[0] must be 0");
  for (int r = 0; r < n_rows; r++)
    if (row_ptr[r + 1] < row_ptr[r]) return fail(FFM_E_INVALID, "row_ptr must be non-decreasing");
  const int32_t nnz = row_ptr[n_rows];
  int rc = check_block(e, n_rows, nnz, row_ptr, field, feat, val, implicit_fields);
  if (rc) return rc;
  if (implicit_fields && !field && e->m.type == FFM_MODEL_FFM)
    for (int r = 0; r < n_rows; r++)
      if (row_ptr[r + 1] - row_ptr[r] != e->m.n_fields)
        return fail(FFM_E_INVALID, "rows without a field array need exactly one entry per field");
  int longest = 1;
  for (int r = 0; r < n_rows; r++) {
    if (row_ptr[r + 1] - row_ptr[r] > e->max_row_nnz)
      return fail(FFM_E_CAPACITY, "a row has more entries than max_row_nnz");
    longest = std::max(longest, row_ptr[r + 1] - row_ptr[r]);
  }
  *nnz_out = nnz;
  *longest_out = longest;
  return FFM_OK;
}

static int stage_block(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr, const int32_t *field,
                       const int32_t *feat, const float *val, const int32_t *label, int32_t *nnz_out) {
  int32_t nnz = 0;
  int longest = 1;
  if (e && e->m.type != FFM_MODEL_FFM) field = nullptr;  // (LR / FM: no fields to upload)
  int rc = validate_host_block(e, n_rows, row_ptr, field, feat, val, &nnz, &longest);
  if (rc) return rc;
  e->staged_row_cap = longest;
  HIP_TRY(hipSetDevice(e->cfg.device_id));
  HIP_TRY(hipMemcpyAsync(e->d_row_ptr, row_ptr, sizeof(int32_t) * (n_rows + 1), hipMemcpyHostToDevice, e->stream));
  if (nnz > 0) {
    if (field) HIP_TRY(hipMemcpyAsync(e->d_field, field, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_feat, feat, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_val, val, sizeof(float) * nnz, hipMemcpyHostToDevice, e->stream));
  }
  if (label && n_rows > 0)
    HIP_TRY(hipMemcpyAsync(e->d_label, label, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, e->stream));
  *nnz_out = nnz;
  return FFM_OK;
}

int ffm_engine_train_batch(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                           const int32_t *field, const int32_t *feat, const float *val,
                           const int32_t *label, float *logit_out, double *loss_sum_out) {
  int32_t nnz = 0;
  int rc = stage_block(e, n_rows, row_ptr, field, feat, val, label, &nnz);
  if (rc) return rc;
  if (n_rows > 0 && !label) return fail(FFM_E_INVALID, "training needs labels");
  rc = ffm_engine_train_batch_device(e, n_rows, nnz, e->d_row_ptr, (field && e->m.type == FFM_MODEL_FFM) ? e->d_field : nullptr,
                                     e->d_feat, e->d_val, e->d_label, e->d_out, e->d_loss_sum);
  if (rc) return rc;
  if (logit_out && n_rows > 0)
    HIP_TRY(hipMemcpyAsync(logit_out, e->d_out, sizeof(float) * n_rows, hipMemcpyDeviceToHost, e->stream));
  if (loss_sum_out)
    HIP_TRY(hipMemcpyAsync(loss_sum_out, e->d_loss_sum, sizeof(double), hipMemcpyDeviceToHost, e->stream));
  return check_device_errors(e);
}

int ffm_engine_predict_batch(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                             const int32_t *field, const int32_t *feat, const float *val,
                             const int32_t *label, int32_t output_prob, float *out,
                             double *loss_sum_out) {
  int32_t nnz = 0;
  int rc = stage_block(e, n_rows, row_ptr, field, feat, val, label, &nnz);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(e->sc[e->cur].counters, 0, kNumCounters * sizeof(int), e->stream));
  rc = ffm_engine_predict_batch_device(e, n_rows, nnz, e->d_row_ptr, (field && e->m.type == FFM_MODEL_FFM) ? e->d_field : nullptr,
                                       e->d_feat, e->d_val, label ? e->d_label : nullptr,
                                       output_prob, e->d_out, e->d_loss_sum);
  if (rc) return rc;
  if (out && n_rows > 0)
    HIP_TRY(hipMemcpyAsync(out, e->d_out, sizeof(float) * n_rows, hipMemcpyDeviceToHost, e->stream));
  if (loss_sum_out) {
    if (label) HIP_TRY(hipMemcpyAsync(loss_sum_out, e->d_loss_sum, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    else *loss_sum_out = 0.0;
  }
  return check_device_errors(e);
}
