/*
 * ffm_engine.h -- C ABI of the MI355X (gfx950) FTRL LR/FM/FFM training engine.
 *
 * This is the drop-in boundary for ONE path of massquantity/Ftrl-FFM: the forward + gradient +
 * FTRL per-coordinate update behind ftrl::FtrlModel (reference src/include/model/ftrl_model.h:14-51)
 * as called by FtrlOffline::one_epoch (src/task/ftrl_offline.cpp:74-83), FtrlOnline::run_task
 * (src/task/ftrl_online.cpp:70-80) and Evaluator::run_task (src/eval/evaluate.cpp:23-33).
 * Plain pointers and sizes only; no C++ or torch types.  Every entry point returns 0 on success
 * and a negative FFM_E_* code on failure (message via ffm_engine_last_error()); nothing throws.
 * The library contains no CPU fallback: without a usable HIP device ffm_engine_create fails.
 *
 * Wire format of a block of rows ("batch"): CSR over (field, feat, val) entries --
 *   row_ptr[n_rows+1] int32, field[nnz] int32, feat[nnz] int32, val[nnz] float32, label[n_rows]
 *   int32 (0/1) -- i.e. the reference's Sample{feat_vec x; int y} (src/include/data/sample.h:6-9,
 *   src/include/utils/types.h:18-19) flattened.  field may be NULL for LR/FM (libsvm rows, field 0).
 *   The host-buffer entry points that upload through a staging slot (ffm_engine_stage_batch,
 *   ffm_engine_train_batch_async[_pinned], ffm_engine_predict_batch_async, and ffm_group_train_batch[_async],
 *   which stage through them) also take field == NULL
 *   for FFM: the block's rows must then hold exactly one entry per field, entry j of a row being
 *   field j's (what python/generate_data.py:272-306 writes and a libffm file without dropped zeros
 *   is; anything else is FFM_E_INVALID) -- the field array is then written on the device instead
 *   of crossing PCIe, a third of the block's bytes.  Same results as with the array passed.
 *   Entries whose feat (FFM: or field) is out of range are ignored exactly as
 *   FtrlModel::remove_out_range / FFM::remove_out_range erase them (ftrl_model.cpp:36-42,
 *   ffm.cpp:30-36); the caller's buffers are never modified.
 *
 * Batch semantics: all rows of one train call see the weights refreshed from the
 * call-start (n,z); each touched (n,z) then receives the reference's per-sample update once per
 * touching (row, pair) in row order.  A call with n_rows == 1 is exactly one reference
 * FtrlModel::train().
 */
#ifndef FFM_ENGINE_H
#define FFM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: + ffm_engine_predict_batch_async, ffm_group_* (additions only: a caller built against 2 runs
 * unchanged) */
#define FFM_ENGINE_ABI_VERSION 4

/* ModelType, reference src/include/utils/types.h:21-25 */
enum { FFM_MODEL_LR = 0, FFM_MODEL_FM = 1, FFM_MODEL_FFM = 2 };

enum {
  FFM_OK = 0,
  FFM_E_INVALID = -1,   /* bad argument (std::invalid_argument in the reference) */
  FFM_E_DEVICE = -2,    /* HIP runtime error / no device */
  FFM_E_NOMEM = -3,     /* device allocation failed */
  FFM_E_CAPACITY = -4,  /* batch exceeds max_batch_rows / max_batch_nnz */
  FFM_E_UNSUPPORTED = -5
};

typedef struct ffm_engine ffm_engine;

/* Replaces config_options (reference src/include/utils/cmd_option.h:29-66) for this path; the
 * defaults written by ffm_engine_default_config are the reference's (:49-63). */
typedef struct ffm_engine_config {
  int32_t model_type;      /* FFM_MODEL_* ; opt.model_type */
  int32_t n_feats;         /* opt.n_feats   (10000) */
  int32_t n_fields;        /* opt.n_fields  (8); ignored for LR/FM */
  int32_t n_factors;       /* opt.n_factors (16); ignored for LR */
  float w_alpha;           /* 1e-4 */
  float w_beta;            /* 1.0 */
  float w_l1;              /* 0.1 */
  float w_l2;              /* 5.0 */
  float init_mean;         /* 0.0 */
  float init_stddev;       /* 0.02 */
  uint64_t seed;           /* the reference is unseeded (utils.h:30-36); here init is reproducible */
  int32_t max_batch_rows;  /* capacity of one train/predict call */
  int32_t max_batch_nnz;   /* capacity in entries of one call */
  int32_t device_id;       /* HIP device ordinal */
  /* Field-pair sharding of the latent tensor over the GPUs of a node (FFM; DESIGN.md
   * "Multi-GPU"): this engine owns both latent slots of the field pairs ffm_engine_shard_plan
   * assigns to shard_rank -- contiguous blocks of the field x field triangle -- and the bias /
   * linear terms that plan gives it. */
  int32_t n_shards;        /* 1 */
  int32_t shard_rank;      /* 0 */
  void *stream;            /* hipStream_t to run on; NULL = the engine creates its own */
  int32_t flags;           /* FFM_FLAG_* */
  int32_t max_row_nnz;     /* longest row a call may hold (LDS staging of the row kernels); 0 = 1024 */
  /* Per-field id ranges: field f owns the ids [field_start[f], field_start[f+1]), n_fields + 1
   * ascending values from 0 to n_feats (python/generate_data.py:272-306 and the bundled data lay
   * ids out like this).  NULL = unknown.  With it a sharded engine stores only the records of the
   * fields it has pairs of, and in each record only the slots it owns -- about 1/n_shards of the
   * (n_feats x n_fields x n_factors) tensor -- and it skips the columns of the other fields; an
   * entry whose id is outside its field's range then voids its block (FFM_E_INVALID at the next
   * sync).  Without it every shard keeps full-length records and only the work is sharded.
   * With n_shards == 1 it is a HINT: the block grouping (the mini-batch scheduler that replaces the
   * per-feature mutexes of src/include/model/ffm.h:32) sorts every field's id range by itself in one
   * launch, and blocks whose rows are one entry per field in field order take shorter forms of the
   * update; ids that sit under another field than their range's are still handled correctly.  The
   * array is copied. */
  const int32_t *field_start;
  int32_t reserved[4];
} ffm_engine_config;

enum {
  FFM_FLAG_SKIP_INIT = 1, /* leave w zeroed; the caller will ffm_engine_set_weights */
  FFM_FLAG_LEARN = 4      /* opt-in "learning" variant, NOT the reference's arithmetic (SURVEY.md
                           * 8(f) rank 4): (1) the lazy refresh keeps a latent slot's initial
                           * weight until its first gradient (n > 0) instead of overwriting it
                           * with W(0,0) = 0 (ffm.cpp:72-88, fm.cpp:69-78), (2) ffm.cpp:118 uses
                           * g2*g2 instead of g2*g1 -- so that FM / FFM factors actually train.
                           * Off: the reference bit for bit.  (Flag value 2 was round 1's
                           * optional fused row kernel: measured slower, removed.) */
};

void ffm_engine_default_config(ffm_engine_config *cfg);

/* Replaces the FtrlModel / LR / FM / FFM constructors (ftrl_model.cpp:12-34, fm.cpp:9-19,
 * ffm.cpp:17-28): allocates bias, linear and latent (w,n,z) in HBM, zeroes n,z and draws w from
 * N(init_mean, init_stddev) with a counter-based generator keyed by cfg->seed. */
int ffm_engine_create(const ffm_engine_config *cfg, ffm_engine **out);
/* The initial weights ffm_engine_create draws, recomputed on the HOST (no device needed): element
 * j of out is lin_w[first + j] (latent == 0) or vec_w flattened [feat][row_len] at first + j
 * (latent != 0).  The generator is a pure function of (seed, array, index) evaluated with
 * correctly rounded operations only (csrc/init_rng.h), so these are the device's bits -- the
 * reference's initialiser (utils.h:30-61) is unseeded and cannot be reproduced at all. */
int ffm_engine_init_weights_host(uint64_t seed, float init_mean, float init_stddev, int32_t latent,
                                 int64_t first, int64_t count, float *out);
void ffm_engine_destroy(ffm_engine *e);
const char *ffm_engine_last_error(void);
int ffm_engine_abi_version(void);
/* Block-update semantics, part of the contract (INTEGRATION.md "Block semantics"): a call with
 * n_rows == 1 is the reference's train() bit for bit (src/model/ffm.cpp:38-49); a block of several
 * rows folds the touches of every (n, z) accumulator by reductions over segments of this many
 * consecutive occurrences of its feature, joined left to right (csrc/kernels_fold.h).  Two builds
 * that return different values here train bit-different models from the same blocks. */
int ffm_engine_block_segment(void);

/* The partition behind n_shards / shard_rank, as plain host arithmetic (no device needed): the
 * fields are cut into contiguous groups and every shard owns whole group x group blocks of field
 * pairs, so the partner fields a shard owns for any field form ONE contiguous range (2 shards: 2
 * groups; 4 and 8 shards: 4 groups -- at 8 a shard needs the columns of only two groups; other
 * counts: triangular strips).  pair_owner[fa * n_fields + fb] = the shard owning the unordered pair
 * {fa, fb} (symmetric; both latent slots of a pair, (i, field_j) and (j, field_i), live there --
 * the locality ffm.cpp:63-65,104-120 needs); lin_owner[f] = the shard that adds and updates the
 * linear terms of field f's entries (with field_map == 0 all of them sit on bias_owner, because a
 * shard then cannot tell a feature's field from its id); *bias_owner = the shard holding the bias.
 * Any output may be NULL. */
int ffm_engine_shard_plan(int32_t n_fields, int32_t n_shards, int32_t field_map, int32_t *pair_owner,
                          int32_t *lin_owner, int32_t *bias_owner);

/* The block scheduler's default block-size ramp for a learning rate (the mini-batch scheduler that
 * replaces the per-sample loop of src/task/ftrl_offline.cpp:74-83 sizes block t as
 * min(batch_size, max(1, rows_seen / ramp)): the weights a row sees are then never staler than
 * 1/ramp of the rows already learned from).  32 at the reference's default w_alpha = 1e-4 and up to
 * 1e-3; above, eight times more per decade of learning rate -- 256 at 0.01, 2048 at w_alpha = 0.1
 * --, which keeps train and eval logloss within 1e-4 of the sequential loop at the rates where
 * weights really move (tests/test_gpu_scale.py pins both ends; at alpha = 0.1 a ramp of 32 is off
 * by 5e-4 and one of 689 still by 1.5e-4 on the held-out rows). */
int32_t ffm_engine_default_batch_ramp(float w_alpha);

/* Row length of the (logical) latent arrays: n_fields*n_factors (FFM), n_factors (FM), 0 (LR). */
int64_t ffm_engine_row_len(const ffm_engine *e);

/* Replaces direct access to the public members bias / lin_w / vec_w (ftrl_model.h:35-37,
 * ffm.h:25, fm.h:20).  Host arrays, row-major [feat][row_len] = the reference's save order
 * (ffm.cpp:138-146).  Any pointer may be NULL to skip that part.  Synchronous. */
int ffm_engine_set_weights(ffm_engine *e, const float *bias, const float *lin_w, const float *vec_w);
int ffm_engine_get_weights(ffm_engine *e, float *bias, float *lin_w, float *vec_w);

/* The FTRL accumulators (protected/private in the reference: ftrl_model.h:45-48, ffm.h:30-31,
 * fm.h:25-26); needed for injected-state parity tests and resumable checkpoints. */
int ffm_engine_set_state(ffm_engine *e, const float *bias_n, const float *bias_z,
                         const float *lin_n, const float *lin_z, const float *vec_n,
                         const float *vec_z);
int ffm_engine_get_state(ffm_engine *e, float *bias_n, float *bias_z, float *lin_n, float *lin_z,
                         float *vec_n, float *vec_z);

/* The same access for a LIST of features instead of the whole model: row j of every array is
 * feature feat_ids[j] (lin_* [n], vec_* [n][row_len], host arrays, any may be NULL).  What a test
 * at the headline size (33 M features: the dense arrays above would be 82 GB each) injects and
 * reads back, and the natural primitive for checkpoint deltas.  Ids must be in [0, n_feats).
 * Synchronous. */
int ffm_engine_get_rows(ffm_engine *e, int32_t n, const int32_t *feat_ids, float *lin_w,
                        float *lin_n, float *lin_z, float *vec_w, float *vec_n, float *vec_z);
int ffm_engine_set_rows(ffm_engine *e, int32_t n, const int32_t *feat_ids, const float *lin_w,
                        const float *lin_n, const float *lin_z, const float *vec_w,
                        const float *vec_n, const float *vec_z);

/* Replaces the loop over FtrlModel::train (ffm.cpp:38-49, fm.cpp:21-32, lr.cpp:9-18) in
 * FtrlOffline::one_epoch / FtrlOnline::run_task for one block of rows held in HOST memory.
 * logit_out[n_rows] receives each row's pre-update logit (train()'s return value); *loss_sum_out
 * the sum of loss(y, logit) (src/include/eval/loss.h:8-12, double).  Either may be NULL.
 * Synchronous: buffers may be reused on return. */
int ffm_engine_train_batch(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                           const int32_t *field, const int32_t *feat, const float *val,
                           const int32_t *label, float *logit_out, double *loss_sum_out);

/* Replaces the loop over FtrlModel::predict (ffm.cpp:51-55, fm.cpp:34-38, lr.cpp:20-24) in
 * Evaluator::run_task / the eval branch of one_epoch.  Uses the STORED w (no lazy refresh).
 * out[n_rows] = logit, or sigmoid(logit) when output_prob != 0; label may be NULL, otherwise
 * *loss_sum_out = sum of loss(y, logit). */
int ffm_engine_predict_batch(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                             const int32_t *field, const int32_t *feat, const float *val,
                             const int32_t *label, int32_t output_prob, float *out,
                             double *loss_sum_out);

/* Same two calls for blocks already resident in HBM (all pointers are DEVICE pointers, including
 * logit_out[n_rows] float and loss_sum_out[1] double; either output may be NULL).  Asynchronous on
 * the engine's stream; nnz is row_ptr[n_rows], passed so the host never reads device memory.
 * A row with more than max_row_nnz entries cannot be seen by the host here: the device detects it
 * before the block touches the model, the WHOLE block is then skipped (its outputs are NaN), and
 * the next ffm_engine_sync / ffm_engine_check_errors / ffm_engine_train_flush returns
 * FFM_E_CAPACITY.  (The host-buffer entry points check row lengths up front.) */
int ffm_engine_train_batch_device(ffm_engine *e, int32_t n_rows, int32_t nnz,
                                  const int32_t *row_ptr, const int32_t *field,
                                  const int32_t *feat, const float *val, const int32_t *label,
                                  float *logit_out, double *loss_sum_out);
int ffm_engine_predict_batch_device(ffm_engine *e, int32_t n_rows, int32_t nnz,
                                    const int32_t *row_ptr, const int32_t *field,
                                    const int32_t *feat, const float *val, const int32_t *label,
                                    int32_t output_prob, float *out, double *loss_sum_out);

/* Prediction on a sharded engine (n_shards > 1), split like training: phase 1 is
 * ffm_engine_predict_batch_device with label = NULL, output_prob = 0, loss_sum_out = NULL -- `out`
 * then receives this shard's PARTIAL logits (shard 0 adds bias + linear); the caller sums them
 * across shards (the same all-reduce as in training); phase 2 turns the full logits into what
 * predict() returns: out[i] = logit or sigmoid(logit) (ffm.cpp:51-55), and the sum of
 * loss(y, logit) (eval/loss.h:8-12) when label is given.  Device pointers; `out` may alias
 * `logit`.  Works on unsharded engines too. */
int ffm_engine_predict_finish_device(ffm_engine *e, int32_t n_rows, const float *logit,
                                     const int32_t *label, int32_t output_prob, float *out,
                                     double *loss_sum_out);

/* Pipelined training from host buffers -- what a trainer that only needs the epoch's loss calls
 * (FtrlOffline::one_epoch, src/task/ftrl_offline.cpp:74-83, accumulates loss(y, logit) and nothing
 * else).  Each call copies the block into a pinned staging slot (the caller's arrays are reusable
 * on return), uploads and groups it on a side stream, and starts training the block handed over
 * by the PREVIOUS call -- so the upload and grouping of block t and the caller's preparation of
 * block t+1 overlap the training of block t-1.  Blocks train in the order they were passed;
 * results are those of ffm_engine_train_batch called block by block.  ffm_engine_train_flush
 * trains the last block, waits, and returns (and resets) the sum of the losses of all blocks since
 * the previous flush.  Do not mix with other training / predict calls before the flush. */
int ffm_engine_train_batch_async(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                                 const int32_t *field, const int32_t *feat, const float *val,
                                 const int32_t *label);
int ffm_engine_train_flush(ffm_engine *e, double *loss_sum_out);
/* The same for a trainer that gathers its blocks in PAGE-LOCKED memory (hipHostMalloc, or
 * ffm_engine_pin_host on its own buffers; every array 16-byte aligned): nothing is copied on the
 * host, the device pulls the rows straight out of the caller's arrays, and the pipeline is three
 * blocks deep -- block t+2 uploads and is grouped while block t trains.  The price: the arrays of a
 * block must stay untouched until it has been uploaded, i.e. until ffm_engine_blocks_pulled() has
 * reached the block's ordinal (blocks handed over so far, through this call or the copying one,
 * counted from 1) -- a ring of five or six blocks never waits.  Same results, same flush. */
int ffm_engine_train_batch_async_pinned(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                                        const int32_t *field, const int32_t *feat, const float *val,
                                        const int32_t *label);

/* Pipelined evaluation from host buffers -- Evaluator::run_task (src/eval/evaluate.cpp:23-33) and the
 * eval branch of FtrlOffline::one_epoch (ftrl_offline.cpp:79-80) accumulate loss(y, predict(x)) and
 * nothing else: the block is uploaded through a staging slot on the side stream (zero_copy as in
 * ffm_engine_stage_batch: page-locked arrays pulled in place, untouched until
 * ffm_engine_blocks_pulled() has reached the block's ordinal) and predicted on the engine's stream,
 * so the upload of block t+1 and the caller's parsing of block t+2 overlap the forward pass of
 * block t.  Returns at once; ffm_engine_train_flush waits and returns (and resets) the sum of the
 * losses of all blocks since the previous flush.  Unsharded engines; not to be mixed with staged
 * training blocks that have not trained yet. */
int ffm_engine_predict_batch_async(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                                   const int32_t *field, const int32_t *feat, const float *val,
                                   const int32_t *label, int32_t zero_copy);

/* The two halves of ffm_engine_train_batch_async, for callers that put something between forward
 * and update -- the sharded trainer's all-reduce: ffm_engine_stage_batch copies the host block into
 * a pinned staging slot, uploads it and groups it on the side stream (returns at once; the
 * caller's arrays are reusable; FFM_E_CAPACITY when three staged blocks are already waiting);
 * ffm_engine_train_forward_staged runs phase 1 of ffm_engine_train_forward_device on the OLDEST
 * staged block (partial_logit: device, may be NULL), to be followed by
 * ffm_engine_train_update_device; ffm_engine_train_staged is the whole step
 * (ffm_engine_train_batch_device: logit_out / loss_sum_out are DEVICE pointers, may be NULL) on the
 * oldest staged block of an unsharded engine.  So rows stream host -> HBM inside the training loop
 * on every rank, overlapped with the previous block's training.
 *   zero_copy != 0: the five arrays are page-locked host memory (hipHostMalloc, hipHostRegister or
 * ffm_engine_pin_host), each 16-byte aligned, and the caller leaves them untouched until the
 * block has been uploaded -- ffm_engine_blocks_pulled() (non-blocking) has reached the block's
 * 1-based staging number, or ffm_engine_sync has returned:
 * the device pulls them straight from there (a kernel reading the mapped host memory), without the
 * copy into the engine's own staging slot -- the host's share of one 8192 x 39 block drops from a
 * 3.9 MB memcpy to one kernel launch. */
int ffm_engine_stage_batch(ffm_engine *e, int32_t n_rows, const int32_t *row_ptr,
                           const int32_t *field, const int32_t *feat, const float *val,
                           const int32_t *label, int32_t zero_copy);
int64_t ffm_engine_blocks_pulled(ffm_engine *e);
int ffm_engine_train_forward_staged(ffm_engine *e, float *partial_logit);
int ffm_engine_train_staged(ffm_engine *e, float *logit_out, double *loss_sum_out);
/* hipHostRegister / hipHostUnregister for callers that do not link the HIP runtime themselves.
 * Give it ranges that own their pages (page-aligned, whole pages: mmap, posix_memalign): the
 * runtime locks whole pages, and a range in the middle of a heap shares its first and last page
 * with whatever the allocator placed next to it. */
int ffm_engine_pin_host(void *p, size_t bytes);
int ffm_engine_unpin_host(void *p);

/* Optional look-ahead of the mini-batch scheduler: start grouping the NEXT block by feature (the
 * integer-only first stage of training) on a side stream while the current block is still being
 * updated.  The arrays must be complete in device memory when this is called and must be the very
 * ones passed to the following ffm_engine_train_batch_device / train_forward_device call (same
 * pointers and sizes); if a different block is trained next the look-ahead is discarded.  Up to
 * three blocks can be prepared ahead; they are consumed in the order they were prepared.  (Two
 * ahead is what hides the grouping: the grouping of block t+2 is scheduled to start when block
 * t-1 ends, runs beside block t's refresh and row phases and has until block t+1 ends.  Prepare
 * block t+2 BEFORE enqueuing block t's training to get that schedule.)  FFM_E_CAPACITY when
 * three prepared blocks are already waiting. */
int ffm_engine_prepare_device(ffm_engine *e, int32_t n_rows, int32_t nnz, const int32_t *row_ptr,
                              const int32_t *field, const int32_t *feat, const float *val);

/* Split-phase training for field-pair sharding over several GPUs (n_shards > 1): phase 1 groups
 * the block, refreshes this shard's weights and writes this shard's PARTIAL logits (shard 0 adds
 * bias + linear) to partial_logit[n_rows] (device); the caller sums them across shards (one RCCL
 * all-reduce of n_rows floats); phase 2 takes the summed logits and applies the updates this shard
 * owns.  ffm_engine_train_batch_device == phase 1 + phase 2 with n_shards == 1. */
int ffm_engine_train_forward_device(ffm_engine *e, int32_t n_rows, int32_t nnz,
                                    const int32_t *row_ptr, const int32_t *field,
                                    const int32_t *feat, const float *val, const int32_t *label,
                                    float *partial_logit);
int ffm_engine_train_update_device(ffm_engine *e, const float *logit, float *logit_out,
                                   double *loss_sum_out);

/* ---- several GPUs in ONE process: a group of field-pair shards --------------------------------
 * The reference has no counterpart ("distributed communication backend: none"); this
 * is what its caller FtrlOffline::one_epoch (src/task/ftrl_offline.cpp:63-103) drives when the
 * model is sharded over the GPUs of a node (BASELINE.json config 5).  ffm_group_create makes one
 * engine per entry of device_ids[n] (cfg->n_shards = n is implied, shard_rank = position; with
 * cfg->field_start every shard stores only its own slots) and one RCCL communicator per device
 * (librccl.so, bound at run time; ncclCommInitAll).  A training block is staged on EVERY engine
 * (each GPU pulls it from the caller's page-locked arrays, or from its own pinned copy), every
 * engine computes the partial logits of its field pairs, ONE ncclAllReduce(sum, float32, n_rows)
 * per block runs on the engines' own streams (ncclGroupStart/End around the n calls), every engine
 * updates its slots.  Same pipelining, ordinals and flush semantics as the one-engine entry points
 * of the same names.  Results: the one-engine results up to the association order of the
 * cross-shard logit sum (rtol 1e-5 per logit).
 *   Engines that SHARE a device (device_ids with repeats: a one-GPU dry run of the orchestration,
 * tests/test_gpu_group.py) cannot be RCCL ranks; their partial logits are summed by a kernel on
 * that device instead -- ffm_group_collective() says which of the two a group uses. */
typedef struct ffm_group ffm_group;
int ffm_group_create(const ffm_engine_config *cfg, int32_t n, const int32_t *device_ids, ffm_group **out);
void ffm_group_destroy(ffm_group *g);
int32_t ffm_group_size(const ffm_group *g);
ffm_engine *ffm_group_engine(ffm_group *g, int32_t rank);
const char *ffm_group_collective(const ffm_group *g); /* "rccl" | "device-local sum" */
/* one block, synchronous: logits (host, may be NULL) and the sum of loss(y, logit) */
int ffm_group_train_batch(ffm_group *g, int32_t n_rows, const int32_t *row_ptr, const int32_t *field,
                          const int32_t *feat, const float *val, const int32_t *label,
                          float *logit_out, double *loss_sum_out);
/* pipelined (ffm_engine_train_batch_async / _pinned): zero_copy != 0 for page-locked arrays */
int ffm_group_train_batch_async(ffm_group *g, int32_t n_rows, const int32_t *row_ptr,
                                const int32_t *field, const int32_t *feat, const float *val,
                                const int32_t *label, int32_t zero_copy);
int ffm_group_train_flush(ffm_group *g, double *loss_sum_out);
int64_t ffm_group_blocks_pulled(ffm_group *g); /* blocks every engine has uploaded */
/* predict(), synchronous: out (host, may be NULL) and the loss sum when label is given */
int ffm_group_predict_batch(ffm_group *g, int32_t n_rows, const int32_t *row_ptr, const int32_t *field,
                            const int32_t *feat, const float *val, const int32_t *label,
                            int32_t output_prob, float *out, double *loss_sum_out);

/* Measurement utility: overwrite ALL accumulators with a reproducible "warm" state drawn on the
 * device -- n ~ U[n_lo, n_hi), z ~ N(0, z_stddev) for bias, linear and latent -- so that
 * benchmarks exercise the full arithmetic (a fresh model has n = z = 0 everywhere, which makes
 * every weight take the |z| <= l1 early exit).  Not used by training itself. */
int ffm_engine_fill_state(ffm_engine *e, uint64_t seed, float n_lo, float n_hi, float z_stddev);

/* Self-test utility: y[i] = the device's sigmoid(x[i]) (utils::sigmoid<float>, utils.h:20-23, with
 * the C library's expf restated on the device); host arrays.  Lets tests compare it with the
 * host libm on millions of inputs. */
int ffm_engine_eval_sigmoid(ffm_engine *e, int32_t n, const float *x, float *y);

/* Blocks until everything queued on the engine's stream has finished; returns (and clears) the
 * error the device raised since the last report, if any -- see the _device entry points. */
int ffm_engine_sync(ffm_engine *e);
int ffm_engine_check_errors(ffm_engine *e); /* the same, named for what device callers use it for */

/* The hipStream_t the engine's kernels run on (cfg->stream, or the one it created): callers that
 * run their own work between two engine calls -- the all-reduce between train_forward_device and
 * train_update_device -- order it against this stream. */
void *ffm_engine_stream(ffm_engine *e);

/* Timing of the dominant kernel, measured with HIP events on the engine's stream around every
 * launch since the last reset (used by bench.py's roofline line). */
int ffm_engine_profile_enable(ffm_engine *e, int32_t on);
int ffm_engine_profile_read(ffm_engine *e, int32_t *n_launches, double *total_ms,
                            char *kernel_name, size_t kernel_name_cap);
/* Every timed launch costs two event records on its stream (measured: ~9 % of the step when all
 * ~17 launches are timed).  After a warm-up with everything timed, this keeps only the kernel that
 * has dominated so far and drops the other timers (and what was recorded). */
int ffm_engine_profile_focus(ffm_engine *e);
/* Text table (one line per kernel: launches, total ms, average us) into buf. */
int ffm_engine_profile_dump(ffm_engine *e, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* FFM_ENGINE_H */
