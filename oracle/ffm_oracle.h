/*
 * ffm_oracle.h -- CPU restatement of the Ftrl-FFM hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X engine in ftrl-ffm_amd/.  It is plain C, it is NOT part
 * of the product, and nothing under ftrl-ffm_amd/ may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED.  oracle/ref_harness.cpp + oracle/Makefile compile the unmodified
 * reference sources where they lie (/root/reference/src/model/{ftrl_model,lr,fm,ffm}.cpp) into
 * oracle/_ref/libftrl_ref.so; tests/test_oracle_golden.py checks this restatement
 * bit-for-bit against it, and the .npz vectors under tests/golden/ (made by tests/golden/make_golden.py from that
 * same build) pin it wherever /root/reference is absent.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef FFM_ORACLE_H
#define FFM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FO_LR = 0, FO_FM = 1, FO_FFM = 2 }; /* src/include/utils/types.h:21-25 */

typedef struct fo_model fo_model;

/* Model with zeroed n,z and zero w (callers inject weights; the reference init is unseeded,
 * src/include/utils/utils.h:30-36).  Row length of the latent arrays is n_fields*n_factors for FFM
 * (src/model/ffm.cpp:19-26), n_factors for FM (src/model/fm.cpp:11-17), 0 for LR. */
fo_model *fo_create(int model_type, int n_feats, int n_fields, int n_factors, float w_alpha,
                    float w_beta, float w_l1, float w_l2);
void fo_destroy(fo_model *m);

/* Opt-in "learning" variant (NOT the reference's behaviour, so not pinned
 * by it): keep a latent slot's initial weight until its first gradient, and use g2*g2 at
 * ffm.cpp:118.  learn = 0 (default) is the reference bit for bit. */
void fo_set_variant(fo_model *m, int learn);

/* Raw state access (row-major [feat][row_len], the reference's save order ffm.cpp:138-146). */
float *fo_bias3(fo_model *m); /* {bias, bias_n, bias_z} */
float *fo_lin_w(fo_model *m);
float *fo_lin_n(fo_model *m);
float *fo_lin_z(fo_model *m);
float *fo_vec_w(fo_model *m);
float *fo_vec_n(fo_model *m);
float *fo_vec_z(fo_model *m);
int64_t fo_row_len(const fo_model *m);

/* Scalar helpers pinned by the reference's own tests (tests/test_utils.cpp:13-24,40-43). */
float fo_sgn(float x);                                  /* utils.h:15-18  (sgn(0) = -1) */
float fo_sigmoid(float x);                              /* utils.h:20-23 */
double fo_loss(int y, double logit);                    /* eval/loss.h:8-12 */
float fo_maybe_zero_weight(const fo_model *m, float n, float z); /* ftrl_model.h:28-33 */

/* One reference train()/predict() call: FFM::train ffm.cpp:38-49, FM::train fm.cpp:21-32,
 * LR::train lr.cpp:9-18; predict ffm.cpp:51-55, fm.cpp:34-38, lr.cpp:20-24.
 * The row is NOT mutated; out-of-range entries are skipped as remove_out_range would erase them. */
float fo_train(fo_model *m, int nnz, const int32_t *field, const int32_t *feat, const float *val,
               int label);
float fo_predict(fo_model *m, int nnz, const int32_t *field, const int32_t *feat, const float *val,
                 int output_prob);

/* Rows in CSR, trained one after another exactly as FtrlOnline::run_task does at n_threads=1
 * (src/task/ftrl_online.cpp:70-80).  Returns sum of loss(y, logit) in double. */
double fo_train_rows(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                     const int32_t *feat, const float *val, const int32_t *label, float *logit_out);

/* Mini-batch semantics of the MI355X engine (DESIGN.md "Batch semantics"):
 *  1. every slot a row of the batch touches is refreshed from the batch-start (n,z);
 *  2. every row's logit and tmp_grad use those frozen weights;
 *  3. every touched (n,z) then receives the block's touches -- the reference's per-sample update
 *     of every touching (row, pair), w and tmp_grad frozen -- folded by REDUCTIONS in row order:
 *     n += sum g*g, z += sum g - w * sum sigma, where the sigmas of plain touches telescope to
 *     (sqrtf(n_T) - sqrtf(n_0)) / alpha and the touches of the ffm.cpp:118 kind are evaluated one
 *     by one against a prefix sum of n (ffm_oracle.c: "block update by reductions" has the exact
 *     tree: segments of FO_SEG = 16 occurrences of the feature, left to right; fo_block_segment()).  An accumulator that ONE row touches twice
 *     (multi-valued field, repeated id) keeps the row-order walk for the whole block.
 *     With n_rows == 1 this IS fo_train, bit for bit. */
int fo_block_segment(void); /* FO_SEG: must equal the engine's ffm_engine_block_segment() */
double fo_train_batch(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                      const int32_t *feat, const float *val, const int32_t *label,
                      float *logit_out);
/* The block update of rounds 1-4: step 3 as a strict walk -- every touch applied, in row order then
 * the reference's pair order, to the running (n, z).  Equal to fo_train_batch in exact arithmetic;
 * kept so that tests can bound the rounding distance between the two. */
double fo_train_batch_rowwalk(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                              const int32_t *feat, const float *val, const int32_t *label,
                              float *logit_out);
double fo_predict_batch(fo_model *m, int n_rows, const int32_t *row_ptr, const int32_t *field,
                        const int32_t *feat, const float *val, const int32_t *label,
                        int output_prob, float *out);

/* Reference-style threaded epoch for the CPU baseline: contiguous chunks of rows per thread over
 * one shared model, per-feature locks (ftrl_offline.cpp:63-91 + ffm.cpp:72-136 structure).
 * Returns seconds spent in the train loop; *loss_sum gets sum of loss(y, logit). */
double fo_train_rows_threaded(fo_model *m, int n_threads, int n_rows, const int32_t *row_ptr,
                              const int32_t *field, const int32_t *feat, const float *val,
                              const int32_t *label, double *loss_sum);

#ifdef __cplusplus
}
#endif
#endif
