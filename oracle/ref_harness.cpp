// ref_harness.cpp -- C entry points around the UNMODIFIED reference model classes.
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with the reference's own
// sources where they lie (/root/reference/src/model/*.cpp, src/compression/*, vendored zstd and
// header-only fmt) into oracle/_ref/libftrl_ref.so.  No reference source is copied into this repo.
// Built with -fno-access-control so the harness can read and write the private n/z arrays
// (src/include/model/ffm.h:30-31, fm.h:25-26, ftrl_model.h:45-48).
//
// The entry points mirror oracle/ffm_oracle.h one for one (fr_ instead of fo_), so the same
// python wrapper drives either and tests compare them bit for bit.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "eval/loss.h"
#include "model/ffm.h"
#include "model/fm.h"
#include "model/lr.h"
#include "utils/cmd_option.h"
#include "utils/utils.h"

namespace {

struct fr_model {
  int model_type;
  int n_feats, n_fields, n_factors;
  int64_t row_len;
  std::unique_ptr<ftrl::FtrlModel> model;
  ftrl::FFM *ffm = nullptr;
  ftrl::FM *fm = nullptr;

  std::vector<std::vector<float>> *vw() { return ffm ? &ffm->vec_w : fm ? &fm->vec_w : nullptr; }
  std::vector<std::vector<float>> *vn() { return ffm ? &ffm->vec_w_n : fm ? &fm->vec_w_n : nullptr; }
  std::vector<std::vector<float>> *vz() { return ffm ? &ffm->vec_w_z : fm ? &fm->vec_w_z : nullptr; }
};

feat_vec make_row(int nnz, const int32_t *field, const int32_t *feat, const float *val) {
  feat_vec x;
  x.reserve(nnz);
  for (int p = 0; p < nnz; p++) x.emplace_back(field ? field[p] : 0, feat[p], val[p]);
  return x;
}

void copy_out(std::vector<std::vector<float>> *v, float *dst, int64_t row_len) {
  if (!v || !dst) return;
  for (size_t i = 0; i < v->size(); i++)
    std::memcpy(dst + i * row_len, (*v)[i].data(), sizeof(float) * row_len);
}
void copy_in(std::vector<std::vector<float>> *v, const float *src, int64_t row_len) {
  if (!v || !src) return;
  for (size_t i = 0; i < v->size(); i++)
    std::memcpy((*v)[i].data(), src + i * row_len, sizeof(float) * row_len);
}

}  // namespace

extern "C" {

// NOTE: the reference constructor draws every weight from a fresh std::random_device
// (utils.h:30-36, ~32 us per weight): keep shapes small and always follow with fr_set_state.
void *fr_create(int model_type, int n_feats, int n_fields, int n_factors, float w_alpha,
                float w_beta, float w_l1, float w_l2) {
  config_options opt;
  opt.n_feats = n_feats;
  opt.n_fields = n_fields;
  opt.n_factors = n_factors;
  opt.w_alpha = w_alpha;
  opt.w_beta = w_beta;
  opt.w_l1 = w_l1;
  opt.w_l2 = w_l2;
  auto *m = new fr_model;
  m->model_type = model_type;
  m->n_feats = n_feats;
  m->n_fields = n_fields;
  m->n_factors = n_factors;
  if (model_type == 2) {
    opt.model_type = "FFM";
    auto p = std::make_unique<ftrl::FFM>(opt);
    m->ffm = p.get();
    m->model = std::move(p);
    m->row_len = static_cast<int64_t>(n_fields) * n_factors;
  } else if (model_type == 1) {
    opt.model_type = "FM";
    auto p = std::make_unique<ftrl::FM>(opt);
    m->fm = p.get();
    m->model = std::move(p);
    m->row_len = n_factors;
  } else {
    opt.model_type = "LR";
    m->model = std::make_unique<ftrl::LR>(opt);
    m->row_len = 0;
  }
  return m;
}

// The same model at a size the reference's own constructor cannot reach in reasonable time (it
// draws each of the n_feats*n_fields*n_factors weights from a fresh std::random_device, ~32 us
// apiece -- 26 minutes for the CPU baseline's 50 M weights): construct the reference object with
// ONE feature, then give its members the real size from here (possible because this harness is
// built with -fno-access-control), exactly as its constructors would (ftrl_model.cpp:28-33,
// ffm.cpp:19-27, fm.cpp:11-18) but zero-filled; callers inject the state with fr_set_state.  The
// reference's train()/predict() code is untouched and runs on these members as on its own.
void *fr_create_sized(int model_type, int n_feats, int n_fields, int n_factors, float w_alpha,
                      float w_beta, float w_l1, float w_l2) {
  auto *m = static_cast<fr_model *>(fr_create(model_type, 1, n_fields, n_factors, w_alpha, w_beta, w_l1, w_l2));
  const size_t nf = static_cast<size_t>(n_feats);
  ftrl::FtrlModel *b = m->model.get();
  b->n_feats = n_feats;
  b->lin_w.assign(nf, 0.0f);
  b->lin_w_n.assign(nf, 0.0f);
  b->lin_w_z.assign(nf, 0.0f);
  b->lin_w_mutex = std::vector<std::mutex>(nf);
  const size_t L = static_cast<size_t>(m->row_len);
  if (m->ffm) {
    m->ffm->vec_w.assign(nf, std::vector<float>(L, 0.0f));
    m->ffm->vec_w_n.assign(nf, std::vector<float>(L, 0.0f));
    m->ffm->vec_w_z.assign(nf, std::vector<float>(L, 0.0f));
    m->ffm->vec_w_mutex = std::vector<std::shared_mutex>(nf);
  } else if (m->fm) {
    m->fm->vec_w.assign(nf, std::vector<float>(L, 0.0f));
    m->fm->vec_w_n.assign(nf, std::vector<float>(L, 0.0f));
    m->fm->vec_w_z.assign(nf, std::vector<float>(L, 0.0f));
    m->fm->vec_w_mutex = std::vector<std::shared_mutex>(nf);
  }
  m->n_feats = n_feats;
  return m;
}

void fr_destroy(void *h) { delete static_cast<fr_model *>(h); }
int64_t fr_row_len(void *h) { return static_cast<fr_model *>(h)->row_len; }

// bias3 = {bias, bias_n, bias_z}; any pointer may be NULL.
void fr_get_state(void *h, float *bias3, float *lin_w, float *lin_n, float *lin_z, float *vec_w,
                  float *vec_n, float *vec_z) {
  auto *m = static_cast<fr_model *>(h);
  auto &b = *m->model;
  if (bias3) { bias3[0] = b.bias; bias3[1] = b.bias_n; bias3[2] = b.bias_z; }
  const size_t nb = sizeof(float) * static_cast<size_t>(m->n_feats);
  if (lin_w) std::memcpy(lin_w, b.lin_w.data(), nb);
  if (lin_n) std::memcpy(lin_n, b.lin_w_n.data(), nb);
  if (lin_z) std::memcpy(lin_z, b.lin_w_z.data(), nb);
  copy_out(m->vw(), vec_w, m->row_len);
  copy_out(m->vn(), vec_n, m->row_len);
  copy_out(m->vz(), vec_z, m->row_len);
}

void fr_set_state(void *h, const float *bias3, const float *lin_w, const float *lin_n,
                  const float *lin_z, const float *vec_w, const float *vec_n, const float *vec_z) {
  auto *m = static_cast<fr_model *>(h);
  auto &b = *m->model;
  if (bias3) { b.bias = bias3[0]; b.bias_n = bias3[1]; b.bias_z = bias3[2]; }
  const size_t nb = sizeof(float) * static_cast<size_t>(m->n_feats);
  if (lin_w) std::memcpy(b.lin_w.data(), lin_w, nb);
  if (lin_n) std::memcpy(b.lin_w_n.data(), lin_n, nb);
  if (lin_z) std::memcpy(b.lin_w_z.data(), lin_z, nb);
  copy_in(m->vw(), vec_w, m->row_len);
  copy_in(m->vn(), vec_n, m->row_len);
  copy_in(m->vz(), vec_z, m->row_len);
}

float fr_sgn(float x) { return utils::sgn<float>(x); }
int fr_sgn_int(int x) { return utils::sgn<int>(x); }
float fr_sigmoid(float x) { return utils::sigmoid<float>(x); }
double fr_loss(int y, double logit) { return loss(y, logit); }
float fr_maybe_zero_weight(void *h, float n, float z) {
  return static_cast<fr_model *>(h)->model->maybe_zero_weight<float>(n, z);
}

float fr_train(void *h, int nnz, const int32_t *field, const int32_t *feat, const float *val,
               int label) {
  feat_vec x = make_row(nnz, field, feat, val);
  return static_cast<fr_model *>(h)->model->train(x, label);
}

float fr_predict(void *h, int nnz, const int32_t *field, const int32_t *feat, const float *val,
                 int output_prob) {
  feat_vec x = make_row(nnz, field, feat, val);
  return static_cast<fr_model *>(h)->model->predict(x, output_prob != 0);
}

// size of the row after the model's remove_out_range (tests/test_model.cpp:27-29,46-48)
int fr_remove_out_range(void *h, int nnz, const int32_t *field, const int32_t *feat,
                        const float *val) {
  feat_vec x = make_row(nnz, field, feat, val);
  static_cast<fr_model *>(h)->model->remove_out_range(x);
  return static_cast<int>(x.size());
}

// persistence of the reference itself (src/model/ffm.cpp:138-200), for file-format cross checks
int fr_save_model(void *h, const char *path, int compressed, int level) {
  auto *m = static_cast<fr_model *>(h);
  if (!m->ffm) return -1;
  if (compressed) m->ffm->save_compressed_model(path, level); else m->ffm->save_model(path);
  return 0;
}
int fr_load_model(void *h, const char *path, int compressed) {
  auto *m = static_cast<fr_model *>(h);
  if (!m->ffm) return -1;
  if (compressed) m->ffm->load_compressed_model(path); else m->ffm->load_model(path);
  return 0;
}

// file order, one thread: what FtrlOnline::run_task does (src/task/ftrl_online.cpp:70-80)
double fr_train_rows(void *h, int n_rows, const int32_t *row_ptr, const int32_t *field,
                     const int32_t *feat, const float *val, const int32_t *label,
                     float *logit_out) {
  auto *m = static_cast<fr_model *>(h);
  double tmp_loss = 0.0;
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r], e = row_ptr[r + 1];
    feat_vec x = make_row(e - b, field ? field + b : nullptr, feat + b, val + b);
    const float logit = m->model->train(x, label[r]);
    if (logit_out) logit_out[r] = logit;
    tmp_loss += loss(label[r], logit);
  }
  return tmp_loss;
}

// The reference's threaded epoch itself (src/task/ftrl_offline.cpp:63-91, the `use_pool == false`
// branch: n_threads std::threads, contiguous chunks of rows, one shared model whose train() takes
// its own per-feature locks), on rows parsed beforehand as Reader::data holds them.  Returns the
// seconds spent in the loop -- what ftrl_offline.cpp:46-48 times; *loss_sum = sum of loss(y, logit).
// The CPU baseline's speed is validated against this (tools/validate_cpu_baseline.py).
double fr_train_rows_threaded(void *h, int n_threads, int n_rows, const int32_t *row_ptr,
                              const int32_t *field, const int32_t *feat, const float *val,
                              const int32_t *label, double *loss_sum) {
  auto *m = static_cast<fr_model *>(h);
  if (n_threads < 1) n_threads = 1;
  std::vector<feat_vec> xs(static_cast<size_t>(n_rows));
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r], e = row_ptr[r + 1];
    xs[r] = make_row(e - b, field ? field + b : nullptr, feat + b, val + b);
  }
  const size_t total = static_cast<size_t>(n_rows);
  const size_t unit = static_cast<size_t>(std::ceil(static_cast<double>(total) / n_threads));
  std::vector<double> losses(static_cast<size_t>(n_threads), 0.0);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> threads;
  for (int i = 0; i < n_threads; i++) {
    const size_t start = std::min(static_cast<size_t>(i) * unit, total);
    const size_t end = std::min(start + unit, total);
    threads.emplace_back([&, i, start, end] {
      double tmp_loss = 0.0;
      for (size_t r = start; r < end; r++) tmp_loss += loss(label[r], m->model->train(xs[r], label[r]));
      losses[static_cast<size_t>(i)] = tmp_loss;
    });
  }
  for (auto &t : threads) t.join();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double sum = 0.0;
  for (double l : losses) sum += l;
  if (loss_sum) *loss_sum = sum;
  return sec;
}

double fr_predict_batch(void *h, int n_rows, const int32_t *row_ptr, const int32_t *field,
                        const int32_t *feat, const float *val, const int32_t *label,
                        int output_prob, float *out) {
  auto *m = static_cast<fr_model *>(h);
  double tmp_loss = 0.0;
  for (int r = 0; r < n_rows; r++) {
    const int b = row_ptr[r], e = row_ptr[r + 1];
    feat_vec x = make_row(e - b, field ? field + b : nullptr, feat + b, val + b);
    const float logit = m->model->predict(x, false);
    if (out) out[r] = output_prob ? utils::sigmoid<float>(logit) : logit;
    if (label) tmp_loss += loss(label[r], logit);
  }
  return tmp_loss;
}

}  // extern "C"
