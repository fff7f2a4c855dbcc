#include "ftrl_model.h"

#include "persist.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>

namespace ftrl {

static void check(int rc, const char *what) {
  if (rc != FFM_OK)
    throw std::runtime_error(std::string(what) + ": " + ffm_engine_last_error());
}

double loss(int y, double logit) {
  const double s = 1 / (1 + std::exp(-logit));
  return -y * std::log(s) - (1 - y) * std::log(1 - s);
}

FtrlModel::FtrlModel(const config_options &opt, int mt)
    : model_type(static_cast<ModelType>(mt)),
      n_feats(opt.n_feats), n_fields(opt.n_fields), n_factors(opt.n_factors) {
  ffm_engine_config cfg;
  ffm_engine_default_config(&cfg);
  cfg.model_type = mt;
  cfg.n_feats = opt.n_feats;
  cfg.n_fields = opt.n_fields;
  cfg.n_factors = opt.n_factors;
  cfg.w_alpha = opt.w_alpha;
  cfg.w_beta = opt.w_beta;
  cfg.w_l1 = opt.w_l1;
  cfg.w_l2 = opt.w_l2;
  cfg.init_mean = opt.init_mean;
  cfg.init_stddev = opt.init_stddev;
  cfg.seed = opt.seed;
  // The reference accepts rows of any length (feat_vec).  Here one row may hold up to 4096
  // entries (the row kernels stage a row in LDS) and one engine call up to max_nnz_ entries; a
  // denser block is fed as several calls (for_each_fitting), so --batch_size never has to be
  // chosen with the data's density in mind.
  max_rows_ = std::max(1, opt.batch_size);
  max_row_nnz_ = 4096;
  max_nnz_ = static_cast<int>(std::min<long long>(std::max<long long>(256ll * max_rows_, max_row_nnz_), 1ll << 28));
  cfg.max_batch_rows = max_rows_;
  cfg.max_batch_nnz = max_nnz_;
  cfg.max_row_nnz = max_row_nnz_;
  cfg.device_id = opt.device;
  if (opt.learn) cfg.flags |= FFM_FLAG_LEARN;
  const int rc = ffm_engine_create(&cfg, &eng_);
  if (rc == FFM_E_INVALID) throw std::invalid_argument(ffm_engine_last_error());
  check(rc, "ffm_engine_create");
  row_len_ = ffm_engine_row_len(eng_);
  lin_w.resize(static_cast<size_t>(n_feats));
  if (row_len_ > 0)
    vec_w.assign(static_cast<size_t>(n_feats), std::vector<float>(static_cast<size_t>(row_len_)));
  pull_weights();
}

FtrlModel::~FtrlModel() { ffm_engine_destroy(eng_); }

void FtrlModel::remove_out_range(feat_vec &feats) {  // ftrl_model.cpp:36-42
  feats.erase(std::remove_if(feats.begin(), feats.end(),
                             [&](const feat &f) {
                               const int i = std::get<1>(f);
                               return i < 0 || i >= n_feats;
                             }),
              feats.end());
}

void FFM::remove_out_range(feat_vec &feats) {
  feats.erase(std::remove_if(feats.begin(), feats.end(),
                             [&](const feat &f) {
                               const auto [field, i, v] = f;
                               (void)v;
                               return field < 0 || i < 0 || field >= n_fields || i >= n_feats;
                             }),
              feats.end());
}

float FtrlModel::train(feat_vec &features, int label) {
  remove_out_range(features);
  one_.clear();
  one_.push(Sample{features, label});
  float logit = 0.0f;
  train_block(one_, &logit);
  return logit;
}

float FtrlModel::predict(feat_vec &features, bool output_prob) {
  remove_out_range(features);
  one_.clear();
  one_.push(Sample{features, 0});
  float out = 0.0f;
  predict_block(one_, output_prob, &out);
  return out;
}

template <typename Fn>
void FtrlModel::for_each_fitting(const CsrBlock &blk, Fn fn) {
  const int n = blk.n_rows();
  if (n <= max_rows_ && blk.row_ptr[n] <= max_nnz_) {
    bool fits = true;
    for (int r = 0; r < n && fits; r++) fits = blk.row_ptr[r + 1] - blk.row_ptr[r] <= max_row_nnz_;
    if (fits) { fn(blk, 0); return; }
  }
  int r0 = 0;
  while (r0 < n) {
    int r1 = r0;
    while (r1 < n && r1 - r0 < max_rows_ && blk.row_ptr[r1 + 1] - blk.row_ptr[r0] <= max_nnz_) {
      if (blk.row_ptr[r1 + 1] - blk.row_ptr[r1] > max_row_nnz_)
        throw std::length_error("a row has " + std::to_string(blk.row_ptr[r1 + 1] - blk.row_ptr[r1]) +
                                " entries; this engine handles up to " + std::to_string(max_row_nnz_));
      r1++;
    }
    if (r1 == r0) throw std::length_error("a row exceeds the engine's block capacity");
    const int b = blk.row_ptr[r0], e = blk.row_ptr[r1];
    part_.row_ptr.resize(static_cast<size_t>(r1 - r0) + 1);
    for (int r = r0; r <= r1; r++) part_.row_ptr[r - r0] = blk.row_ptr[r] - b;
    part_.field.assign(blk.field.begin() + b, blk.field.begin() + e);
    part_.feat.assign(blk.feat.begin() + b, blk.feat.begin() + e);
    part_.val.assign(blk.val.begin() + b, blk.val.begin() + e);
    part_.label.assign(blk.label.begin() + r0, blk.label.begin() + r1);
    fn(part_, r0);
    r0 = r1;
  }
}

double FtrlModel::train_block(const CsrBlock &blk, float *logit_out) {
  double total = 0.0;
  for_each_fitting(blk, [&](const CsrBlock &b, int r0) {
    double loss_sum = 0.0;
    check(ffm_engine_train_batch(eng_, b.n_rows(), b.row_ptr.data(), b.field.data(), b.feat.data(),
                                 b.val.data(), b.label.data(), logit_out ? logit_out + r0 : nullptr,
                                 &loss_sum),
          "ffm_engine_train_batch");
    total += loss_sum;
  });
  return total;
}

void FtrlModel::train_block_async(const CsrBlock &blk) {
  for_each_fitting(blk, [&](const CsrBlock &b, int) {
    check(ffm_engine_train_batch_async(eng_, b.n_rows(), b.row_ptr.data(), b.field.data(),
                                       b.feat.data(), b.val.data(), b.label.data()),
          "ffm_engine_train_batch_async");
    handed_over_++;
  });
}

bool FtrlModel::pin_block(CsrBlock &blk) {
  blk.row_ptr.reserve(static_cast<size_t>(max_rows_) + 1);
  blk.label.reserve(static_cast<size_t>(max_rows_));
  blk.field.reserve(static_cast<size_t>(max_nnz_));
  blk.feat.reserve(static_cast<size_t>(max_nnz_));
  blk.val.reserve(static_cast<size_t>(max_nnz_));
  void *ptr[5] = {blk.row_ptr.data(), blk.label.data(), blk.field.data(), blk.feat.data(), blk.val.data()};
  auto pages = [](size_t n) { return (4 * n + 4095) & ~static_cast<size_t>(4095); };  // (PageAllocator's)
  const size_t bytes[5] = {pages(blk.row_ptr.capacity()), pages(blk.label.capacity()), pages(blk.field.capacity()),
                           pages(blk.feat.capacity()), pages(blk.val.capacity())};
  for (int i = 0; i < 5; i++)
    if (ffm_engine_pin_host(ptr[i], bytes[i]) != FFM_OK) {
      for (int j = 0; j < i; j++) ffm_engine_unpin_host(ptr[j]);
      return false;
    }
  return true;
}

void FtrlModel::unpin_block(CsrBlock &blk) {
  void *ptr[5] = {blk.row_ptr.data(), blk.label.data(), blk.field.data(), blk.feat.data(), blk.val.data()};
  for (void *p : ptr) ffm_engine_unpin_host(p);
}

long long FtrlModel::train_block_pinned(const CsrBlock &blk) {
  const int n = blk.n_rows();
  bool fits = n <= max_rows_ && blk.row_ptr[n] <= max_nnz_;
  for (int r = 0; r < n && fits; r++) fits = blk.row_ptr[r + 1] - blk.row_ptr[r] <= max_row_nnz_;
  if (!fits) {  // (split into several calls, each copied: its arrays are free again on return)
    train_block_async(blk);
    return handed_over_;
  }
  check(ffm_engine_train_batch_async_pinned(eng_, n, blk.row_ptr.data(), blk.field.data(), blk.feat.data(),
                                            blk.val.data(), blk.label.data()),
        "ffm_engine_train_batch_async_pinned");
  return ++handed_over_;
}

long long FtrlModel::blocks_pulled() { return ffm_engine_blocks_pulled(eng_); }

double FtrlModel::train_flush() {
  double loss_sum = 0.0;
  check(ffm_engine_train_flush(eng_, &loss_sum), "ffm_engine_train_flush");
  return loss_sum;
}

double FtrlModel::predict_block(const CsrBlock &blk, bool output_prob, float *out) {
  double total = 0.0;
  for_each_fitting(blk, [&](const CsrBlock &b, int r0) {
    double loss_sum = 0.0;
    check(ffm_engine_predict_batch(eng_, b.n_rows(), b.row_ptr.data(), b.field.data(),
                                   b.feat.data(), b.val.data(), b.label.data(), output_prob ? 1 : 0,
                                   out ? out + r0 : nullptr, &loss_sum),
          "ffm_engine_predict_batch");
    total += loss_sum;
  });
  return total;
}

void FtrlModel::pull_weights() {
  std::vector<float> flat(static_cast<size_t>(n_feats) * static_cast<size_t>(row_len_));
  check(ffm_engine_get_weights(eng_, &bias, lin_w.data(), flat.empty() ? nullptr : flat.data()),
        "ffm_engine_get_weights");
  for (size_t i = 0; i < vec_w.size(); i++)
    std::copy(flat.begin() + i * row_len_, flat.begin() + (i + 1) * row_len_, vec_w[i].begin());
}

void FtrlModel::push_weights() {
  std::vector<float> flat(static_cast<size_t>(n_feats) * static_cast<size_t>(row_len_));
  for (size_t i = 0; i < vec_w.size(); i++)
    std::copy(vec_w[i].begin(), vec_w[i].end(), flat.begin() + i * row_len_);
  check(ffm_engine_set_weights(eng_, &bias, lin_w.data(), flat.empty() ? nullptr : flat.data()),
        "ffm_engine_set_weights");
}

static ModelWeights gather(FtrlModel &m, int64_t row_len) {
  m.pull_weights();
  ModelWeights w;
  w.bias = m.bias;
  w.lin_w = m.lin_w;
  w.vec_w.reserve(m.vec_w.size() * static_cast<size_t>(row_len));
  for (const auto &v : m.vec_w) w.vec_w.insert(w.vec_w.end(), v.begin(), v.end());
  return w;
}

static void scatter(FtrlModel &m, const ModelWeights &w, int64_t row_len) {
  m.bias = w.bias;
  m.lin_w = w.lin_w;
  for (size_t i = 0; i < m.vec_w.size(); i++)
    std::copy(w.vec_w.begin() + i * row_len, w.vec_w.begin() + (i + 1) * row_len, m.vec_w[i].begin());
  m.push_weights();
}

void FtrlModel::save_model(const std::string &file_name) {
  write_text_model(file_name, gather(*this, row_len_), n_feats, static_cast<size_t>(row_len_));
}
void FtrlModel::load_model(const std::string &file_name) {
  scatter(*this, read_text_model(file_name, n_feats, static_cast<size_t>(row_len_)), row_len_);
}
void FtrlModel::save_compressed_model(const std::string &file_name, int compress_level) {
  write_compressed_model(file_name, gather(*this, row_len_), compress_level);
}
void FtrlModel::load_compressed_model(const std::string &file_name) {
  scatter(*this, read_compressed_model(file_name, n_feats, static_cast<size_t>(row_len_)), row_len_);
}

void FtrlModel::save_state(const std::string &file_name, int compress_level) {
  const size_t nf = static_cast<size_t>(n_feats), nv = nf * static_cast<size_t>(row_len_);
  std::vector<float> flat(2 + 2 * nf + 2 * nv);
  check(ffm_engine_get_state(eng_, &flat[0], &flat[1], &flat[2], &flat[2 + nf],
                             nv ? &flat[2 + 2 * nf] : nullptr, nv ? &flat[2 + 2 * nf + nv] : nullptr),
        "ffm_engine_get_state");
  write_compressed_floats(file_name, flat, compress_level);
}
void FtrlModel::load_state(const std::string &file_name) {
  const size_t nf = static_cast<size_t>(n_feats), nv = nf * static_cast<size_t>(row_len_);
  const std::vector<float> flat = read_compressed_floats(file_name);
  if (flat.size() != 2 + 2 * nf + 2 * nv) throw std::runtime_error(file_name + ": wrong shape");
  check(ffm_engine_set_state(eng_, &flat[0], &flat[1], &flat[2], &flat[2 + nf],
                             nv ? &flat[2 + 2 * nf] : nullptr, nv ? &flat[2 + 2 * nf + nv] : nullptr),
        "ffm_engine_set_state");
}

bool FtrlModel::has_zero_weights() {  // utils.h:63-76 over lin_w and vec_w
  pull_weights();
  if (std::any_of(lin_w.begin(), lin_w.end(), [](float w) { return w == 0.0f; })) return true;
  for (const auto &v : vec_w)
    if (std::any_of(v.begin(), v.end(), [](float w) { return w == 0.0f; })) return true;
  return false;
}

std::unique_ptr<FtrlModel> make_model(const config_options &opt) {
  if (opt.model_type == "LR") return std::make_unique<LR>(opt);
  if (opt.model_type == "FM") return std::make_unique<FM>(opt);
  if (opt.model_type == "FFM") return std::make_unique<FFM>(opt);
  std::fprintf(stderr, "Invalid model_type: %s, expect `LR`, `FM` or `FFM`.\n", opt.model_type.c_str());
  throw std::invalid_argument("invalid model_type");
}

}  // namespace ftrl
