#include "ftrl_model.h"

#include "persist.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
  n_gpus_ = std::max(1, opt.n_gpus);
  int rc;
  const char *force_group = std::getenv("FFM_GROUP_RCCL");  // (a group of one: the RCCL path on a one-GPU box)
  if (n_gpus_ > 1 || (force_group && force_group[0] == '1' && mt == FFM_MODEL_FFM)) {
    // field-pair sharding needs field(i): --field_ranges uniform says field f owns the ids
    // [f * n_feats / n_fields, (f + 1) * n_feats / n_fields) (python/generate_data.py:272-306 lays
    // ids out per field); every shard then stores only the slots it owns
    if (mt != FFM_MODEL_FFM) throw std::invalid_argument("--n_gpus > 1 shards the FFM field pairs: model_type must be FFM");
    if (n_gpus_ > 1 && opt.field_ranges != "uniform")
      throw std::invalid_argument("--n_gpus > 1 needs --field_ranges uniform (per-field id ranges)");
    std::vector<int32_t> fs(static_cast<size_t>(opt.n_fields) + 1);
    if (n_gpus_ <= 1) std::printf("FFM_GROUP_RCCL=1: one engine behind the group path (synchronous evaluation, RCCL all-reduce of one rank)\n");
    if (opt.field_ranges == "uniform") {
      if (opt.n_feats < opt.n_fields)
        throw std::invalid_argument("--field_ranges uniform needs n_feats >= n_fields (every field owns an id range)");
      per_field_ = opt.n_feats / opt.n_fields;
      for (int f = 0; f <= opt.n_fields; f++) fs[f] = f == opt.n_fields ? opt.n_feats : f * per_field_;
      if (n_gpus_ > 1) cfg.field_start = fs.data();
    }
    std::vector<int32_t> devs(static_cast<size_t>(n_gpus_));
    const bool same = std::getenv("FTRL_SAME_DEVICE") != nullptr;  // one-GPU dry run of the orchestration
    for (int r = 0; r < n_gpus_; r++) devs[r] = same ? opt.device : opt.device + r;
    rc = ffm_group_create(&cfg, n_gpus_, devs.data(), &grp_);
    if (rc == FFM_E_INVALID) throw std::invalid_argument(ffm_engine_last_error());
    check(rc, "ffm_group_create");
    eng_ = ffm_group_engine(grp_, 0);
    lin_owner_of_field_.resize(static_cast<size_t>(opt.n_fields));
    int32_t bo = 0;
    check(ffm_engine_shard_plan(opt.n_fields, n_gpus_, n_gpus_ > 1 ? 1 : 0, nullptr, lin_owner_of_field_.data(), &bo), "ffm_engine_shard_plan");
    bias_owner_ = bo;
    std::printf("%d field-pair shards, collective: %s\n", n_gpus_, ffm_group_collective(grp_));
  } else {
    // one engine: --field_ranges uniform is a hint (the grouping sorts every field's id range by itself,
    // and a block whose rows are one entry per field in field order takes the short forms of the fold)
    std::vector<int32_t> fs;
    if (mt == FFM_MODEL_FFM && opt.field_ranges == "uniform" && opt.n_feats >= opt.n_fields) {
      fs.resize(static_cast<size_t>(opt.n_fields) + 1);
      const int per = opt.n_feats / opt.n_fields;
      for (int f = 0; f <= opt.n_fields; f++) fs[f] = f == opt.n_fields ? opt.n_feats : f * per;
      cfg.field_start = fs.data();  // (copied by ffm_engine_create)
    }
    rc = ffm_engine_create(&cfg, &eng_);
    if (rc == FFM_E_INVALID) throw std::invalid_argument(ffm_engine_last_error());
    check(rc, "ffm_engine_create");
  }
  row_len_ = ffm_engine_row_len(eng_);
  lin_w.resize(static_cast<size_t>(n_feats));
  vec_w.owner_ = this;
  vec_w.row_len_ = static_cast<size_t>(row_len_);
  vec_w.n_ = row_len_ > 0 ? static_cast<size_t>(n_feats) : 0;
  pull_linear();
}

FtrlModel::~FtrlModel() {
  if (grp_) ffm_group_destroy(grp_); else ffm_engine_destroy(eng_);
}

ffm_engine *FtrlModel::shard(int r) const { return grp_ ? ffm_group_engine(grp_, r) : eng_; }

// bias and lin_w from their owners (one engine: the engine)
void FtrlModel::pull_linear() {
  if (!grp_) {
    check(ffm_engine_get_weights(eng_, &bias, lin_w.data(), nullptr), "ffm_engine_get_weights");
    return;
  }
  std::vector<float> tmp(lin_w.size());
  for (int r = 0; r < n_gpus_; r++) {
    float b = 0.0f;
    check(ffm_engine_get_weights(shard(r), &b, tmp.data(), nullptr), "ffm_engine_get_weights");
    if (r == bias_owner_) bias = b;
    for (size_t i = 0; i < tmp.size(); i++)
      if (lin_owner_of_field_[field_of(static_cast<int>(i))] == r) lin_w[i] = tmp[i];
  }
}
void FtrlModel::push_linear() {
  for (int r = 0; r < n_gpus_; r++)
    check(ffm_engine_set_weights(shard(r), &bias, lin_w.data(), nullptr), "ffm_engine_set_weights");
}

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
    if (grp_)
      check(ffm_group_train_batch(grp_, b.n_rows(), b.row_ptr.data(), b.field.data(), b.feat.data(),
                                  b.val.data(), b.label.data(), logit_out ? logit_out + r0 : nullptr, &loss_sum),
            "ffm_group_train_batch"), handed_over_++;  // (a group stages every block: its ordinal counts)
    else
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
    if (grp_)
      check(ffm_group_train_batch_async(grp_, b.n_rows(), b.row_ptr.data(), b.field.data(), b.feat.data(),
                                        b.val.data(), b.label.data(), 0),
            "ffm_group_train_batch_async");
    else
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
  if (grp_)
    check(ffm_group_train_batch_async(grp_, n, blk.row_ptr.data(), blk.field.data(), blk.feat.data(),
                                      blk.val.data(), blk.label.data(), 1),
          "ffm_group_train_batch_async");
  else
    check(ffm_engine_train_batch_async_pinned(eng_, n, blk.row_ptr.data(), blk.field.data(), blk.feat.data(),
                                              blk.val.data(), blk.label.data()),
          "ffm_engine_train_batch_async_pinned");
  return ++handed_over_;
}

long long FtrlModel::blocks_pulled() { return grp_ ? ffm_group_blocks_pulled(grp_) : ffm_engine_blocks_pulled(eng_); }

double FtrlModel::train_flush() {
  double loss_sum = 0.0;
  if (grp_) check(ffm_group_train_flush(grp_, &loss_sum), "ffm_group_train_flush");
  else check(ffm_engine_train_flush(eng_, &loss_sum), "ffm_engine_train_flush");
  return loss_sum;
}

long long FtrlModel::predict_block_async(const CsrBlock &blk, bool pinned) {
  const int n = blk.n_rows();
  bool fits = n <= max_rows_ && blk.row_ptr[n] <= max_nnz_;
  for (int r = 0; r < n && fits; r++) fits = blk.row_ptr[r + 1] - blk.row_ptr[r] <= max_row_nnz_;
  if (grp_ || !fits) {  // a group predicts block by block (every shard, then the sum); so do split blocks
    eval_loss_pending_ += predict_block(blk, false);
    return handed_over_;
  }
  check(ffm_engine_predict_batch_async(eng_, n, blk.row_ptr.data(), blk.field.data(), blk.feat.data(),
                                       blk.val.data(), blk.label.data(), pinned ? 1 : 0),
        "ffm_engine_predict_batch_async");
  return ++handed_over_;
}

double FtrlModel::eval_flush() {
  double loss_sum = 0.0;
  if (!grp_) check(ffm_engine_train_flush(eng_, &loss_sum), "ffm_engine_train_flush");
  loss_sum += eval_loss_pending_;
  eval_loss_pending_ = 0.0;
  return loss_sum;
}

double FtrlModel::predict_block(const CsrBlock &blk, bool output_prob, float *out) {
  double total = 0.0;
  for_each_fitting(blk, [&](const CsrBlock &b, int r0) {
    double loss_sum = 0.0;
    if (grp_)
      check(ffm_group_predict_batch(grp_, b.n_rows(), b.row_ptr.data(), b.field.data(), b.feat.data(),
                                    b.val.data(), b.label.data(), output_prob ? 1 : 0,
                                    out ? out + r0 : nullptr, &loss_sum),
            "ffm_group_predict_batch");
    else
      check(ffm_engine_predict_batch(eng_, b.n_rows(), b.row_ptr.data(), b.field.data(),
                                     b.feat.data(), b.val.data(), b.label.data(), output_prob ? 1 : 0,
                                     out ? out + r0 : nullptr, &loss_sum),
            "ffm_engine_predict_batch");
    total += loss_sum;
  });
  return total;
}

// ---- the lazy latent mirror --------------------------------------------------------------------

size_t FtrlModel::stream_chunk() const {  // ~64 MB of floats per component and chunk
  const size_t per = static_cast<size_t>(std::max<int64_t>(row_len_, 1));
  return std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(n_feats), (16u << 20) / per));
}

void FtrlModel::get_latent_rows(int component, size_t first, size_t count, float *out) {
  std::vector<int32_t> ids(count);
  for (size_t j = 0; j < count; j++) ids[j] = static_cast<int32_t>(first + j);
  std::vector<float> part;
  for (int r = 0; r < n_gpus_; r++) {
    float *dst = out;
    if (r > 0) {  // a shard stores only the slots it owns; the others read as 0: the model is the sum
      part.resize(count * static_cast<size_t>(row_len_));
      dst = part.data();
    }
    check(ffm_engine_get_rows(shard(r), static_cast<int32_t>(count), ids.data(), nullptr, nullptr, nullptr,
                              component == 0 ? dst : nullptr, component == 1 ? dst : nullptr,
                              component == 2 ? dst : nullptr),
          "ffm_engine_get_rows");
    if (r > 0)
      for (size_t i = 0; i < part.size(); i++) out[i] += part[i];
  }
}
void FtrlModel::set_latent_rows(int component, size_t first, size_t count, const float *in) {
  std::vector<int32_t> ids(count);
  for (size_t j = 0; j < count; j++) ids[j] = static_cast<int32_t>(first + j);
  for (int r = 0; r < n_gpus_; r++)  // (every shard keeps what it owns of them)
    check(ffm_engine_set_rows(shard(r), static_cast<int32_t>(count), ids.data(), nullptr, nullptr, nullptr,
                              component == 0 ? in : nullptr, component == 1 ? in : nullptr,
                              component == 2 ? in : nullptr),
          "ffm_engine_set_rows");
}

LatentMirror::row &LatentMirror::operator[](size_t i) {
  if (dense_ready_) return dense_[i];
  auto it = cache_.find(i);
  if (it != cache_.end()) return it->second;
  row r(row_len_);
  owner_->get_latent_rows(0, i, 1, r.data());
  return cache_.emplace(i, std::move(r)).first->second;
}
LatentMirror::row &LatentMirror::at(size_t i) {
  if (i >= n_) throw std::out_of_range("vec_w");
  return (*this)[i];
}
void LatentMirror::pull_all() {
  if (dense_ready_) return;
  dense_.assign(n_, row(row_len_));
  const size_t chunk = owner_->stream_chunk();
  std::vector<float> buf(chunk * row_len_);
  for (size_t f0 = 0; f0 < n_; f0 += chunk) {
    const size_t nf = std::min(chunk, n_ - f0);
    owner_->get_latent_rows(0, f0, nf, buf.data());
    for (size_t j = 0; j < nf; j++) std::copy(buf.begin() + j * row_len_, buf.begin() + (j + 1) * row_len_, dense_[f0 + j].begin());
  }
  for (auto &kv : cache_) dense_[kv.first] = std::move(kv.second);  // (rows edited before the pull win)
  cache_.clear();
  dense_ready_ = true;
}

void FtrlModel::pull_weights() {
  pull_linear();
  if (vec_w.dense_ready_) {
    vec_w.dense_ready_ = false;
    vec_w.cache_.clear();
    vec_w.pull_all();
  } else {
    for (auto &kv : vec_w.cache_) get_latent_rows(0, kv.first, 1, kv.second.data());
  }
}

void FtrlModel::push_weights() {
  push_linear();
  if (vec_w.dense_ready_) {
    const size_t chunk = stream_chunk(), rl = static_cast<size_t>(row_len_);
    std::vector<float> buf(chunk * rl);
    for (size_t f0 = 0; f0 < vec_w.n_; f0 += chunk) {
      const size_t nf = std::min(chunk, vec_w.n_ - f0);
      for (size_t j = 0; j < nf; j++) std::copy(vec_w.dense_[f0 + j].begin(), vec_w.dense_[f0 + j].end(), buf.begin() + j * rl);
      set_latent_rows(0, f0, nf, buf.data());
    }
  } else {
    for (auto &kv : vec_w.cache_) set_latent_rows(0, kv.first, 1, kv.second.data());
  }
}

// ---- model files, streamed (persist.h) -----------------------------------------------------------

void FtrlModel::save_model(std::string_view file_name) {  // ffm.cpp:163-180
  pull_linear();
  TextModelWriter w{std::string(file_name)};
  w.scalar(bias);
  for (float v : lin_w) w.scalar(v);
  const size_t chunk = stream_chunk(), rl = static_cast<size_t>(row_len_), nf = rl ? static_cast<size_t>(n_feats) : 0;
  std::vector<float> buf(chunk * rl);
  for (size_t f0 = 0; f0 < nf; f0 += chunk) {
    const size_t n = std::min(chunk, nf - f0);
    get_latent_rows(0, f0, n, buf.data());
    w.rows(buf.data(), n, rl);
  }
  w.finish();
}
void FtrlModel::load_model(std::string_view file_name) {  // ffm.cpp:182-200
  TextModelReader r{std::string(file_name)};
  bias = r.scalar();
  for (auto &v : lin_w) v = r.scalar();
  push_linear();
  const size_t chunk = stream_chunk(), rl = static_cast<size_t>(row_len_), nf = rl ? static_cast<size_t>(n_feats) : 0;
  std::vector<float> buf(chunk * rl);
  for (size_t f0 = 0; f0 < nf; f0 += chunk) {
    const size_t n = std::min(chunk, nf - f0);
    r.rows(buf.data(), n, rl);
    set_latent_rows(0, f0, n, buf.data());
  }
  vec_w.dense_ready_ = false;
  vec_w.dense_.clear();
  vec_w.cache_.clear();
}
void FtrlModel::save_compressed_model(std::string_view file_name, int compress_level) {  // ffm.cpp:138-146
  pull_linear();
  const size_t rl = static_cast<size_t>(row_len_), nf = static_cast<size_t>(n_feats);
  FloatFrameWriter w(std::string(file_name), 1 + nf + nf * rl, compress_level);
  w.write(&bias, 1);
  w.write(lin_w.data(), nf);
  const size_t chunk = stream_chunk();
  std::vector<float> buf(chunk * rl);
  for (size_t f0 = 0; rl && f0 < nf; f0 += chunk) {
    const size_t n = std::min(chunk, nf - f0);
    get_latent_rows(0, f0, n, buf.data());
    w.write(buf.data(), n * rl);
  }
  w.finish();
}
void FtrlModel::load_compressed_model(std::string_view file_name) {  // ffm.cpp:148-161
  const size_t rl = static_cast<size_t>(row_len_), nf = static_cast<size_t>(n_feats);
  FloatFrameReader r{std::string(file_name)};
  if (r.total_floats() != 1 + nf + nf * rl) throw std::runtime_error(std::string(file_name) + ": holds a model of a different shape");
  auto need = [&](float *p, size_t n) {
    if (r.read(p, n) != n) throw std::runtime_error(std::string(file_name) + ": zstd frame is truncated");
  };
  need(&bias, 1);
  need(lin_w.data(), nf);
  push_linear();
  const size_t chunk = stream_chunk();
  std::vector<float> buf(chunk * rl);
  for (size_t f0 = 0; rl && f0 < nf; f0 += chunk) {
    const size_t n = std::min(chunk, nf - f0);
    need(buf.data(), n * rl);
    set_latent_rows(0, f0, n, buf.data());
  }
  std::printf("loading from %s, floats: %zu\n", std::string(file_name).c_str(), r.total_floats());
  vec_w.dense_ready_ = false;
  vec_w.dense_.clear();
  vec_w.cache_.clear();
}

// [bias_n, bias_z, lin_n[], lin_z[], vec_n[], vec_z[]], one zstd frame
void FtrlModel::save_state(std::string_view file_name, int compress_level) {
  const size_t nf = static_cast<size_t>(n_feats), rl = static_cast<size_t>(row_len_);
  FloatFrameWriter w(std::string(file_name), 2 + 2 * nf + 2 * nf * rl, compress_level);
  float b2[2];
  std::vector<float> lin(nf);
  // bias and linear accumulators from the shards that own them
  auto linear_state = [&](bool z_row) {
    std::vector<float> tmp(nf);
    for (int r = 0; r < n_gpus_; r++) {
      float bn = 0.0f, bz = 0.0f;
      check(ffm_engine_get_state(shard(r), &bn, &bz, z_row ? nullptr : tmp.data(), z_row ? tmp.data() : nullptr, nullptr, nullptr),
            "ffm_engine_get_state");
      if (r == bias_owner_ || !grp_) { b2[0] = bn; b2[1] = bz; }
      for (size_t i = 0; i < nf; i++)
        if (!grp_ || lin_owner_of_field_[field_of(static_cast<int>(i))] == r) lin[i] = tmp[i];
    }
  };
  linear_state(false);
  w.write(b2, 2);
  w.write(lin.data(), nf);
  linear_state(true);
  w.write(lin.data(), nf);
  const size_t chunk = stream_chunk();
  std::vector<float> buf(chunk * rl);
  for (int comp = 1; comp <= 2 && rl; comp++)
    for (size_t f0 = 0; f0 < nf; f0 += chunk) {
      const size_t n = std::min(chunk, nf - f0);
      get_latent_rows(comp, f0, n, buf.data());
      w.write(buf.data(), n * rl);
    }
  w.finish();
}
void FtrlModel::load_state(std::string_view file_name) {
  const size_t nf = static_cast<size_t>(n_feats), rl = static_cast<size_t>(row_len_);
  FloatFrameReader r{std::string(file_name)};
  if (r.total_floats() != 2 + 2 * nf + 2 * nf * rl) throw std::runtime_error(std::string(file_name) + ": wrong shape");
  auto need = [&](float *p, size_t n) {
    if (r.read(p, n) != n) throw std::runtime_error(std::string(file_name) + ": zstd frame is truncated");
  };
  float b2[2];
  std::vector<float> lin(nf);
  need(b2, 2);
  need(lin.data(), nf);
  for (int r = 0; r < n_gpus_; r++)
    check(ffm_engine_set_state(shard(r), &b2[0], &b2[1], lin.data(), nullptr, nullptr, nullptr), "ffm_engine_set_state");
  need(lin.data(), nf);
  for (int r = 0; r < n_gpus_; r++)
    check(ffm_engine_set_state(shard(r), nullptr, nullptr, nullptr, lin.data(), nullptr, nullptr), "ffm_engine_set_state");
  const size_t chunk = stream_chunk();
  std::vector<float> buf(chunk * rl);
  for (int comp = 1; comp <= 2 && rl; comp++)
    for (size_t f0 = 0; f0 < nf; f0 += chunk) {
      const size_t n = std::min(chunk, nf - f0);
      need(buf.data(), n * rl);
      set_latent_rows(comp, f0, n, buf.data());
    }
}

bool FtrlModel::has_zero_weights() {  // utils.h:63-76 over lin_w, then vec_w (ftrl_offline.cpp:105-119)
  pull_linear();
  if (std::any_of(lin_w.begin(), lin_w.end(), [](float w) { return w == 0.0f; })) return true;
  const size_t chunk = stream_chunk(), rl = static_cast<size_t>(row_len_), nf = rl ? static_cast<size_t>(n_feats) : 0;
  std::vector<float> buf(chunk * rl);
  for (size_t f0 = 0; f0 < nf; f0 += chunk) {  // streamed: stops at the first chunk that holds a zero
    const size_t n = std::min(chunk, nf - f0);
    get_latent_rows(0, f0, n, buf.data());
    if (std::any_of(buf.begin(), buf.begin() + n * rl, [](float w) { return w == 0.0f; })) return true;
  }
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
