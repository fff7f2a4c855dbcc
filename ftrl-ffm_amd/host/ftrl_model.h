// ftrl_model.h -- host-side mirror of the reference's model interface
// (src/include/model/ftrl_model.h:14-51, ffm.h:11-33, fm.h:11-28, lr.h:10-17) over the C ABI of
// include/ffm_engine.h.  Same class names, same train()/predict() meaning and return values, same
// public members; the arithmetic happens in HBM behind an ffm_engine handle.
//
//   float train(feat_vec&, int)   one reference train(): returns the pre-update logit and erases
//                                 out-of-range entries from the caller's vector, as the reference.
//   float predict(feat_vec&, bool)
//   train_block / predict_block   the same for a block of rows (what the trainers call).
//
// bias / lin_w / vec_w are host mirrors of the device weights, as public as in the reference
// (ftrl_model.h:35-37, ffm.h:25, fm.h:20).  bias and lin_w (4 bytes per feature) are pulled at
// construction; vec_w is LAZY: `model.vec_w.size()` and `model.vec_w[i].size()` answer without
// moving anything, `model.vec_w[i]` pulls row i from HBM the first time it is touched, whole-model
// walks (begin()/end(), pull_all()) pull everything -- the 33 M-feature headline model has 82 GB of
// vec_w, which no constructor should copy.  pull_weights() refreshes what is mirrored after
// training, push_weights() writes edited mirrors back (the reference's tests poke them directly).
// Model files are streamed record chunk by record chunk (persist.h): host memory stays bounded.
#pragma once
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/ffm_engine.h"
#include "cmd_option.h"
#include "types.h"

namespace ftrl {

class FtrlModel;

// std::vector<std::vector<float>>-shaped view of the latent weights in HBM (see the file comment).
class LatentMirror {
 public:
  using row = std::vector<float>;
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  row &operator[](size_t i);
  row &at(size_t i);
  std::vector<row>::iterator begin() { pull_all(); return dense_.begin(); }
  std::vector<row>::iterator end() { pull_all(); return dense_.end(); }
  // `a.vec_w = b.vec_w` copies the VALUES of b's rows (all of them: b is pulled) into a's mirror;
  // a.push_weights() then writes them to a's device
  LatentMirror() = default;
  LatentMirror(const LatentMirror &) = delete;
  LatentMirror &operator=(const LatentMirror &o) {
    if (this != &o) {
      const_cast<LatentMirror &>(o).pull_all();
      dense_ = o.dense_;
      dense_ready_ = true;
      cache_.clear();
    }
    return *this;
  }
  void pull_all();                    // every row, dense (small models, the reference's tests)
  bool is_dense() const { return dense_ready_; }
  size_t rows_mirrored() const { return dense_ready_ ? n_ : cache_.size(); }

 private:
  friend class FtrlModel;
  FtrlModel *owner_ = nullptr;
  size_t n_ = 0, row_len_ = 0;
  bool dense_ready_ = false;
  std::vector<row> dense_;
  std::unordered_map<size_t, row> cache_;  // rows touched one by one
};

class FtrlModel {
 public:
  explicit FtrlModel(const config_options &opt, int model_type);
  virtual ~FtrlModel();
  FtrlModel(const FtrlModel &) = delete;
  FtrlModel &operator=(const FtrlModel &) = delete;

  virtual float train(feat_vec &features, int label);
  virtual float predict(feat_vec &features, bool output_prob);
  virtual void remove_out_range(feat_vec &feats);

  // Block entry points used by FtrlOffline / FtrlOnline.  Return sum of loss(y, logit).
  double train_block(const CsrBlock &blk, float *logit_out = nullptr);
  double predict_block(const CsrBlock &blk, bool output_prob, float *out = nullptr);
  // Pipelined training for callers that need only the loss (the trainers): queue blocks, then
  // train_flush() returns the sum of loss(y, logit) over all blocks queued since the last flush.
  // The block's arrays may be reused as soon as train_block_async returns.
  void train_block_async(const CsrBlock &blk);
  double train_flush();
  // The same without the host copy, for blocks gathered in page-locked memory: pin_block() sizes a
  // block's arrays for one engine call and page-locks them (false: could not, use the copying
  // path); train_block_pinned() hands such a block over -- it must then stay untouched until
  // blocks_pulled() has reached the number it returns (blocks handed over so far).  A block that
  // does not fit one engine call goes through the copying path instead.
  bool pin_block(CsrBlock &blk);
  void unpin_block(CsrBlock &blk);
  long long train_block_pinned(const CsrBlock &blk);
  long long blocks_pulled();
  // Pipelined evaluation (Evaluator::run_task, evaluate.cpp:23-33, needs the loss only): the block
  // is uploaded on the side stream while the previous one is predicted; `pinned`: a pin_block()ed
  // block, pulled in place and untouched until blocks_pulled() has reached the returned ordinal.
  // eval_flush() waits and returns the sum of loss(y, predict(x)) since the last flush.
  long long predict_block_async(const CsrBlock &blk, bool pinned);
  double eval_flush();
  int n_gpus() const { return n_gpus_; }

  // Model files in the reference's formats (ffm.cpp:138-200, lr.cpp:26-39); available for every
  // model type here (the reference has none for FM).  save_state/load_state add the FTRL
  // accumulators, which make a checkpoint resumable.
  void save_model(std::string_view file_name);
  void load_model(std::string_view file_name);
  void save_compressed_model(std::string_view file_name, int compress_level);
  void load_compressed_model(std::string_view file_name);
  void save_state(std::string_view file_name, int compress_level = 3);
  void load_state(std::string_view file_name);

  void pull_linear();   // device -> bias, lin_w (from the shards that own them)
  void push_linear();
  void pull_weights();  // device -> bias, lin_w and the rows of vec_w that are mirrored
  void push_weights();  // bias, lin_w and the mirrored rows of vec_w -> device
  bool has_zero_weights();

  ModelType model_type;
  float bias = 0.0f;
  std::vector<float> lin_w;
  LatentMirror vec_w;  // [n_feats][row_len], lazy; empty for LR

  // features per chunk when a whole model is streamed (files, has_zero_weights, pull_all)
  size_t stream_chunk() const;
  void get_latent_rows(int component, size_t first, size_t count, float *out);  // 0 = w, 1 = n, 2 = z
  void set_latent_rows(int component, size_t first, size_t count, const float *in);

  ffm_engine *engine() { return eng_; }
  int64_t row_len() const { return row_len_; }

 protected:
  int n_feats, n_fields, n_factors;
  int64_t row_len_ = 0;
  ffm_engine *eng_ = nullptr;   // the engine (shard 0 of the group when there is one)
  // --n_gpus > 1: one field-pair shard engine per device, RCCL all-reduce of the partial logits per
  // block (include/ffm_engine.h: ffm_group_*)
  ffm_group *grp_ = nullptr;
  int n_gpus_ = 1;
  std::vector<int32_t> lin_owner_of_field_;  // [n_fields] shard that owns a field's linear terms
  int bias_owner_ = 0;
  double eval_loss_pending_ = 0.0;  // group: evaluation blocks are predicted synchronously
  ffm_engine *shard(int r) const;
  int field_of(int feat) const { return per_field_ > 0 ? std::min(feat / per_field_, n_fields - 1) : 0; }
  int per_field_ = 0;  // ids per field under --field_ranges uniform
  CsrBlock one_;  // scratch for the one-row shims
  // engine capacities chosen at construction; blocks beyond max_nnz_ are split into several
  // engine calls (each still a block in row order), a single row beyond max_row_nnz_ is an error
  int max_rows_ = 0, max_nnz_ = 0, max_row_nnz_ = 0;
  long long handed_over_ = 0;  // blocks passed to the engine's pipelined entry points so far
  CsrBlock part_;
  // calls `fn(sub-block)` for consecutive row ranges of blk that fit the engine
  template <typename Fn> void for_each_fitting(const CsrBlock &blk, Fn fn);
};

class LR : public FtrlModel {
 public:
  explicit LR(const config_options &opt) : FtrlModel(opt, FFM_MODEL_LR) {}
};
class FM : public FtrlModel {
 public:
  explicit FM(const config_options &opt) : FtrlModel(opt, FFM_MODEL_FM) {}
};
class FFM : public FtrlModel {
 public:
  explicit FFM(const config_options &opt) : FtrlModel(opt, FFM_MODEL_FFM) {}
  void remove_out_range(feat_vec &feats) override;  // also filters the field (ffm.cpp:30-36)
};

std::unique_ptr<FtrlModel> make_model(const config_options &opt);  // throws std::invalid_argument

// loss(int y, double logit), src/include/eval/loss.h:8-12
double loss(int y, double logit);

}  // namespace ftrl
