// trainer.h -- FtrlOffline / FtrlOnline / Evaluator over the block engine.
//
// Replaces the reference's per-sample thread pool and producer/consumer queue
// (src/task/ftrl_offline.cpp:63-103, src/task/ftrl_online.cpp:42-80, src/eval/evaluate.cpp:23-49,
// src/include/concurrent/*) with a mini-batch scheduler: rows are packed into CSR blocks on the
// host and handed to the engine one block at a time.  Same constructors, train(), evaluate(),
// has_zero_weights(), model_ptr and printed lines as the reference.
//
// Block-size ramp: block t holds min(batch_size, max(1, rows_seen / batch_ramp)) rows, so the
// staleness of the weights a row sees never exceeds 1/batch_ramp of the rows already learned
// from -- that keeps the epoch logloss within 1e-4 of the reference's strictly sequential loop
// (DESIGN.md "Batch semantics"; batch_ramp = 0 turns the ramp off; the default follows the learning
// rate: ffm_engine_default_batch_ramp(w_alpha), 32 at the reference's rates).
#pragma once
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "ftrl_model.h"
#include "csr_reader.h"
#include "csr_stream.h"
#include "reader.h"

namespace ftrl {

class BlockScheduler {
 public:
  BlockScheduler(int batch_size, int batch_ramp) : batch_(batch_size), ramp_(batch_ramp) {}
  int next_block_rows() const {
    if (ramp_ <= 0) return batch_;
    const long long r = seen_ / ramp_;
    return static_cast<int>(std::min<long long>(batch_, std::max<long long>(1, r)));
  }
  int max_block_rows() const { return batch_; }  // evaluation has no staleness: full blocks
  void consumed(int rows) { seen_ += rows; }
  long long rows_seen() const { return seen_; }

 private:
  int batch_, ramp_;
  long long seen_ = 0;
};

// A ring of page-locked blocks the device pulls from in place (FtrlModel::train_block_pinned /
// predict_block_async): an entry is refilled once the engine has uploaded the block it carried.
class BlockRing {
 public:
  static constexpr int kRing = 6;
  explicit BlockRing(FtrlModel *m) : model_(m) {}
  ~BlockRing();
  bool ready();                       // pins the entries on first use; false: no page-locked memory
  CsrBlock &acquire();                // the next entry, free to be refilled
  void handed_over(long long ordinal) { seq_[slot_] = ordinal; slot_ = (slot_ + 1) % kRing; }
  size_t row_capacity() { return ring_[0].row_ptr.capacity() - 1; }
  size_t nnz_capacity() { return ring_[0].feat.capacity(); }

 private:
  FtrlModel *model_;
  std::vector<CsrBlock> ring_;
  std::vector<long long> seq_;
  int slot_ = 0;
  bool tried_ = false;
};

// Evaluator (reference src/include/eval/evaluate.h:18-33, src/eval/evaluate.cpp:7-54): streams the
// eval file through predict() and reports the mean logloss.  Here: chunks parsed by n_threads
// workers (CsrStream), blocks gathered in page-locked memory, uploaded and predicted pipelined
// (FtrlModel::predict_block_async).
class Evaluator {
 public:
  explicit Evaluator(const config_options &opt);
  ~Evaluator();
  void load_trained_model(std::shared_ptr<FtrlModel> &train_model);
  void run();          // one pass over the eval file (PcTask::run in the reference)
  double get_loss();   // mean loss of the last pass (resets it)

 private:
  std::shared_ptr<FtrlModel> eval_model;
  std::unique_ptr<CsrStream> stream_;
  std::unique_ptr<BlockRing> ring_;
  int batch_;
  double loss_sum_ = 0.0;
  unsigned long long rows_ = 0;
};

class FtrlOffline {
 public:
  explicit FtrlOffline(const config_options &opt);
  ~FtrlOffline();
  void train();
  void evaluate(int epoch = 0);
  double one_epoch(std::vector<Sample> &samples, bool train, bool use_pool);
  bool has_zero_weights() { return model_ptr->has_zero_weights(); }

  std::unique_ptr<FtrlModel> model_ptr;

 private:
  int n_epochs, n_threads;
  uint64_t seed_;
  int epoch_no_ = 0;
  BlockScheduler sched_;
  std::unique_ptr<Reader> train_data_loader, eval_data_loader;  // API parity (data stays empty
                                                                // unless load_samples() is called)
  CsrData train_csr_, eval_csr_;                                // what train()/evaluate() walk
  bool has_eval_ = false;
  double csr_epoch(const CsrData &d, bool train);
  // Training and evaluation blocks are gathered straight into a ring of page-locked blocks which
  // the device pulls from: no host copy, three blocks in flight.
  std::unique_ptr<BlockRing> ring_;
};

class FtrlOnline {
 public:
  explicit FtrlOnline(const config_options &opt);
  void train();
  void evaluate(int epoch = 0);
  double get_loss();
  bool has_zero_weights() { return model_ptr->has_zero_weights(); }

  std::shared_ptr<FtrlModel> model_ptr;
  std::unique_ptr<Evaluator> evaluator;  // (ftrl_online.h:31; null without --eval_data)

 private:
  void run_train_file();
  int n_epochs;
  bool cmd_;
  BlockScheduler sched_;
  std::unique_ptr<CsrStream> train_stream_;  // chunks of <= 20 000 lines parsed by n_threads workers
  std::unique_ptr<BlockRing> ring_;
  double loss_sum_ = 0.0;
  unsigned long long loss_rows_ = 0;
};

}  // namespace ftrl
