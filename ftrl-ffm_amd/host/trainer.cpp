#include "trainer.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <thread>

namespace ftrl {

using timer = std::chrono::steady_clock;
static double seconds_since(timer::time_point t0) {
  return std::chrono::duration<double>(timer::now() - t0).count();
}

// ---------------- offline: data in memory, seeded shuffle per epoch ----------------

FtrlOffline::FtrlOffline(const config_options &opt)
    : model_ptr(make_model(opt)), n_epochs(opt.epoch), n_threads(opt.thread_num), seed_(opt.seed),
      sched_(opt.batch_size, opt.batch_ramp < 0 ? ffm_engine_default_batch_ramp(opt.w_alpha) : opt.batch_ramp) {
  // the files go straight into CSR (csr_reader.h); the Sample-based readers exist for callers
  // that use one_epoch(std::vector<Sample>&, ...) as the reference's tests do
  train_data_loader = std::make_unique<Reader>(opt.file_type);
  std::printf("Loading data from file: %s\n", opt.train_path.c_str());
  const auto t0 = timer::now();
  train_csr_ = load_csr(opt.train_path, opt.file_type, n_threads);
  std::printf("Total number of samples loaded: %zu\nparsing data time: %.4lfs\n",
              train_csr_.n_rows(), seconds_since(t0));
  if (!opt.eval_path.empty()) {
    eval_data_loader = std::make_unique<Reader>(opt.file_type);
    eval_csr_ = load_csr(opt.eval_path, opt.file_type, n_threads);
    has_eval_ = true;
  }
}

FtrlOffline::~FtrlOffline() = default;

// ---------------- the ring of page-locked blocks ----------------

BlockRing::~BlockRing() {
  for (auto &b : ring_) model_->unpin_block(b);
}
bool BlockRing::ready() {
  if (tried_) return !ring_.empty();
  tried_ = true;
  ring_.resize(kRing);
  seq_.assign(kRing, 0);
  for (int i = 0; i < kRing; i++)
    if (!model_->pin_block(ring_[i])) {  // no page-locked memory to be had: the copying path
      for (int j = 0; j < i; j++) model_->unpin_block(ring_[j]);
      ring_.clear();
      return false;
    }
  return true;
}
CsrBlock &BlockRing::acquire() {
  while (model_->blocks_pulled() < seq_[slot_]) std::this_thread::yield();
  return ring_[slot_];
}

// One pass over a CSR file image: training visits the rows in a seeded shuffle, block by block;
// evaluation in file order.  Mean of loss(y, logit) over all rows.
double FtrlOffline::csr_epoch(const CsrData &d, bool train) {
  const size_t total = d.n_rows();
  if (total == 0) return 0.0;
  std::vector<int> indices;
  if (train) {
    indices.resize(total);
    std::iota(indices.begin(), indices.end(), 0);
    std::shuffle(indices.begin(), indices.end(), std::mt19937_64{seed_ + (++epoch_no_)});
  }
  double total_loss = 0.0;
  CsrBlock blk;
  if (!ring_) ring_ = std::make_unique<BlockRing>(model_ptr.get());
  const bool ring = ring_->ready();
  size_t pos = 0;
  while (pos < total) {
    const size_t rows = std::min<size_t>(train ? sched_.next_block_rows() : sched_.max_block_rows(), total - pos);
    // both passes are pipelined: this block is uploaded (and, training, grouped) while the previous
    // ones run and the next one is gathered here -- straight into page-locked memory when there is a
    // ring entry it fits (the device then pulls it from there; the entry is reused once it has)
    const size_t nnz = train ? d.gather_nnz(indices.data() + pos, rows)
                             : static_cast<size_t>(d.row_ptr[pos + rows] - d.row_ptr[pos]);
    if (ring && rows <= ring_->row_capacity() && nnz <= ring_->nnz_capacity()) {
      CsrBlock &rb = ring_->acquire();
      if (train) d.gather(indices.data() + pos, rows, rb, n_threads); else d.slice(pos, pos + rows, rb);
      ring_->handed_over(train ? model_ptr->train_block_pinned(rb) : model_ptr->predict_block_async(rb, true));
    } else if (train) {
      d.gather(indices.data() + pos, rows, blk, n_threads);
      model_ptr->train_block_async(blk);
    } else {
      d.slice(pos, pos + rows, blk);
      model_ptr->predict_block_async(blk, false);
    }
    if (train) sched_.consumed(static_cast<int>(rows));
    pos += rows;
  }
  total_loss = train ? model_ptr->train_flush() : model_ptr->eval_flush();
  return total_loss / static_cast<double>(total);
}

void FtrlOffline::train() {
  for (int i = 1; i <= n_epochs; i++) {
    const auto t0 = timer::now();
    const double train_loss = csr_epoch(train_csr_, true);
    std::printf("epoch %d train time: %.4lfs, train loss: %.4lf\n", i, seconds_since(t0), train_loss);
    if (has_eval_) evaluate(i);
  }
}

void FtrlOffline::evaluate(int epoch) {
  const auto t0 = timer::now();
  const double eval_loss = csr_epoch(eval_csr_, false);
  std::printf("epoch %d eval time: %.4lfs, eval loss: %.4lf\n", epoch, seconds_since(t0), eval_loss);
}

// One pass over `samples`: training visits them in a seeded shuffle (the reference shuffles from
// std::random_device, ftrl_offline.cpp:67-71), block by block; evaluation in order.  Returns the
// mean of loss(y, logit) over all rows, as ftrl_offline.cpp:101-102.
double FtrlOffline::one_epoch(std::vector<Sample> &samples, bool train, bool /*use_pool*/) {
  const size_t total = samples.size();
  if (total == 0) return 0.0;
  std::vector<int> indices(total);
  std::iota(indices.begin(), indices.end(), 0);
  if (train) std::shuffle(indices.begin(), indices.end(), std::mt19937_64{seed_ + (++epoch_no_)});
  double total_loss = 0.0;
  CsrBlock blk;
  size_t pos = 0;
  while (pos < total) {
    const size_t rows = std::min<size_t>(train ? sched_.next_block_rows() : sched_.max_block_rows(), total - pos);
    blk.clear();
    for (size_t r = 0; r < rows; r++) blk.push(samples[indices[pos + r]]);
    if (train) model_ptr->train_block_async(blk); else total_loss += model_ptr->predict_block(blk, false);
    if (train) sched_.consumed(static_cast<int>(rows));
    pos += rows;
  }
  if (train) total_loss = model_ptr->train_flush();
  return total_loss / static_cast<double>(total);
}

// ---------------- online: streaming from the file, file order ----------------

FtrlOnline::FtrlOnline(const config_options &opt)
    : model_ptr(make_model(opt)), n_epochs(opt.epoch), cmd_(opt.cmd),
      sched_(opt.batch_size, opt.batch_ramp < 0 ? ffm_engine_default_batch_ramp(opt.w_alpha) : opt.batch_ramp) {
  if (!cmd_) {
    train_stream_ = std::make_unique<CsrStream>(opt.train_path, opt.file_type, opt.thread_num);
    if (!opt.eval_path.empty()) {
      evaluator = std::make_unique<Evaluator>(opt);
      evaluator->load_trained_model(model_ptr);
    }
  }
}

// One pass over the training file in FILE ORDER (the reference's deterministic order at one
// thread): worker threads parse chunks of <= 20 000 lines ahead (parse time is inside
// the timed region, as in the reference's online mode), the rows are cut into blocks by the
// block-size ramp, gathered in page-locked ring entries and handed to the engine, which uploads
// and groups block t+2 while block t trains.
void FtrlOnline::run_train_file() {
  if (!ring_) ring_ = std::make_unique<BlockRing>(model_ptr.get());
  const bool ring = ring_->ready();
  CsrBlock blk;
  unsigned long long rows = 0, next_report = 1000000;
  for (;;) {
    const size_t want = static_cast<size_t>(sched_.next_block_rows());
    size_t got;
    if (ring) {
      CsrBlock &rb = ring_->acquire();
      got = train_stream_->next(std::min(want, ring_->row_capacity()), rb, ring_->nnz_capacity(), true);
      if (got == 0) break;
      ring_->handed_over(model_ptr->train_block_pinned(rb));
    } else {
      got = train_stream_->next(want, blk);
      if (got == 0) break;
      model_ptr->train_block_async(blk);
    }
    sched_.consumed(static_cast<int>(got));
    rows += got;
    if (rows >= next_report) {  // pc_task.cpp:47-49
      std::printf("%llu lines finished...\n", next_report);
      next_report += 1000000;
    }
  }
  loss_sum_ = model_ptr->train_flush();
  loss_rows_ = rows;
  train_stream_->rewind();
}

double FtrlOnline::get_loss() {
  const double r = loss_rows_ ? loss_sum_ / static_cast<double>(loss_rows_) : 0.0;
  loss_sum_ = 0.0;
  loss_rows_ = 0;
  return r;
}

void FtrlOnline::train() {
  if (cmd_) return;  // stdin mode is a TODO stub in the reference too (ftrl_online.cpp:55-57)
  for (int i = 1; i <= n_epochs; i++) {
    const auto t0 = timer::now();
    run_train_file();
    const double train_loss = get_loss();
    std::printf("epoch %d train time: %.4lfs, train loss: %.4lf\n", i, seconds_since(t0), train_loss);
    if (evaluator) evaluate(i);
  }
}

void FtrlOnline::evaluate(int epoch) {
  if (!evaluator) return;
  const auto t0 = timer::now();
  evaluator->run();
  const double eval_loss = evaluator->get_loss();
  std::printf("epoch %d eval time: %.4lfs, eval loss: %.4lf\n", epoch, seconds_since(t0), eval_loss);
}

// ---------------- Evaluator ----------------

Evaluator::Evaluator(const config_options &opt)
    : stream_(std::make_unique<CsrStream>(opt.eval_path, opt.file_type, opt.thread_num)),
      batch_(std::max(1, opt.batch_size)) {}
Evaluator::~Evaluator() = default;

void Evaluator::load_trained_model(std::shared_ptr<FtrlModel> &train_model) {  // evaluate.cpp:35-37
  eval_model = train_model;
  ring_ = std::make_unique<BlockRing>(eval_model.get());
}

void Evaluator::run() {
  if (!eval_model) throw std::logic_error("Evaluator::run before load_trained_model");
  const bool ring = ring_->ready();
  CsrBlock blk;
  unsigned long long rows = 0;
  for (;;) {
    size_t got;
    if (ring) {
      CsrBlock &rb = ring_->acquire();
      got = stream_->next(std::min<size_t>(batch_, ring_->row_capacity()), rb, ring_->nnz_capacity(), true);
      if (got == 0) break;
      ring_->handed_over(eval_model->predict_block_async(rb, true));
    } else {
      got = stream_->next(static_cast<size_t>(batch_), blk);
      if (got == 0) break;
      eval_model->predict_block_async(blk, false);
    }
    rows += got;
  }
  loss_sum_ = eval_model->eval_flush();
  rows_ = rows;
  stream_->rewind();
}

double Evaluator::get_loss() {
  const double r = rows_ ? loss_sum_ / static_cast<double>(rows_) : 0.0;
  loss_sum_ = 0.0;
  rows_ = 0;
  return r;
}

}  // namespace ftrl
