// csr_stream.h -- a text file as a stream of CSR blocks in FILE ORDER, parsed by a pool of threads.
//
// What the reference's online mode does with PcTask (src/concurrent/pc_task.cpp:22-80): one
// producer reads <= 20 000 lines at a time, consumers parse them (src/task/ftrl_online.cpp:70-80).
// Here the file is mmap-ed, a scanner cuts it into chunks of <= kChunkLines whole lines, n_threads
// workers parse chunks ahead of the consumer with the from_chars scanner of csr_reader.cpp (no
// std::string per line, no std::vector<std::tuple> per row), and next() hands out the rows strictly
// in file order -- block boundaries (the block-size ramp) are the consumer's, not the chunks'.
#pragma once
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "csr_reader.h"
#include "types.h"

namespace ftrl {

class CsrStream {
 public:
  static constexpr size_t kChunkLines = 20000;  // pc_task.h:30 (buf_size)
  CsrStream(const std::string &path, const std::string &file_type, int n_threads);
  ~CsrStream();
  CsrStream(const CsrStream &) = delete;
  CsrStream &operator=(const CsrStream &) = delete;
  // The next `want` rows (fewer at the end of the file) into `out` (replaced); 0 at the end.
  // max_nnz: stop early rather than exceed that many entries (a block must fit one engine call).
  // fixed_capacity: `out` must not grow (a page-locked ring entry, registered at its size): a single
  // row with more entries than max_nnz is then refused (std::length_error) instead of going out alone.
  size_t next(size_t want, CsrBlock &out, size_t max_nnz = static_cast<size_t>(-1), bool fixed_capacity = false);
  // The next parsed chunk in place, no copy (the whole-file loader): valid until release_chunk().
  // Not to be mixed with a partly consumed chunk of next().  false at the end of the file.
  bool acquire_chunk(const CsrPart **part, const std::vector<int64_t> **row_ptr);
  void release_chunk();
  void rewind();  // back to the first line (the next epoch)
  unsigned long long lines_delivered() const { return delivered_; }

 private:
  struct Chunk {
    CsrPart part;
    std::vector<int64_t> row_ptr;  // prefix sums of part.nnz
    size_t id = 0;
    bool ready = false;
  };
  void worker();
  bool claim(size_t *id, const char **b, const char **e);  // next chunk's byte range (locked scan)
  void start_workers();
  void stop_workers();

  const char *base_ = nullptr;
  size_t len_ = 0;
  int fd_ = -1;
  bool has_field_ = false;
  int n_threads_ = 1;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_ready_;
  std::vector<Chunk> ring_;       // chunk id -> ring_[id % ring_.size()]
  size_t scan_pos_ = 0;           // first byte not yet assigned to a chunk
  size_t next_id_ = 0;            // chunks assigned so far
  size_t consume_id_ = 0;         // chunk the consumer reads from
  size_t consume_row_ = 0;        // rows of it already delivered
  bool stop_ = false;
  std::string error_;             // a worker's parse error (std::out_of_range text), rethrown by next()
  unsigned long long delivered_ = 0;
};

}  // namespace ftrl
