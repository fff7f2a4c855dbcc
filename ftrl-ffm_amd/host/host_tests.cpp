// host_tests.cpp -- the reference's own unit/integration tests for this path, restated against the
// host mirror (reference tests/test_data.cpp:9-18, tests/test_model.cpp:22-49,
// tests/test_task.cpp:25-43, tests/test_utils.cpp:40-43), plus flag parsing and parser errors.
//   host_tests cpu   -> everything that needs no GPU (reader, parsers, flags, loss)
//   host_tests gpu   -> model shapes, remove_out_range, weight round trip, online/offline tasks
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>

#include "cmd_option.h"
#include "csr_reader.h"
#include "csr_stream.h"
#include "ftrl_model.h"
#include "persist.h"
#include "reader.h"
#include "trainer.h"

static int g_failed = 0, g_checked = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    g_checked++;                                                                 \
    if (!(cond)) { g_failed++; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

// the 10-row libffm fixture of the reference's tests (tests/common.h:14-24): data, not code
static const char *kSamples =
    "0 0:1:1 1:13:1 2:21:1 3:31:1\n1 0:4:1 1:11:1 2:23:1 3:32:1\n1 0:2:1 1:13:1 2:25:1 3:34:1\n"
    "0 0:1:1 1:14:1 2:21:1 3:32:1\n0 0:2:1 1:15:1 2:22:1 3:34:1\n1 0:4:1 1:11:1 2:21:1 3:35:1\n"
    "1 0:5:1 1:12:1 2:23:1 3:31:1\n1 0:5:1 1:12:1 2:25:1 3:38:1\n0 0:2:1 1:11:1 2:24:1 3:37:1\n"
    "1 0:1:1 1:15:1 2:22:1 3:35:1";
static const char *kPath = "./host_test_file.txt";
static void write_fixture() { std::ofstream(kPath) << kSamples; }

static void test_reader_and_parsers() {
  write_fixture();
  ftrl::Reader reader("libffm");
  reader.load_from_file(kPath, 4);
  CHECK(reader.get_size() == 10);
  CHECK(reader.data[0].y == 0);
  CHECK(reader.data[0].x[0] == std::make_tuple(0, 1, 1.0f));
  CHECK(reader.data[0].x[3] == std::make_tuple(3, 31, 1.0f));
  CHECK(reader.data[9].y == 1 && reader.data[9].x.size() == 4);
  ftrl::LibsvmParser sp;
  Sample s;
  sp.parse("-1 3:0.5 7:0 9:2", s);  // label <= 0 -> 0; zero value dropped; field 0
  CHECK(s.y == 0 && s.x.size() == 2 && s.x[1] == std::make_tuple(0, 9, 2.0f));
  sp.parse("  2 5:1  ", s);
  CHECK(s.y == 1 && s.x.size() == 1);
  ftrl::FFMParser fp;
  bool threw = false;
  try { fp.parse("1 0:5", s); } catch (const std::out_of_range &) { threw = true; }
  CHECK(threw);  // malformed token -> std::out_of_range (parser.cpp:24-27)
  threw = false;
  try { sp.parse("1 12", s); } catch (const std::out_of_range &) { threw = true; }
  CHECK(threw);
  CHECK(detect_file_type(kPath) == "libffm");
  std::remove(kPath);
}

static void test_csr_reader_matches_line_parsers() {
  // a ragged libffm file: zero values, spaces, negative label, multi-valued fields
  std::string text;
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int r = 0; r < 5000; r++) {
    text += std::to_string(static_cast<int>(rnd() % 3) - 1);
    const int nnz = rnd() % 12;
    for (int j = 0; j < nnz; j++) {
      char tok[64];
      const int v = rnd() % 5;
      std::snprintf(tok, sizeof tok, " %u:%u:%s", rnd() % 8, rnd() % 10000,
                    v == 0 ? "0" : v == 1 ? "1" : v == 2 ? "0.6182" : v == 3 ? "2.5e-3" : "17");
      text += tok;
      if (rnd() % 7 == 0) text += " ";
    }
    text += "\n";
  }
  { std::ofstream(kPath) << text; }
  ftrl::Reader reader("libffm");
  reader.load_from_file(kPath, 3);
  for (int threads : {1, 4}) {
    const ftrl::CsrData d = ftrl::load_csr(kPath, "libffm", threads);
    CHECK(d.n_rows() == reader.get_size());
    bool same = d.n_rows() == reader.get_size();
    for (size_t r = 0; same && r < d.n_rows(); r++) {
      const auto &x = reader.data[r].x;
      same = same && d.label[r] == reader.data[r].y &&
             static_cast<size_t>(d.row_ptr[r + 1] - d.row_ptr[r]) == x.size();
      for (size_t j = 0; same && j < x.size(); j++) {
        const int64_t p = d.row_ptr[r] + static_cast<int64_t>(j);
        same = d.field[p] == std::get<0>(x[j]) && d.feat[p] == std::get<1>(x[j]) &&
               d.val[p] == std::get<2>(x[j]);
      }
    }
    CHECK(same);
    CsrBlock b1, b2;
    d.slice(10, 20, b1);
    const int idx[3] = {19, 10, 4000};
    d.gather(idx, 3, b2);
    CHECK(b1.n_rows() == 10 && b2.n_rows() == 3 && b2.label[1] == d.label[10]);
  }
  bool threw = false;
  { std::ofstream(kPath) << "1 0:5:1\n0 3:7\n"; }
  try { ftrl::load_csr(kPath, "libffm", 1); } catch (const std::out_of_range &) { threw = true; }
  CHECK(threw);
  std::remove(kPath);
}

static void test_csr_fast_tokens_are_strtof() {
  // the reader's fast path (ids by digit loop, short decimals by one float division) against
  // strtof on every spelling it accepts and on those it must hand to the general path
  std::vector<std::string> vals = {"1", "0", "-1", "17", "0.6182", "0.551", "-0.25", "1.", "007.5000", "0.1", "0.3",
                                   "9999999", "0.9999999", "16777215", "16777217", "0.0000001", "0.00000000015",
                                   "123.4567", "1234567.8", "0.1234567", "0.12345678", "1e-3", "2.5e-3", "+3.5",
                                   "3.4028235e38", "1e-45", "0.0000000001", "4.7", "33.333334", "8.25", ".5", "-.5"};
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int i = 0; i < 20000; i++) {
    char buf[64];
    const int decimals = rnd() % 9;
    const double mag = static_cast<double>(rnd() % 10000000) / std::pow(10.0, rnd() % 8);
    std::snprintf(buf, sizeof buf, "%s%.*f", rnd() % 5 == 0 ? "-" : "", decimals, mag);
    vals.push_back(buf);
  }
  std::string text;
  for (size_t i = 0; i < vals.size(); i++)
    text += "1 " + std::to_string(i % 39) + ":" + std::to_string(i) + ":" + vals[i] + " 3:000123:1\n";
  ftrl::CsrPart part;
  ftrl::parse_csr_range(text.data(), text.data() + text.size(), true, part);
  CHECK(part.label.size() == vals.size());
  bool same = part.label.size() == vals.size();
  size_t p = 0;
  for (size_t i = 0; same && i < vals.size(); i++) {
    const float want = std::strtof(vals[i].c_str(), nullptr);
    if (want != 0.0f) {
      same = p < part.val.size() && part.feat[p] == static_cast<int>(i) && part.field[p] == static_cast<int>(i % 39) &&
             std::memcmp(&part.val[p], &want, 4) == 0;
      if (!same) std::printf("value %s: got %.9g want %.9g\n", vals[i].c_str(), p < part.val.size() ? part.val[p] : -1.0f, want);
      p++;
    }
    same = same && p < part.val.size() && part.feat[p] == 123 && part.field[p] == 3 && part.val[p] == 1.0f;
    p++;
  }
  CHECK(same);
  CHECK(p == part.val.size());
}

static void test_csr_stream_is_the_file_in_order() {
  // 45 000 ragged rows (more than two 20 000-line chunks), blank lines, one giant row
  std::string text;
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int r = 0; r < 45000; r++) {
    if (r % 9973 == 0) text += "\n";
    text += std::to_string(static_cast<int>(rnd() % 2));
    const int nnz = r == 30000 ? 300 : rnd() % 10;
    for (int j = 0; j < nnz; j++) {
      char tok[64];
      std::snprintf(tok, sizeof tok, " %u:%u:%s", rnd() % 8, rnd() % 10000, rnd() % 4 ? "1" : "0.25");
      text += tok;
    }
    text += "\n";
  }
  { std::ofstream(kPath) << text; }
  const ftrl::CsrData d = ftrl::load_csr(kPath, "libffm", 2);
  for (int threads : {1, 3}) {
    ftrl::CsrStream st(kPath, "libffm", threads);
    for (int pass = 0; pass < 2; pass++) {  // second pass: after rewind()
      size_t row = 0;
      bool same = true;
      CsrBlock b;
      unsigned step = 0;
      for (;;) {
        const size_t want = 1 + (step++ * 7919u) % 5000;  // ragged block sizes across chunk edges
        const size_t got = st.next(want, b, 20000);
        if (got == 0) break;
        same = same && static_cast<size_t>(b.n_rows()) == got && got <= want;
        same = same && (b.feat.size() <= 20000 || got == 1);
        for (size_t r = 0; same && r < got; r++, row++) {
          const int64_t p0 = d.row_ptr[row], len = d.row_ptr[row + 1] - p0;
          same = b.label[r] == d.label[row] && b.row_ptr[r + 1] - b.row_ptr[r] == len;
          for (int64_t j = 0; same && j < len; j++)
            same = b.field[b.row_ptr[r] + j] == d.field[p0 + j] && b.feat[b.row_ptr[r] + j] == d.feat[p0 + j] &&
                   b.val[b.row_ptr[r] + j] == d.val[p0 + j];
        }
        if (!same) break;
      }
      CHECK(same);
      CHECK(row == d.n_rows());
      st.rewind();
    }
  }
  { std::ofstream(kPath) << "1 0:5:1\n0 3:7\n"; }
  bool threw = false;
  try {
    ftrl::CsrStream bad(kPath, "libffm", 2);
    CsrBlock b;
    while (bad.next(10, b)) {}
  } catch (const std::out_of_range &) { threw = true; }
  CHECK(threw);
  std::remove(kPath);
}

static void test_flags() {
  write_fixture();
  config_options d;
  CHECK(d.model_type == "FFM" && d.n_fields == 8 && d.n_feats == 10000 && d.n_factors == 16);
  CHECK(d.w_alpha == 1e-4f && d.w_beta == 1.0f && d.w_l1 == 0.1f && d.w_l2 == 5.0f && d.online);
  const char *argv[] = {"main", "--train_data", kPath, "--model_type", "ffm", "--n_fields", "4",
                        "--n_feats", "50", "--n_factors", "4", "--n_epochs", "2", "--online",
                        "false", "--w_alpha", "0.1", "--batch_size", "8"};
  config_options o;
  o.parse_option(19, const_cast<char **>(argv));
  CHECK(o.model_type == "FFM" && o.file_type == "libffm" && o.n_fields == 4 && o.epoch == 2);
  CHECK(!o.online && o.w_alpha == 0.1f && o.batch_size == 8);
  const char *bad[] = {"main", "--epoch", "3"};  // the reference accepts only --n_epochs
  bool threw = false;
  try { config_options b; b.parse_option(3, const_cast<char **>(bad)); }
  catch (const std::invalid_argument &) { threw = true; }
  CHECK(threw);
  std::remove(kPath);
}

static void test_loss_known_answers() {  // tests/test_utils.cpp:40-43
  CHECK(std::fabs(ftrl::loss(1, 2) - 0.1269) < 1e-4);
  CHECK(std::fabs(ftrl::loss(0, 1) - 1.3133) < 1e-4);
  ftrl::BlockScheduler s(256, 32);
  CHECK(s.next_block_rows() == 1);
  s.consumed(64);
  CHECK(s.next_block_rows() == 2);
  s.consumed(1 << 20);
  CHECK(s.next_block_rows() == 256);
}

static config_options small_args(const char *type) {
  config_options a;
  a.n_fields = 4; a.n_feats = 50; a.n_factors = 4;
  a.model_type = type;
  a.batch_size = 16;
  return a;
}

static void test_models_gpu() {  // tests/test_model.cpp:22-49
  {
    ftrl::LR model{small_args("LR")};
    CHECK(model.model_type == ModelType::LR);
    CHECK(model.lin_w.size() == 50);
    feat_vec invalid = {{1, -1, 3}, {1, 0, 1}, {1, 100, 0}};
    model.remove_out_range(invalid);
    CHECK(invalid.size() == 1);
  }
  {
    ftrl::FM model{small_args("FM")};
    CHECK(model.model_type == ModelType::FM);
    CHECK(model.vec_w.size() == 50 && model.vec_w[0].size() == 4);
  }
  {
    ftrl::FFM model{small_args("FFM")};
    CHECK(model.model_type == ModelType::FFM);
    CHECK(model.vec_w[0].size() == 16);
    feat_vec invalid = {{1, -1, 3}, {44, 0, 1}, {1, 100, 0}};
    model.remove_out_range(invalid);
    CHECK(invalid.empty());
    // weights are public, editable, and what predict() uses (save/load round trip analogue)
    feat_vec sample = {{1, 3, 3}, {1, 0, 1}, {1, 2, 0}, {3, 10, 1}, {12, 4, 0}, {111, 1, 0}, {8, 8, 8}};
    feat_vec s1 = sample;
    const float pred = model.predict(s1, false);
    CHECK(s1.size() == 4);  // out-of-range entries erased from the caller's row, as the reference
    auto a2 = small_args("FFM");
    a2.seed = 7;
    ftrl::FFM other{a2};
    feat_vec s2 = sample;
    CHECK(other.predict(s2, false) != pred);
    other.bias = model.bias; other.lin_w = model.lin_w; other.vec_w = model.vec_w;
    other.push_weights();
    feat_vec s3 = sample;
    CHECK(other.predict(s3, false) == pred);
    // one train() from a fresh model returns logit 0 after the lazy refresh zeroes what it touches
    feat_vec row = {{0, 1, 1.0f}, {1, 13, 1.0f}};
    CHECK(model.train(row, 1) == 0.0f);
    bool threw = false;
    auto bad = small_args("XX");
    try { ftrl::make_model(bad); } catch (const std::invalid_argument &) { threw = true; }
    CHECK(threw);
  }
}

static void test_persistence_gpu() {  // tests/test_model.cpp:51-102
  feat_vec sample = {{1, 3, 3}, {1, 0, 1}, {1, 2, 0}, {3, 10, 1}, {12, 4, 0}, {111, 1, 0}, {8, 8, 8}};
  {  // LR save & load compressed
    ftrl::LR model{small_args("LR")};
    feat_vec s1 = sample;
    const float pred = model.predict(s1, false);
    model.save_compressed_model("lr.zst", 10);
    auto a2 = small_args("LR");
    a2.seed = 9;
    ftrl::LR fresh{a2};
    feat_vec s2 = sample;
    CHECK(fresh.predict(s2, false) != pred);
    fresh.load_compressed_model("lr.zst");
    feat_vec s3 = sample;
    CHECK(fresh.predict(s3, false) == pred);
    std::remove("lr.zst");
  }
  {  // FFM save & load, text (approximate) and compressed (exact)
    ftrl::FFM model{small_args("FFM")};
    feat_vec s1 = sample;
    const float pred = model.predict(s1, false);
    model.save_model("ffm.txt");
    model.save_compressed_model("ffm.zst", 10);
    auto a2 = small_args("FFM");
    a2.seed = 9;
    ftrl::FFM t{a2}, z{a2};
    feat_vec s2 = sample;
    CHECK(t.predict(s2, false) != pred);
    t.load_model("ffm.txt");
    feat_vec s3 = sample;
    CHECK(std::fabs(t.predict(s3, false) - pred) <= 1e-4f * std::fabs(pred) + 1e-7f);
    z.load_compressed_model("ffm.zst");
    feat_vec s4 = sample;
    CHECK(z.predict(s4, false) == pred);
    // resumable: weights + accumulators restored -> the next train() steps agree exactly
    feat_vec r1 = {{0, 1, 1.0f}, {1, 13, 1.0f}, {2, 21, 0.5f}}, r2 = r1, r3 = r1, r4 = r1;
    model.train(r1, 1);
    model.save_compressed_model("ffm2.zst", 3);
    model.save_state("ffm2.nz");
    const float next = model.train(r2, 0);
    z.load_compressed_model("ffm2.zst");
    z.load_state("ffm2.nz");
    CHECK(z.train(r3, 0) == next);
    std::remove("ffm.txt"); std::remove("ffm.zst"); std::remove("ffm2.zst"); std::remove("ffm2.nz");
  }
}

static void test_tasks_gpu() {  // tests/test_task.cpp:25-43
  for (int online = 0; online < 2; online++) {
    write_fixture();
    config_options opt = small_args("FFM");
    opt.train_path = kPath;
    opt.eval_path = kPath;
    opt.epoch = 2;
    opt.thread_num = 4;
    opt.file_type = "libffm";
    opt.online = online != 0;
    if (online) { ftrl::FtrlOnline t(opt); t.train(); CHECK(t.has_zero_weights()); }
    else { ftrl::FtrlOffline t(opt); t.train(); CHECK(t.has_zero_weights()); }
    std::remove(kPath);
  }
}

// host_tests convert <in> <out> <n_feats> <row_len>: re-encode a model file (".zst" = one zstd
// frame, anything else = text) with the pure file functions; used to cross-check the formats
// against files written / read by the compiled reference.
static int convert(int argc, char **argv) {
  if (argc < 6) return 2;
  const std::string in = argv[2], out = argv[3];
  const size_t nf = std::stoul(argv[4]), rl = std::stoul(argv[5]);
  auto is_zst = [](const std::string &p) { return p.size() > 4 && p.substr(p.size() - 4) == ".zst"; };
  // streamed through the file classes alone: [bias, lin_w[nf], vec_w[nf][rl]]
  std::vector<float> lin(nf + 1), rows(nf * rl);
  if (is_zst(in)) {
    ftrl::FloatFrameReader r(in);
    if (r.total_floats() != 1 + nf + nf * rl) return 3;
    if (r.read(lin.data(), nf + 1) != nf + 1 || r.read(rows.data(), nf * rl) != nf * rl) return 4;
  } else {
    ftrl::TextModelReader r(in);
    for (auto &v : lin) v = r.scalar();
    r.rows(rows.data(), nf, rl);
  }
  if (is_zst(out)) {
    ftrl::FloatFrameWriter w(out, 1 + nf + nf * rl, 3);
    w.write(lin.data(), nf + 1);
    for (size_t f0 = 0; f0 < nf; f0 += 7) w.write(rows.data() + f0 * rl, std::min<size_t>(7, nf - f0) * rl);  // in pieces
    w.finish();
  } else {
    ftrl::TextModelWriter w(out);
    for (float v : lin) w.scalar(v);
    w.rows(rows.data(), nf, rl);
    w.finish();
  }
  return 0;
}

// host_tests ingest <file> <libffm|libsvm> <threads>: rows/s of the two readers
static int ingest(int argc, char **argv) {
  if (argc < 5) return 2;
  const int threads = std::stoi(argv[4]);
  auto t0 = std::chrono::steady_clock::now();
  const ftrl::CsrData d = ftrl::load_csr(argv[2], argv[3], threads);
  const double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  t0 = std::chrono::steady_clock::now();
  ftrl::Reader reader(argv[3]);
  reader.load_from_file(argv[2], threads);
  const double s2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("rows %zu  csr_reader %.0f rows/s  sample_reader %.0f rows/s  (threads %d)\n",
              d.n_rows(), d.n_rows() / s1, reader.get_size() / s2, threads);
  return d.n_rows() == reader.get_size() ? 0 : 1;
}

// host_tests stream <file> <libffm|libsvm> <threads>: rows/s of the chunked stream reader alone
// (parse only: what the online trainer's worker threads sustain), blocks of 8192 rows
static int stream_rate(int argc, char **argv) {
  if (argc < 5) return 2;
  const int threads = std::stoi(argv[4]);
  ftrl::CsrStream st(argv[2], argv[3], threads);
  CsrBlock b;
  for (int pass = 0; pass < 2; pass++) {  // (the second pass reads the page cache)
    const auto t0 = std::chrono::steady_clock::now();
    size_t rows = 0, got;
    while ((got = st.next(8192, b)) != 0) rows += got;
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("pass %d: rows %zu  csr_stream %.0f rows/s  (threads %d)\n", pass, rows, rows / s, threads);
    st.rewind();
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::strcmp(argv[1], "convert") == 0) return convert(argc, argv);
  if (argc > 1 && std::strcmp(argv[1], "stream") == 0) return stream_rate(argc, argv);
  if (argc > 1 && std::strcmp(argv[1], "ingest") == 0) return ingest(argc, argv);
  const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
  test_reader_and_parsers();
  test_csr_reader_matches_line_parsers();
  test_csr_fast_tokens_are_strtof();
  test_csr_stream_is_the_file_in_order();
  test_flags();
  test_loss_known_answers();
  if (gpu) {
    test_models_gpu();
    test_persistence_gpu();
    test_tasks_gpu();
  }
  std::printf("%s: %d checks, %d failed\n", gpu ? "host tests (cpu+gpu)" : "host tests (cpu)", g_checked, g_failed);
  return g_failed ? 1 : 0;
}
