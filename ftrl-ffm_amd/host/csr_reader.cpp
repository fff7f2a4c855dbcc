#include "csr_reader.h"

#include "csr_stream.h"

#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <charconv>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <iostream>
#include <stdexcept>

namespace ftrl {
namespace {

using Part = CsrPart;

[[noreturn]] void bad_line(const char *b, const char *e) {
  const std::string line(b, e);
  std::cout << "wrong input: " << line << std::endl;
  throw std::out_of_range(line);
}

inline int to_int(const char *b, const char *e, const char *lb, const char *le) {
  int v = 0;
  if (b < e && *b == '+') b++;
  const auto r = std::from_chars(b, e, v);
  if (r.ec != std::errc() || r.ptr == b) bad_line(lb, le);
  return v;
}
inline float to_float(const char *b, const char *e, const char *lb, const char *le) {
  float v = 0.0f;
  if (b < e && *b == '+') b++;
  const auto r = std::from_chars(b, e, v);
  if (r.ec == std::errc::result_out_of_range) return std::strtof(std::string(b, e).c_str(), nullptr);
  if (r.ec != std::errc() || r.ptr == b) bad_line(lb, le);
  return v;
}

// Fast path of one "field:feat:value" / "feat:value" token starting at q (a non-blank): plain
// unsigned decimal ids of at most 9 digits and a plain decimal value ([-]digits[.digits], no
// exponent) of at most 7 significant digits and 10 fraction digits, ended by a blank or the line's
// end.  For those the value is EXACTLY what from_chars returns: the digit string D < 10^7 < 2^24
// and 10^k (k <= 10) are both exact floats, so one IEEE division rounds the decimal correctly
// (Clinger's fast path).  Anything else returns false with nothing consumed and the caller takes
// the general path (from_chars, the reference's error behaviour).
inline bool fast_uint(const char *&c, const char *stop, int &v) {
  const char *b = c;
  unsigned x = 0;
  while (c < stop && static_cast<unsigned>(*c - '0') <= 9u) x = x * 10u + static_cast<unsigned>(*c++ - '0');
  if (c == b || c - b > 9) return false;
  v = static_cast<int>(x);
  return true;
}
inline bool fast_token(const char *&q, const char *stop, bool has_field, int &fld, int &ft, float &v) {
  static const float kPow10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
  const char *c = q;
  fld = 0;
  if (has_field) {
    if (!fast_uint(c, stop, fld) || c >= stop || *c != ':') return false;
    c++;
  }
  if (!fast_uint(c, stop, ft) || c >= stop || *c != ':') return false;
  c++;
  bool neg = false;
  if (c < stop && *c == '-') { neg = true; c++; }
  unsigned digits = 0;  // the significand as an integer
  int n_sig = 0, n_frac = 0, n_any = 0;
  while (c < stop && static_cast<unsigned>(*c - '0') <= 9u) {
    digits = digits * 10u + static_cast<unsigned>(*c - '0');
    n_sig += (n_sig > 0 || *c != '0') ? 1 : 0;
    n_any++;
    c++;
    if (n_sig > 7) return false;
  }
  if (c < stop && *c == '.') {
    c++;
    while (c < stop && static_cast<unsigned>(*c - '0') <= 9u) {
      digits = digits * 10u + static_cast<unsigned>(*c - '0');
      n_sig += (n_sig > 0 || *c != '0') ? 1 : 0;
      n_any++;
      n_frac++;
      c++;
      if (n_sig > 7 || n_frac > 10) return false;
    }
  }
  if (n_any == 0 || (c < stop && *c != ' ')) return false;
  const float mag = n_frac ? static_cast<float>(digits) / kPow10[n_frac] : static_cast<float>(digits);
  v = neg ? -mag : mag;
  q = c;
  return true;
}

void parse_range(const char *p, const char *end, bool has_field, Part &out) {
  while (p < end) {
    const char *le = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    if (!le) le = end;
    const char *lb = p;
    const char *q = lb, *stop = le;
    if (stop > lb && stop[-1] == '\r') stop--;
    while (q < stop && *q == ' ') q++;
    if (q < stop) {
      const char *t = q;
      while (t < stop && *t != ' ') t++;
      out.label.push_back(to_int(q, t, lb, le) > 0 ? 1 : 0);
      int n = 0;
      q = t;
      while (true) {
        while (q < stop && *q == ' ') q++;
        if (q >= stop) break;
        {
          int ffld, fft;
          float fv;
          if (fast_token(q, stop, has_field, ffld, fft, fv)) {
            if (fv != 0.0f) {
              out.field.push_back(ffld);
              out.feat.push_back(fft);
              out.val.push_back(fv);
              n++;
            }
            continue;
          }
        }
        t = q;
        while (t < stop && *t != ' ') t++;
        const char *c1 = q;
        while (c1 < t && *c1 != ':') c1++;
        if (c1 >= t) bad_line(lb, le);
        int fld = 0, ft;
        const char *vb;
        if (has_field) {
          fld = to_int(q, c1, lb, le);
          const char *c2 = c1 + 1;
          while (c2 < t && *c2 != ':') c2++;
          if (c2 >= t) bad_line(lb, le);
          ft = to_int(c1 + 1, c2, lb, le);
          vb = c2 + 1;
        } else {
          ft = to_int(q, c1, lb, le);
          vb = c1 + 1;
        }
        if (vb >= t) bad_line(lb, le);
        const float v = to_float(vb, t, lb, le);
        if (v != 0.0f) {
          out.field.push_back(fld);
          out.feat.push_back(ft);
          out.val.push_back(v);
          n++;
        }
        q = t;
      }
      out.nnz.push_back(n);
    }
    p = le + 1;
  }
}

}  // namespace

void parse_csr_range(const char *begin, const char *end, bool has_field, CsrPart &out) {
  parse_range(begin, end, has_field, out);
}

void CsrData::slice(size_t r0, size_t r1, CsrBlock &out) const {
  out.clear();
  const int64_t b = row_ptr[r0], e = row_ptr[r1];
  out.field.assign(field.begin() + b, field.begin() + e);
  out.feat.assign(feat.begin() + b, feat.begin() + e);
  out.val.assign(val.begin() + b, val.begin() + e);
  out.label.assign(label.begin() + r0, label.begin() + r1);
  out.row_ptr.resize(r1 - r0 + 1);
  for (size_t r = r0; r <= r1; r++) out.row_ptr[r - r0] = static_cast<int32_t>(row_ptr[r] - b);
}

size_t CsrData::gather_nnz(const int *idx, size_t n) const {
  size_t total = 0;
  for (size_t j = 0; j < n; j++) total += static_cast<size_t>(row_ptr[idx[j] + 1] - row_ptr[idx[j]]);
  return total;
}

void CsrData::gather(const int *idx, size_t n, CsrBlock &out, int n_threads) const {
  out.row_ptr.resize(n + 1);
  out.label.resize(n);
  out.row_ptr[0] = 0;
  int32_t total = 0;
  for (size_t j = 0; j < n; j++) {
    total += static_cast<int32_t>(row_ptr[idx[j] + 1] - row_ptr[idx[j]]);
    out.row_ptr[j + 1] = total;
  }
  out.field.resize(static_cast<size_t>(total));
  out.feat.resize(static_cast<size_t>(total));
  out.val.resize(static_cast<size_t>(total));
  int32_t *of = out.field.data(), *oi = out.feat.data(), *ol = out.label.data();
  float *ov = out.val.data();
  const int32_t *orp = out.row_ptr.data();
#pragma omp parallel for schedule(static) num_threads(n_threads) if (n_threads > 1 && n >= 1024)
  for (size_t j = 0; j < n; j++) {
    const int64_t b = row_ptr[idx[j]];
    const size_t len = static_cast<size_t>(row_ptr[idx[j] + 1] - b);
    std::memcpy(of + orp[j], field.data() + b, 4 * len);
    std::memcpy(oi + orp[j], feat.data() + b, 4 * len);
    std::memcpy(ov + orp[j], val.data() + b, 4 * len);
    ol[j] = label[idx[j]];
  }
}

// The whole file as one CSR: the chunked stream reader (csr_stream.h: chunks of <= 20 000 lines parsed
// ahead by n_threads workers into buffers that are reused) handed over in file order and appended.
// (Round 3 cut the file into n_threads byte ranges, parsed each into freshly reserved vectors and
// concatenated them: 3.1 - 3.7 M rows/s at 8 threads whatever the concatenation did -- gigabytes of
// first-touch page faults from eight threads of one address space -- against the stream's 16.6 M.)
CsrData load_csr(const std::string &path, const std::string &file_type, int n_threads) {
  CsrData out;
  struct stat st {};
  if (stat(path.c_str(), &st) != 0) {
    std::cerr << "fail to open " << path << std::endl;
    std::exit(EXIT_FAILURE);
  }
  if (st.st_size == 0) return out;
  CsrStream stream(path, file_type, n_threads);
  // a token is >= 4 bytes (libsvm) / 6 (libffm), a row >= 2 ("1\n"): upper bounds, reserved once
  // (untouched reserve costs nothing where memory is overcommitted; the arrays grow into it page by
  // page).  Under strict overcommit or a virtual-memory limit the reservation may be refused: the
  // arrays then simply grow geometrically as they are filled.
  // Each array by itself, so that one refusal does not cost the others their reservation; the row
  // arrays at one row per 8 bytes (a libffm row of one entry): shorter rows than that make them grow
  // once or twice, where bytes / 2 asked for 4.5 times the file size in address space for them alone
  // (ADVICE r05).
  const size_t bytes = static_cast<size_t>(st.st_size), tok = file_type == "libffm" ? 6 : 4;
  auto try_reserve = [](auto &v, size_t n) {
    try { v.reserve(n); } catch (const std::bad_alloc &) {}
  };
  try_reserve(out.field, bytes / tok);
  try_reserve(out.feat, bytes / tok);
  try_reserve(out.val, bytes / tok);
  try_reserve(out.row_ptr, bytes / 8 + 2);
  try_reserve(out.label, bytes / 8 + 1);
  // first touch of ~1 GB per million rows is what this loop would otherwise spend its time on: ask for
  // huge pages (512 times fewer faults where transparent huge pages are on "madvise" or "always")
  auto huge = [](void *p, size_t n) {
    const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + n) & ~uintptr_t(4095);
    if (e > a) (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
  };
  huge(out.field.data(), 4 * out.field.capacity());
  huge(out.feat.data(), 4 * out.feat.capacity());
  huge(out.val.data(), 4 * out.val.capacity());
  // The consumer is one thread copying ~470 B per row into untouched memory while the stream's workers
  // parse ahead: the chunk is read in place (no block in between) and its three entry arrays are
  // copied by three helpers side by side with the row arrays.
  struct Copy { void *dst; const void *src; size_t n; };
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  Copy jobs[3] = {};
  unsigned long long gen = 0;
  int pending = 0;
  bool quit = false;
  std::vector<std::thread> helpers;
  for (int h = 0; h < 3; h++)
    helpers.emplace_back([&, h] {
      unsigned long long seen = 0;
      for (;;) {
        Copy j;
        {
          std::unique_lock<std::mutex> lock(mu);
          cv_go.wait(lock, [&] { return quit || gen != seen; });
          if (quit) return;
          seen = gen;
          j = jobs[h];
        }
        if (j.n) std::memcpy(j.dst, j.src, j.n);
        {
          std::lock_guard<std::mutex> lock(mu);
          pending--;
        }
        cv_done.notify_one();
      }
    });
  auto stop_helpers = [&] {
    {
      std::lock_guard<std::mutex> lock(mu);
      quit = true;
    }
    cv_go.notify_all();
    for (auto &t : helpers) t.join();
  };
  const CsrPart *part = nullptr;
  const std::vector<int64_t> *rp = nullptr;
  try {
    while (stream.acquire_chunk(&part, &rp)) {
      const size_t got = part->nnz.size(), ne = part->feat.size();
      const size_t e0 = out.feat.size(), r0 = out.label.size();
      out.field.resize(e0 + ne); out.feat.resize(e0 + ne); out.val.resize(e0 + ne);
      out.label.resize(r0 + got); out.row_ptr.resize(r0 + got + 1);
      {
        std::lock_guard<std::mutex> lock(mu);
        jobs[0] = {out.field.data() + e0, part->field.data(), 4 * ne};
        jobs[1] = {out.feat.data() + e0, part->feat.data(), 4 * ne};
        jobs[2] = {out.val.data() + e0, part->val.data(), 4 * ne};
        pending = 3;
        gen++;
      }
      cv_go.notify_all();
      if (got) std::memcpy(out.label.data() + r0, part->label.data(), 4 * got);
      for (size_t r = 0; r < got; r++) out.row_ptr[r0 + r + 1] = static_cast<int64_t>(e0) + (*rp)[r + 1];
      {
        std::unique_lock<std::mutex> lock(mu);
        cv_done.wait(lock, [&] { return pending == 0; });
      }
      stream.release_chunk();
    }
  } catch (...) {
    stop_helpers();
    throw;
  }
  stop_helpers();
  // (no shrink_to_fit: it would copy the arrays once more; the untouched tail of a reserve costs no memory)
  return out;
}

}  // namespace ftrl
