// Memory-mapped uint16 token stream and batch assembly, free of any tensor library so that it also
// builds as a plain C++ test target under AddressSanitizer / UBSan (tests/native/). dataloader.cpp
// wraps it for Python. Replaces the per-sample numpy slicing loop of the reference loader
// (reference: single-gpu/train.py:233-245).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace dpa {

class TokenStream {
 public:
  explicit TokenStream(const std::string& path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("TokenFile: cannot open " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0) fail("TokenFile: cannot stat " + path);
    bytes_ = (size_t)st.st_size;
    if (bytes_ < 2 || bytes_ % 2 != 0) fail("TokenFile: " + path + " is not a uint16 token stream");
    void* p = mmap(nullptr, bytes_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (p == MAP_FAILED) fail("TokenFile: mmap failed for " + path);
    madvise(p, bytes_, MADV_RANDOM);
    data_ = static_cast<const uint16_t*>(p);
    n_ = (int64_t)(bytes_ / 2);
  }
  TokenStream(const TokenStream&) = delete;
  TokenStream& operator=(const TokenStream&) = delete;
  ~TokenStream() { release(); }

  int64_t size() const { return n_; }
  const uint16_t* data() const { return data_; }

  // B windows of T + 1 tokens starting at starts[i], widened to int64: x[i] = w[:-1], y[i] = w[1:].
  // x and y each hold B * T values. Throws before writing anything if a window leaves the file.
  void gather(const int64_t* starts, int64_t B, int64_t T, int64_t* x, int64_t* y, int64_t threads) const {
    if (B < 0 || T < 1) throw std::invalid_argument("TokenFile: need B >= 0 and T >= 1");
    if (T >= n_) throw std::out_of_range("TokenFile: a window of T + 1 tokens is longer than the file");
    const int64_t last = n_ - T - 1;   // T < n_, so no overflow
    for (int64_t i = 0; i < B; ++i)
      if (starts[i] < 0 || starts[i] > last)
        throw std::out_of_range("window " + std::to_string(starts[i]) + " runs past the end of the token file");
    auto work = [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        const uint16_t* w = data_ + starts[i];
        int64_t* xi = x + i * T;
        int64_t* yi = y + i * T;
        for (int64_t t = 0; t < T; ++t) {
          xi[t] = w[t];
          yi[t] = w[t + 1];
        }
      }
    };
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(threads, B));
    if (nt == 1) {
      work(0, B);
      return;
    }
    std::vector<std::thread> pool;
    const int64_t per = (B + nt - 1) / nt;
    for (int64_t k = 0; k < nt; ++k) {
      const int64_t lo = k * per, hi = std::min(B, lo + per);
      if (lo < hi) pool.emplace_back(work, lo, hi);
    }
    for (auto& th : pool) th.join();
  }

 private:
  void release() {
    if (data_) munmap(const_cast<uint16_t*>(data_), bytes_);
    if (fd_ >= 0) ::close(fd_);
    data_ = nullptr;
    fd_ = -1;
  }
  [[noreturn]] void fail(const std::string& msg) {
    release();
    throw std::runtime_error(msg);
  }
  const uint16_t* data_ = nullptr;
  int64_t n_ = 0;
  size_t bytes_ = 0;
  int fd_ = -1;
};

}  // namespace dpa
