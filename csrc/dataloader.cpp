// Native token-file reader: memory-maps a flat uint16 token stream and assembles a training
// batch — B windows of T+1 tokens widened to int64, split into inputs x = w[:-1] and targets
// y = w[1:] — straight into a (pinned) staging tensor with a few worker threads. Replaces the
// per-sample numpy slicing loop of the reference loader (reference: single-gpu/train.py:233-245).
#include <ATen/ATen.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <torch/custom_class.h>
#include <torch/library.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace {

struct TokenFile : torch::CustomClassHolder {
  const uint16_t* data = nullptr;
  int64_t n = 0;
  size_t bytes = 0;
  int fd = -1;

  explicit TokenFile(const std::string& path) {
    fd = ::open(path.c_str(), O_RDONLY);
    TORCH_CHECK(fd >= 0, "TokenFile: cannot open ", path);
    struct stat st;
    TORCH_CHECK(fstat(fd, &st) == 0, "TokenFile: cannot stat ", path);
    bytes = (size_t)st.st_size;
    TORCH_CHECK(bytes >= 2 && bytes % 2 == 0, "TokenFile: ", path, " is not a uint16 token stream");
    void* p = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    TORCH_CHECK(p != MAP_FAILED, "TokenFile: mmap failed for ", path);
    madvise(p, bytes, MADV_RANDOM);
    data = static_cast<const uint16_t*>(p);
    n = (int64_t)(bytes / 2);
  }
  ~TokenFile() override {
    if (data) munmap(const_cast<uint16_t*>(data), bytes);
    if (fd >= 0) ::close(fd);
  }
  int64_t size() const { return n; }

  // out: int64 [2, B, T] (CPU, contiguous; may be pinned). starts: int64 [B].
  void gather(const at::Tensor& starts, int64_t T, at::Tensor out, int64_t threads) const {
    TORCH_CHECK(starts.device().is_cpu() && starts.scalar_type() == at::kLong && starts.is_contiguous(),
                "starts: contiguous CPU int64");
    const int64_t B = starts.numel();
    TORCH_CHECK(out.device().is_cpu() && out.scalar_type() == at::kLong && out.is_contiguous() &&
                    out.numel() == 2 * B * T,
                "out: contiguous CPU int64 [2, B, T]");
    const int64_t* s = starts.data_ptr<int64_t>();
    for (int64_t i = 0; i < B; ++i)
      TORCH_CHECK(s[i] >= 0 && s[i] + T + 1 <= n, "window ", s[i], " runs past the end of the token file");
    int64_t* x = out.data_ptr<int64_t>();
    int64_t* y = x + B * T;
    auto work = [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        const uint16_t* w = data + s[i];
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
};

}  // namespace

TORCH_LIBRARY_FRAGMENT(dpa, m) {
  m.class_<TokenFile>("TokenFile")
      .def(torch::init<std::string>())
      .def("size", &TokenFile::size)
      .def("gather", &TokenFile::gather);
}
