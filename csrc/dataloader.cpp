// Python face of the native token-file reader (token_file.h): a torch custom class that assembles a
// training batch straight into a (pinned) staging tensor with a few worker threads
// (reference loader being replaced: single-gpu/train.py:233-245).
#include <ATen/ATen.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include "token_file.h"

namespace {

struct TokenFile : torch::CustomClassHolder {
  dpa::TokenStream stream;
  explicit TokenFile(const std::string& path) : stream(path) {}
  int64_t size() const { return stream.size(); }

  // out: int64 [2, B, T] (CPU, contiguous; may be pinned). starts: int64 [B].
  void gather(const at::Tensor& starts, int64_t T, at::Tensor out, int64_t threads) const {
    TORCH_CHECK(starts.device().is_cpu() && starts.scalar_type() == at::kLong && starts.is_contiguous(),
                "starts: contiguous CPU int64");
    const int64_t B = starts.numel();
    TORCH_CHECK(T >= 1 && out.device().is_cpu() && out.scalar_type() == at::kLong && out.is_contiguous() &&
                    out.numel() == 2 * B * T,
                "out: contiguous CPU int64 [2, B, T]");
    int64_t* x = out.data_ptr<int64_t>();
    stream.gather(starts.data_ptr<int64_t>(), B, T, x, x + B * T, threads);
  }
};

}  // namespace

TORCH_LIBRARY_FRAGMENT(dpa, m) {
  m.class_<TokenFile>("TokenFile")
      .def(torch::init<std::string>())
      .def("size", &TokenFile::size)
      .def("gather", &TokenFile::gather);
}
