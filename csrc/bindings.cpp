// torch.ops.dpa.* : the only translation unit that sees PyTorch. It validates shapes, dtypes
// and strides on the host (a kernel launched on operands that do not match its indexing can
// fault the GPU), allocates outputs through the caching allocator and forwards raw pointers
// plus the current HIP stream to the launchers in kernels.h.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cmath>
#include <optional>
#include <tuple>
#include <vector>

#include "kernels.h"

namespace {

using at::Tensor;

hipStream_t cur_stream() { return c10::hip::getCurrentHIPStream().stream(); }

bool is_bf16(const Tensor& t) { return t.scalar_type() == at::kBFloat16; }

void check_act(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must live on the GPU");
  TORCH_CHECK(t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kFloat, name, " must be bf16 or fp32");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, " must be 16-byte aligned");
}

const void* opt_ptr(const std::optional<Tensor>& t) { return t.has_value() ? t->data_ptr() : nullptr; }

// ---------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor> layernorm_fwd(const Tensor& x, const std::optional<Tensor>& res,
                                                         const Tensor& w, const Tensor& b, double eps) {
  check_act(x, "x");
  check_act(w, "weight");
  check_act(b, "bias");
  const int64_t C = x.size(-1);
  TORCH_CHECK(C % 8 == 0 && C <= dpa::layernorm_max_cols(), "layernorm: unsupported width ", C);
  TORCH_CHECK(w.numel() == C && b.numel() == C, "layernorm: weight/bias size mismatch");
  TORCH_CHECK(w.scalar_type() == x.scalar_type() && b.scalar_type() == x.scalar_type(),
              "layernorm: x, weight and bias must share a dtype");
  const int64_t rows = x.numel() / C;
  Tensor y = at::empty_like(x);
  Tensor sum;
  if (res.has_value()) {
    check_act(*res, "residual");
    TORCH_CHECK(res->sizes() == x.sizes() && res->scalar_type() == x.scalar_type(), "layernorm: residual mismatch");
    sum = at::empty_like(x);
  }
  Tensor mean = at::empty({rows}, x.options().dtype(at::kFloat));
  Tensor rstd = at::empty({rows}, x.options().dtype(at::kFloat));
  if (rows > 0)
    dpa::layernorm_fwd(x.data_ptr(), opt_ptr(res), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                       res.has_value() ? sum.data_ptr() : nullptr, mean.data_ptr<float>(), rstd.data_ptr<float>(),
                       (int)rows, (int)C, (float)eps, is_bf16(x), cur_stream());
  return {y, sum, mean, rstd};
}

// dw_acc / db_acc given: the parameter gradients are ADDED into them (the engine's flat gradient views)
// and the returned dw / db are empty; otherwise fresh dw / db are returned.
std::tuple<Tensor, Tensor, Tensor> layernorm_bwd_impl(const Tensor& dy, const Tensor& x, const Tensor& w,
                                                      const Tensor& mean, const Tensor& rstd,
                                                      const std::optional<Tensor>& dres,
                                                      const std::optional<Tensor>& dw_acc,
                                                      const std::optional<Tensor>& db_acc) {
  check_act(dy, "dy");
  check_act(x, "x");
  check_act(w, "weight");
  const int64_t C = x.size(-1);
  TORCH_CHECK(C % 8 == 0 && C <= 4096, "layernorm backward: unsupported width ", C);
  TORCH_CHECK(dy.sizes() == x.sizes() && dy.scalar_type() == x.scalar_type(), "layernorm backward: dy mismatch");
  TORCH_CHECK(w.numel() == C && w.scalar_type() == x.scalar_type(), "layernorm backward: weight mismatch");
  const int64_t rows = x.numel() / C;
  TORCH_CHECK(mean.numel() == rows && rstd.numel() == rows && mean.scalar_type() == at::kFloat &&
                  rstd.scalar_type() == at::kFloat,
              "layernorm backward: statistics mismatch");
  if (dres.has_value()) {
    check_act(*dres, "dres");
    TORCH_CHECK(dres->sizes() == x.sizes() && dres->scalar_type() == x.scalar_type(), "dres mismatch");
  }
  const bool acc = dw_acc.has_value();
  if (acc) {
    TORCH_CHECK(db_acc.has_value(), "layernorm backward: dw_acc and db_acc go together");
    for (const Tensor* t : {&*dw_acc, &*db_acc})
      TORCH_CHECK(t->is_cuda() && t->is_contiguous() && t->numel() == C && t->scalar_type() == w.scalar_type(),
                  "layernorm backward: accumulation targets must be contiguous [C] of the weight dtype");
  }
  Tensor dx = at::empty_like(x);
  Tensor dw = acc ? *dw_acc : at::empty_like(w);
  Tensor db = acc ? *db_acc : at::empty_like(w);
  if (rows == 0) {
    if (!acc) {
      dw.zero_();
      db.zero_();
    }
    return {dx, acc ? Tensor() : dw, acc ? Tensor() : db};
  }
  const int nparts = dpa::layernorm_bwd_parts((int)rows, (int)C);
  Tensor scratch = at::empty({2 * (int64_t)nparts * C}, x.options().dtype(at::kFloat));
  dpa::layernorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr<float>(), rstd.data_ptr<float>(),
                     opt_ptr(dres), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), scratch.data_ptr<float>(), nparts,
                     (int)rows, (int)C, acc, is_bf16(x), cur_stream());
  return {dx, acc ? Tensor() : dw, acc ? Tensor() : db};
}

std::tuple<Tensor, Tensor, Tensor> layernorm_bwd(const Tensor& dy, const Tensor& x, const Tensor& w,
                                                 const Tensor& mean, const Tensor& rstd,
                                                 const std::optional<Tensor>& dres) {
  return layernorm_bwd_impl(dy, x, w, mean, rstd, dres, std::nullopt, std::nullopt);
}

Tensor layernorm_bwd_acc(const Tensor& dy, const Tensor& x, const Tensor& w, const Tensor& mean, const Tensor& rstd,
                         const std::optional<Tensor>& dres, Tensor dw_acc, Tensor db_acc) {
  return std::get<0>(layernorm_bwd_impl(dy, x, w, mean, rstd, dres, dw_acc, db_acc));
}

// ---------------------------------------------------------------------------------------
void adamw_step(Tensor master, Tensor m, Tensor v, const Tensor& grad, const std::optional<Tensor>& param_out,
                std::vector<int64_t> seg_ends, std::vector<double> seg_wds, double lr, double beta1, double beta2,
                double eps, int64_t step, const std::optional<Tensor>& norm2, double max_norm, double grad_mul,
                const std::optional<Tensor>& dyn) {
  const int64_t n = master.numel();
  TORCH_CHECK(n % 8 == 0, "adamw: flat buffers must be padded to a multiple of 8 elements");
  for (const Tensor* t : {&master, &m, &v}) {
    check_act(*t, "adamw state");
    TORCH_CHECK(t->scalar_type() == at::kFloat && t->numel() == n, "adamw: state must be fp32 of equal size");
  }
  check_act(grad, "grad");
  TORCH_CHECK(grad.numel() == n, "adamw: grad size mismatch");
  if (param_out.has_value()) {
    check_act(*param_out, "param_out");
    TORCH_CHECK(param_out->numel() == n, "adamw: param_out size mismatch");
  }
  TORCH_CHECK(seg_ends.size() == seg_wds.size() && !seg_ends.empty() && seg_ends.size() <= 64,
              "adamw: 1..64 weight-decay segments");
  dpa::AdamSegs segs;
  segs.nseg = (int)seg_ends.size();
  int64_t prev = 0;
  for (size_t i = 0; i < seg_ends.size(); ++i) {
    TORCH_CHECK(seg_ends[i] % 8 == 0 && seg_ends[i] >= prev, "adamw: segment ends must ascend in multiples of 8");
    prev = seg_ends[i];
    segs.end[i] = seg_ends[i];
    segs.wd[i] = (float)seg_wds[i];
  }
  TORCH_CHECK(prev == n, "adamw: segments must cover the buffer");
  const float* n2 = nullptr;
  if (norm2.has_value()) {
    TORCH_CHECK(norm2->is_cuda() && norm2->scalar_type() == at::kFloat && norm2->numel() == 1, "norm2: 1 fp32");
    n2 = norm2->data_ptr<float>();
  }
  const float* dynp = nullptr;
  if (dyn.has_value()) {
    TORCH_CHECK(dyn->is_cuda() && dyn->scalar_type() == at::kFloat && dyn->numel() == 3 && dyn->is_contiguous(),
                "dyn: 3 fp32 {lr, bc1, bc2_sqrt} on the device");
    dynp = dyn->data_ptr<float>();
  }
  if (n == 0) return;
  const float bc1 = 1.f - (float)std::pow(beta1, (double)step);
  const float bc2s = (float)std::sqrt(1.0 - std::pow(beta2, (double)step));
  dpa::adamw_step(master.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(), grad.data_ptr(), is_bf16(grad),
                  param_out.has_value() ? param_out->data_ptr() : nullptr,
                  param_out.has_value() ? is_bf16(*param_out) : false, n, segs, (float)lr, (float)beta1,
                  (float)beta2, (float)eps, bc1, bc2s, n2, (float)max_norm, (float)grad_mul, dynp, cur_stream());
}

void sumsq(const Tensor& g, Tensor out, bool accumulate) {
  check_act(g, "g");
  TORCH_CHECK(g.numel() % 8 == 0, "sumsq: length must be a multiple of 8");
  TORCH_CHECK(out.is_cuda() && out.scalar_type() == at::kFloat && out.numel() == 1, "sumsq: out is 1 fp32");
  if (g.numel() == 0) {
    if (!accumulate) out.zero_();
    return;
  }
  Tensor partials = at::empty({dpa::sumsq_parts(g.numel())}, out.options());
  dpa::sumsq(g.data_ptr(), is_bf16(g), g.numel(), partials.data_ptr<float>(), out.data_ptr<float>(), accumulate,
             cur_stream());
}

// ---------------------------------------------------------------------------------------
void check_logits(const Tensor& logits, const Tensor& targets) {
  TORCH_CHECK(logits.is_cuda() && logits.dim() == 2 && logits.stride(1) == 1, "logits: [rows, V], unit inner stride");
  TORCH_CHECK(logits.scalar_type() == at::kBFloat16 || logits.scalar_type() == at::kFloat, "logits: bf16 or fp32");
  TORCH_CHECK(logits.size(1) % 8 == 0 && logits.stride(0) % 8 == 0, "logits: V and row stride multiples of 8");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(logits.data_ptr()) % 16 == 0, "logits: 16-byte aligned");
  TORCH_CHECK(targets.is_cuda() && targets.scalar_type() == at::kLong && targets.is_contiguous() &&
                  targets.numel() == logits.size(0),
              "targets: int64 [rows]");
}

Tensor xent_fused(Tensor logits, const Tensor& targets, const Tensor& grad_scale, int64_t ignore_index,
                  bool write_grad) {
  check_logits(logits, targets);
  TORCH_CHECK(grad_scale.is_cuda() && grad_scale.scalar_type() == at::kFloat && grad_scale.numel() == 1);
  Tensor loss = at::empty({logits.size(0)}, logits.options().dtype(at::kFloat));
  dpa::xent(0, logits.data_ptr(), is_bf16(logits), (const long long*)targets.data_ptr<int64_t>(),
            loss.data_ptr<float>(), nullptr, nullptr, nullptr, nullptr, grad_scale.data_ptr<float>(),
            (int)logits.size(0), (int)logits.size(1), (int)logits.stride(0), 0, ignore_index, write_grad,
            cur_stream());
  return loss;
}

std::tuple<Tensor, Tensor, Tensor> xent_stats(const Tensor& logits, const Tensor& targets, int64_t vocab_start,
                                              int64_t ignore_index) {
  check_logits(logits, targets);
  auto opt = logits.options().dtype(at::kFloat);
  Tensor mx = at::empty({logits.size(0)}, opt), sm = at::empty({logits.size(0)}, opt),
         tg = at::empty({logits.size(0)}, opt);
  dpa::xent(1, logits.data_ptr(), is_bf16(logits), (const long long*)targets.data_ptr<int64_t>(), nullptr,
            mx.data_ptr<float>(), sm.data_ptr<float>(), tg.data_ptr<float>(), nullptr, nullptr, (int)logits.size(0),
            (int)logits.size(1), (int)logits.stride(0), (int)vocab_start, ignore_index, false, cur_stream());
  return {mx, sm, tg};
}

Tensor xent_grad(Tensor logits, const Tensor& targets, const Tensor& lse, const Tensor& tgt_logit,
                 const Tensor& grad_scale, int64_t vocab_start, int64_t ignore_index, bool write_grad) {
  check_logits(logits, targets);
  for (const Tensor* t : {&lse, &tgt_logit})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->is_contiguous() && t->numel() == logits.size(0));
  TORCH_CHECK(grad_scale.is_cuda() && grad_scale.scalar_type() == at::kFloat && grad_scale.numel() == 1);
  Tensor loss = at::empty({logits.size(0)}, logits.options().dtype(at::kFloat));
  dpa::xent(2, logits.data_ptr(), is_bf16(logits), (const long long*)targets.data_ptr<int64_t>(),
            loss.data_ptr<float>(), nullptr, nullptr, const_cast<float*>(tgt_logit.data_ptr<float>()),
            lse.data_ptr<float>(), grad_scale.data_ptr<float>(), (int)logits.size(0), (int)logits.size(1),
            (int)logits.stride(0), (int)vocab_start, ignore_index, write_grad, cur_stream());
  return loss;
}

// ---------------------------------------------------------------------------------------
Tensor embed_fwd(const Tensor& idx, const Tensor& wte, const std::optional<Tensor>& pos, int64_t pos0) {
  check_act(wte, "wte");
  TORCH_CHECK(idx.is_cuda() && idx.scalar_type() == at::kLong && idx.is_contiguous() && idx.dim() == 2,
              "idx: int64 [B, T]");
  const int64_t C = wte.size(1), T = idx.size(1);
  TORCH_CHECK(C % 8 == 0, "embedding width must be a multiple of 8");
  if (pos.has_value()) {
    check_act(*pos, "pos");
    TORCH_CHECK(pos->size(1) == C && pos->scalar_type() == wte.scalar_type() && pos0 + T <= pos->size(0),
                "position table too short or mismatched");
  }
  Tensor out = at::empty({idx.size(0), T, C}, wte.options());
  dpa::embed_fwd((const long long*)idx.data_ptr<int64_t>(), wte.data_ptr(), opt_ptr(pos), out.data_ptr(),
                 idx.numel(), (int)T, (int)pos0, (int)C, wte.size(0), is_bf16(wte), cur_stream());
  return out;
}

// Caches [B, S, H, hs] contiguous; new rows [B, 1, H, hs] with packed heads (any batch stride); pos int64 [1].
void kv_append(Tensor k_cache, Tensor v_cache, const Tensor& k_new, const Tensor& v_new, const Tensor& pos) {
  TORCH_CHECK(k_cache.is_cuda() && k_cache.scalar_type() == at::kBFloat16 && k_cache.is_contiguous() &&
                  k_cache.dim() == 4 && v_cache.sizes() == k_cache.sizes() && v_cache.is_contiguous() &&
                  v_cache.scalar_type() == at::kBFloat16,
              "kv_append: contiguous bf16 caches [B, S, H, hs]");
  const int64_t B = k_cache.size(0), S = k_cache.size(1), H = k_cache.size(2), hs = k_cache.size(3), W = H * hs;
  for (const Tensor* t : {&k_new, &v_new}) {
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kBFloat16 && t->dim() == 4 && t->size(0) == B && t->size(1) == 1 &&
                    t->size(2) == H && t->size(3) == hs && t->stride(3) == 1 && (H == 1 || t->stride(2) == hs) &&
                    t->stride(0) % 8 == 0 && reinterpret_cast<uintptr_t>(t->data_ptr()) % 16 == 0,
                "kv_append: new rows [B, 1, H, hs], packed heads, 16-byte aligned");
  }
  TORCH_CHECK(W % 8 == 0 && pos.is_cuda() && pos.scalar_type() == at::kLong && pos.numel() == 1, "kv_append: pos int64 [1]");
  dpa::kv_append(k_cache.data_ptr(), v_cache.data_ptr(), k_new.data_ptr(), v_new.data_ptr(),
                 (const long long*)pos.data_ptr<int64_t>(), (int)B, (int)S, (int)W, k_new.stride(0), v_new.stride(0),
                 cur_stream());
}

void embed_bwd(const Tensor& dout, const Tensor& ids_sorted, const Tensor& perm, Tensor dwte) {
  check_act(dout, "dout");
  check_act(dwte, "dwte");
  const int64_t C = dwte.size(1), N = ids_sorted.numel();
  TORCH_CHECK(dout.numel() == N * C && dout.scalar_type() == dwte.scalar_type(), "embed_bwd: dout mismatch");
  for (const Tensor* t : {&ids_sorted, &perm})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kLong && t->is_contiguous() && t->numel() == N);
  if (N == 0) return;
  Tensor scratch = at::empty({2 * (int64_t)dpa::embed_bwd_chunks(N) * C}, dwte.options().dtype(at::kFloat));
  dpa::embed_bwd(dout.data_ptr(), (const long long*)ids_sorted.data_ptr<int64_t>(),
                 (const long long*)perm.data_ptr<int64_t>(), dwte.data_ptr(), scratch.data_ptr<float>(), N, (int)C,
                 dwte.size(0), is_bf16(dwte), cur_stream());
}

void pos_bwd(const Tensor& dout, Tensor dpos, int64_t pos0) {
  check_act(dout, "dout");
  check_act(dpos, "dpos");
  TORCH_CHECK(dout.dim() == 3 && dout.size(2) == dpos.size(1) && dout.scalar_type() == dpos.scalar_type() &&
                  pos0 + dout.size(1) <= dpos.size(0),
              "pos_bwd: shape mismatch");
  dpa::pos_bwd(dout.data_ptr(), dpos.data_ptr(), (int)dout.size(0), (int)dout.size(1), (int)pos0, (int)dout.size(2),
               is_bf16(dpos), cur_stream());
}

// ---------------------------------------------------------------------------------------
Tensor gelu_fwd(const Tensor& x) {
  check_act(x, "x");
  TORCH_CHECK(x.numel() % 8 == 0);
  Tensor y = at::empty_like(x);
  dpa::gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), is_bf16(x), cur_stream());
  return y;
}
Tensor gelu_bwd(const Tensor& dy, const Tensor& x) {
  check_act(x, "x");
  check_act(dy, "dy");
  TORCH_CHECK(x.numel() % 8 == 0 && dy.sizes() == x.sizes() && dy.scalar_type() == x.scalar_type());
  Tensor dx = at::empty_like(x);
  dpa::gelu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), is_bf16(x), cur_stream());
  return dx;
}
void check_sample_args(const Tensor& logits, const Tensor& seed) {
  TORCH_CHECK(logits.is_cuda() && logits.dim() == 2 && logits.stride(1) == 1 &&
                  (logits.scalar_type() == at::kBFloat16 || logits.scalar_type() == at::kFloat),
              "sample_topk: logits [rows, V] bf16/fp32 with unit inner stride on the GPU");
  TORCH_CHECK(seed.is_cuda() && seed.scalar_type() == at::kLong && seed.numel() == 1, "sample_topk: int64 [1] seed on the GPU");
  TORCH_CHECK(logits.size(1) < (1 << 30) && dpa::sample_topk_ok((int)logits.size(1)) &&
                  (logits.stride(0) * logits.element_size()) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(logits.data_ptr()) % 16 == 0,
              "sample_topk: vocabulary must be a multiple of 8, at most 57344, rows 16-byte aligned");
}

Tensor sample_topk(const Tensor& logits, double temperature, int64_t topk, const Tensor& seed) {
  check_sample_args(logits, seed);
  Tensor out = at::empty({logits.size(0), 1}, logits.options().dtype(at::kLong));
  dpa::sample_topk(logits.data_ptr(), logits.stride(0), (int)logits.size(0), (int)logits.size(1), (float)temperature,
                   (int)std::max<int64_t>(topk, 0), (const long long*)seed.data_ptr<int64_t>(),
                   (long long*)out.data_ptr<int64_t>(), is_bf16(logits), cur_stream());
  return out;
}

// Sampling plus the bookkeeping of one captured decode step: the token goes to `tok` [rows, 1] (the next step's
// input) and to history[:, history_pos]; pos, len_after and history_pos then advance by one (see SampleStep).
void sample_topk_step(const Tensor& logits, double temperature, int64_t topk, const Tensor& seed, Tensor tok,
                      Tensor history, Tensor history_pos, Tensor pos, Tensor len_after, Tensor ticket) {
  check_sample_args(logits, seed);
  const int64_t rows = logits.size(0);
  auto scalar = [](const Tensor& t, at::ScalarType ty) { return t.is_cuda() && t.scalar_type() == ty && t.numel() == 1; };
  TORCH_CHECK(tok.is_cuda() && tok.scalar_type() == at::kLong && tok.is_contiguous() && tok.numel() == rows,
              "sample_topk_step: tok is contiguous int64 [rows, 1] on the GPU");
  TORCH_CHECK(history.is_cuda() && history.scalar_type() == at::kLong && history.dim() == 2 && history.size(0) == rows &&
                  history.stride(1) == 1,
              "sample_topk_step: history is int64 [rows, steps] on the GPU");
  TORCH_CHECK(scalar(history_pos, at::kLong) && scalar(pos, at::kLong) && scalar(len_after, at::kInt) && scalar(ticket, at::kInt),
              "sample_topk_step: history_pos / pos int64 [1], len_after / ticket int32 [1], on the GPU");
  dpa::SampleStep st;
  st.history = (long long*)history.data_ptr<int64_t>();
  st.history_ld = history.stride(0);
  st.history_pos = (long long*)history_pos.data_ptr<int64_t>();
  st.pos = (long long*)pos.data_ptr<int64_t>();
  st.len_after = len_after.data_ptr<int>();
  st.ticket = (unsigned*)ticket.data_ptr<int>();
  dpa::sample_topk(logits.data_ptr(), logits.stride(0), (int)rows, (int)logits.size(1), (float)temperature,
                   (int)std::max<int64_t>(topk, 0), (const long long*)seed.data_ptr<int64_t>(),
                   (long long*)tok.data_ptr<int64_t>(), is_bf16(logits), cur_stream(), st);
}

void spin(int64_t blocks, int64_t usec, Tensor sink) {
  TORCH_CHECK(sink.is_cuda() && sink.scalar_type() == at::kFloat && sink.numel() >= 1 && blocks <= 256 && usec <= 20000,
              "spin: fp32 sink on the GPU, at most 256 blocks and 20 ms");
  dpa::spin((int)blocks, (int)usec, sink.data_ptr<float>(), cur_stream());
}

Tensor act_fwd(const Tensor& x, int64_t kind) {
  check_act(x, "x");
  TORCH_CHECK(x.numel() % 8 == 0 && kind >= 0 && kind < dpa::act_kinds(), "act_fwd: bad size or kind");
  Tensor y = at::empty_like(x);
  dpa::act_apply((int)kind, nullptr, x.data_ptr(), y.data_ptr(), x.numel(), is_bf16(x), cur_stream());
  return y;
}
Tensor act_bwd(const Tensor& dy, const Tensor& x, int64_t kind) {
  check_act(x, "x");
  check_act(dy, "dy");
  TORCH_CHECK(x.numel() % 8 == 0 && dy.sizes() == x.sizes() && dy.scalar_type() == x.scalar_type() && kind >= 0 &&
                  kind < dpa::act_kinds(),
              "act_bwd: bad size or kind");
  Tensor dx = at::empty_like(x);
  dpa::act_apply((int)kind, dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), is_bf16(x), cur_stream());
  return dx;
}
Tensor swiglu_fwd(const Tensor& h) {
  check_act(h, "h");
  const int64_t U = h.size(-1) / 2;
  TORCH_CHECK(h.size(-1) % 16 == 0, "swiglu: inner width must be a multiple of 16");
  auto sizes = h.sizes().vec();
  sizes.back() = U;
  Tensor y = at::empty(sizes, h.options());
  dpa::swiglu_fwd(h.data_ptr(), y.data_ptr(), h.numel() / (2 * U), (int)U, is_bf16(h), cur_stream());
  return y;
}
Tensor swiglu_bwd(const Tensor& dy, const Tensor& h) {
  check_act(h, "h");
  check_act(dy, "dy");
  const int64_t U = h.size(-1) / 2;
  TORCH_CHECK(h.size(-1) % 16 == 0 && dy.numel() * 2 == h.numel() && dy.scalar_type() == h.scalar_type());
  Tensor dh = at::empty_like(h);
  dpa::swiglu_bwd(dy.data_ptr(), h.data_ptr(), dh.data_ptr(), h.numel() / (2 * U), (int)U, is_bf16(h), cur_stream());
  return dh;
}

// ---------------------------------------------------------------------------------------
// A [B, T, H, hs] view with unit inner stride and head stride hs (a slice of a fused projection).
void check_heads(const Tensor& t, const char* name, int64_t hs) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kBFloat16 && t.dim() == 4, name, ": bf16 [B, T, H, hs]");
  TORCH_CHECK(t.size(3) == hs && t.stride(3) == 1 && (t.size(2) == 1 || t.stride(2) == hs), name,
              ": heads must be packed");
  TORCH_CHECK(t.stride(1) % 8 == 0 && t.stride(0) % 8 == 0 && reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0,
              name, ": 16-byte aligned rows");
}

dpa::AttnParams attn_params(const Tensor& q, const Tensor& k, const Tensor& v, double scale) {
  TORCH_CHECK(q.dim() == 4 && v.dim() == 4, "attention: [B, T, H, width] tensors");
  const int64_t hs = q.size(3), dv = v.size(3);
  TORCH_CHECK(((hs == 64 || hs == 128) && dv == hs) || dpa::attn_wide_ok((int)hs, (int)dv),
              "attention: (key width, value width) must be (64, 64), (128, 128), (128, 64) or (256, 128), got (", hs, ", ", dv, ")");
  check_heads(q, "q", hs);
  check_heads(k, "k", hs);
  check_heads(v, "v", dv);
  TORCH_CHECK(k.size(0) == q.size(0) && v.size(0) == k.size(0) && v.size(1) == k.size(1) && v.size(2) == k.size(2),
              "attention: k/v shape mismatch");
  TORCH_CHECK(q.size(2) % k.size(2) == 0, "attention: n_head must be a multiple of n_kv_heads");
  TORCH_CHECK(k.size(1) >= q.size(1), "attention: Tk < Tq");
  dpa::AttnParams p{};
  p.q = q.data_ptr();
  p.k = k.data_ptr();
  p.v = v.data_ptr();
  p.B = (int)q.size(0);
  p.Tq = (int)q.size(1);
  p.nh = (int)q.size(2);
  p.Tk = (int)k.size(1);
  p.nkv = (int)k.size(2);
  p.hs = (int)hs;
  p.vw = (int)dv;
  p.q_bs = q.stride(0);
  p.q_ts = (int)q.stride(1);
  p.k_bs = k.stride(0);
  p.k_ts = (int)k.stride(1);
  p.v_bs = v.stride(0);
  p.v_ts = (int)v.stride(1);
  p.scale = (float)scale;
  return p;
}

// Dropout: p_drop in [0, 1) and, when positive, a one-element int64 device tensor holding the seed.
void set_dropout(dpa::AttnParams& p, double p_drop, const std::optional<Tensor>& seed) {
  TORCH_CHECK(p_drop >= 0.0 && p_drop < 1.0, "attention: dropout probability must be in [0, 1)");
  p.p_drop = (float)p_drop;
  p.seed = nullptr;
  if (p_drop > 0.0) {
    TORCH_CHECK(seed.has_value() && seed->is_cuda() && seed->scalar_type() == at::kLong && seed->numel() == 1,
                "attention: dropout needs a one-element int64 seed tensor on the GPU");
    p.seed = (const long long*)seed->data_ptr<int64_t>();
  }
}

Tensor attn_dropout_mask(int64_t B, int64_t nh, int64_t Tq, int64_t Tk, double p_drop, const Tensor& seed) {
  TORCH_CHECK(seed.is_cuda() && seed.scalar_type() == at::kLong && seed.numel() == 1, "seed: int64 [1] on the GPU");
  TORCH_CHECK(p_drop >= 0.0 && p_drop < 1.0 && B >= 0 && nh >= 0 && Tq >= 0 && Tk > 0, "attn_dropout_mask: bad arguments");
  Tensor keep = at::empty({B, nh, Tq, Tk}, seed.options().dtype(at::kByte));
  dpa::attn_dropout_mask(keep.data_ptr<uint8_t>(), (int)B, (int)nh, (int)Tq, (int)Tk, (float)p_drop,
                         (const long long*)seed.data_ptr<int64_t>(), cur_stream());
  return keep.to(at::kBool);
}

std::tuple<Tensor, Tensor> attn_fwd(const Tensor& q, const Tensor& k, const Tensor& v, double scale, double p_drop,
                                    const std::optional<Tensor>& seed, const std::optional<Tensor>& kv_len) {
  dpa::AttnParams p = attn_params(q, k, v, scale);
  set_dropout(p, p_drop, seed);
  if (kv_len.has_value()) {
    TORCH_CHECK(kv_len->is_cuda() && kv_len->scalar_type() == at::kInt && kv_len->numel() == 1,
                "attention: kv_len must be a one-element int32 tensor on the GPU");
    p.kv_len = kv_len->data_ptr<int>();
  }
  TORCH_CHECK(!kv_len.has_value() || v.size(3) == q.size(3), "attention: a cache fill level needs equal key and value widths");
  Tensor o = at::empty({q.size(0), q.size(1), q.size(2), v.size(3)}, q.options());
  Tensor lse = at::empty({q.size(0), q.size(2), q.size(1)}, q.options().dtype(at::kFloat));
  p.o = o.data_ptr();
  p.o_bs = o.stride(0);
  p.o_ts = (int)o.stride(1);
  p.lse = lse.data_ptr<float>();
  dpa::attn_fwd(p, cur_stream());
  return {o, lse};
}

// One new token per sequence over a preallocated cache: q [B, 1, nh, hs], caches [B, S, nkv, hs], kv_len int32 [1]
// on the device (csrc/attn_decode.hip).
Tensor attn_decode(const Tensor& q, const Tensor& k, const Tensor& v, double scale, const Tensor& kv_len) {
  const int64_t hs = q.size(3);
  TORCH_CHECK(q.dim() == 4 && q.size(1) == 1 && (hs == 64 || hs == 128), "attn_decode: q is [B, 1, nh, 64 | 128]");
  check_heads(q, "q", hs);
  check_heads(k, "k", hs);
  check_heads(v, "v", hs);
  TORCH_CHECK(k.size(0) == q.size(0) && v.sizes() == k.sizes() && q.size(2) % k.size(2) == 0, "attn_decode: shape mismatch");
  TORCH_CHECK(kv_len.is_cuda() && kv_len.scalar_type() == at::kInt && kv_len.numel() == 1,
              "attn_decode: kv_len must be a one-element int32 tensor on the GPU");
  Tensor o = at::empty({q.size(0), 1, q.size(2), hs}, q.options());
  dpa::AttnDecodeArgs p{};
  p.q = q.data_ptr();
  p.k = k.data_ptr();
  p.v = v.data_ptr();
  p.o = o.data_ptr();
  p.q_bs = q.stride(0);
  p.o_bs = o.stride(0);
  p.k_bs = k.stride(0);
  p.v_bs = v.stride(0);
  p.k_ts = (int)k.stride(1);
  p.v_ts = (int)v.stride(1);
  p.B = (int)q.size(0);
  p.nh = (int)q.size(2);
  p.nkv = (int)k.size(2);
  p.hs = (int)hs;
  p.S = (int)k.size(1);
  p.scale = (float)scale;
  p.kv_len = kv_len.data_ptr<int>();
  dpa::attn_decode(p, cur_stream());
  return o;
}

// fp32 [positions, hs / 2] cos / sin tables on the GPU covering positions [0, need)
void check_rope_tables(const Tensor& c, const Tensor& s, int64_t hs, int64_t need, const char* who) {
  TORCH_CHECK(c.is_cuda() && s.is_cuda() && c.scalar_type() == at::kFloat && s.scalar_type() == at::kFloat &&
                  c.dim() == 2 && c.sizes() == s.sizes() && c.is_contiguous() && s.is_contiguous() && c.size(1) * 2 == hs,
              who, ": rotary tables are contiguous fp32 [positions, hs / 2] on the GPU");
  TORCH_CHECK(need <= c.size(0), who, ": rotary tables hold ", c.size(0), " positions, ", need, " needed");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(c.data_ptr()) % 16 == 0 && reinterpret_cast<uintptr_t>(s.data_ptr()) % 16 == 0,
              who, ": rotary tables must be 16-byte aligned");
}

void attn_bwd(const Tensor& dout, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& o,
              const Tensor& lse, Tensor dq, Tensor dk, Tensor dv, double scale, double p_drop,
              const std::optional<Tensor>& seed, const std::optional<Tensor>& rope_cos,
              const std::optional<Tensor>& rope_sin, int64_t rope_pos0) {
  dpa::AttnParams p = attn_params(q, k, v, scale);
  set_dropout(p, p_drop, seed);
  TORCH_CHECK(rope_cos.has_value() == rope_sin.has_value(), "attention backward: both rotary tables or none");
  if (rope_cos.has_value()) {
    TORCH_CHECK(rope_pos0 >= 0, "attention backward: rope_pos0 >= 0");
    TORCH_CHECK(v.size(3) == q.size(3), "attention backward: rotary tables need equal key and value widths");
    check_rope_tables(*rope_cos, *rope_sin, q.size(3), rope_pos0 + k.size(1), "attention backward");
    p.rope_cos = rope_cos->data_ptr<float>();
    p.rope_sin = rope_sin->data_ptr<float>();
    p.rope_pos0 = (int)rope_pos0;
  }
  const int64_t hs = q.size(3), vw = v.size(3);
  check_heads(dout, "dout", vw);
  check_heads(o, "o", vw);
  check_heads(dq, "dq", hs);
  check_heads(dk, "dk", hs);
  check_heads(dv, "dv", vw);
  TORCH_CHECK(dout.sizes() == o.sizes() && o.size(0) == q.size(0) && o.size(1) == q.size(1) && o.size(2) == q.size(2) &&
                  dq.sizes() == q.sizes() && dk.sizes() == k.sizes() && dv.sizes() == v.sizes(),
              "attention backward: shape mismatch");
  TORCH_CHECK(lse.is_cuda() && lse.scalar_type() == at::kFloat && lse.is_contiguous() &&
                  lse.numel() == q.size(0) * q.size(1) * q.size(2),
              "attention backward: lse mismatch");
  // row constants handed from the dQ kernel to the dK/dV kernel: [2][B * nh][Tq rounded up to 128]
  Tensor delta = at::empty({2, q.size(0) * q.size(2), (q.size(1) + 127) / 128 * 128}, lse.options());
  p.o = o.data_ptr();
  p.o_bs = o.stride(0);
  p.o_ts = (int)o.stride(1);
  p.lse = lse.data_ptr<float>();
  p.dout = dout.data_ptr();
  p.do_bs = dout.stride(0);
  p.do_ts = (int)dout.stride(1);
  p.delta = delta.data_ptr<float>();
  p.dq = dq.data_ptr();
  p.dq_bs = dq.stride(0);
  p.dq_ts = (int)dq.stride(1);
  p.dk = dk.data_ptr();
  p.dk_bs = dk.stride(0);
  p.dk_ts = (int)dk.stride(1);
  p.dv = dv.data_ptr();
  p.dv_bs = dv.stride(0);
  p.dv_ts = (int)dv.stride(1);
  dpa::attn_bwd(p, cur_stream());
}

void rope_(Tensor x, const Tensor& cos_t, const Tensor& sin_t, int64_t pos0, bool inverse) {
  TORCH_CHECK(x.is_cuda() && x.dim() == 4 && (x.scalar_type() == at::kBFloat16 || x.scalar_type() == at::kFloat),
              "rope: [B, T, H, hs] bf16/fp32");
  const int64_t hs = x.size(3);
  TORCH_CHECK(hs % 8 == 0 && x.stride(3) == 1 && (x.size(2) == 1 || x.stride(2) == hs), "rope: packed heads, hs % 8 == 0");
  const int64_t esz = x.element_size();
  TORCH_CHECK((x.stride(1) * esz) % 16 == 0 && (x.stride(0) * esz) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0,
              "rope: 16-byte aligned rows");
  for (const Tensor* t : {&cos_t, &sin_t})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->is_contiguous() && t->dim() == 2 &&
                    t->size(1) == hs / 2 && pos0 + x.size(1) <= t->size(0),
                "rope: cos/sin tables are fp32 [>=pos0+T, hs/2]");
  dpa::rope_inplace(x.data_ptr(), x.stride(0), (int)x.stride(1), (int)x.size(0), (int)x.size(1), (int)x.size(2),
                    (int)hs, cos_t.data_ptr<float>(), sin_t.data_ptr<float>(), (int)pos0, inverse, is_bf16(x),
                    cur_stream());
}

// ---------------------------------------------------------------------------------------
void check_rows2d(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kBFloat16 && t.dim() == 2 && t.stride(1) == 1, name,
              ": bf16 [rows, cols] with unit inner stride");
  TORCH_CHECK(t.stride(0) % 8 == 0 && reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name,
              ": rows must be 16-byte aligned");
}

void wgrad_into(const Tensor& dy, const Tensor& x, Tensor ws, const std::optional<Tensor>& dbias) {
  check_rows2d(dy, "dy");
  check_rows2d(x, "x");
  const int64_t M = dy.size(0), N = dy.size(1), K = x.size(1);
  TORCH_CHECK(x.size(0) == M, "wgrad: dy and x must have the same number of rows");
  TORCH_CHECK(N % 128 == 0 && K % 128 == 0, "wgrad: output dims must be multiples of 128, got ", N, " x ", K);
  TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.numel() == N * K,
              "wgrad: workspace must be fp32 [N, K]");
  float* db = nullptr;
  if (dbias.has_value()) {
    TORCH_CHECK(dbias->is_cuda() && dbias->scalar_type() == at::kFloat && dbias->is_contiguous() &&
                    dbias->numel() == N,
                "wgrad: dbias must be fp32 [N]");
    db = dbias->data_ptr<float>();
  }
  const int splits = dpa::wgrad_splits((int)M, (int)N, (int)K);
  dpa::wgrad_gemm(dy.data_ptr(), x.data_ptr(), ws.data_ptr<float>(), db, (int)M, (int)N, (int)K, dy.stride(0),
                  x.stride(0), splits, cur_stream());
}

void ws_to(const Tensor& ws, Tensor out, bool accumulate) {
  TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous(), "ws_to: fp32 workspace");
  check_act(out, "out");
  TORCH_CHECK(out.numel() > 0 && ws.numel() % out.numel() == 0 && out.numel() % 8 == 0, "ws_to: size mismatch");
  dpa::wgrad_finish(ws.data_ptr<float>(), out.data_ptr(), out.numel(), (int)(ws.numel() / out.numel()), accumulate,
                    is_bf16(out), cur_stream());
}

// dw [N, K] = dy[M, N]^T x[M, K] through per-split fp32 slabs (no atomics).
Tensor wgrad2(const Tensor& dy, const Tensor& x) {
  check_rows2d(dy, "dy");
  check_rows2d(x, "x");
  const int64_t M = dy.size(0), N = dy.size(1), K = x.size(1);
  TORCH_CHECK(x.size(0) == M && dpa::wgrad2_ok((int)M, (int)N, (int)K), "wgrad2: unsupported shape ", M, "x", N, "x", K);
  const int splits = dpa::wgrad2_splits((int)M, (int)N, (int)K);
  Tensor slabs = at::empty({splits, N, K}, dy.options().dtype(at::kFloat));
  dpa::wgrad2_gemm(dy.data_ptr(), x.data_ptr(), slabs.data_ptr<float>(), (int)M, (int)N, (int)K, dy.stride(0),
                   x.stride(0), splits, cur_stream());
  Tensor dw = at::empty({N, K}, dy.options());
  dpa::wgrad_finish(slabs.data_ptr<float>(), dw.data_ptr(), N * K, splits, false, true, cur_stream());
  return dw;
}

// The per-split fp32 slabs [splits, N, K] of dy^T x (the caller folds them with ws_to, possibly
// straight into a flat gradient buffer with accumulate = true).
Tensor wgrad2_slabs(const Tensor& dy, const Tensor& x) {
  check_rows2d(dy, "dy");
  check_rows2d(x, "x");
  const int64_t M = dy.size(0), N = dy.size(1), K = x.size(1);
  TORCH_CHECK(x.size(0) == M && dpa::wgrad2_ok((int)M, (int)N, (int)K), "wgrad2: unsupported shape ", M, "x", N, "x", K);
  const int splits = dpa::wgrad2_splits((int)M, (int)N, (int)K);
  Tensor slabs = at::empty({splits, N, K}, dy.options().dtype(at::kFloat));
  dpa::wgrad2_gemm(dy.data_ptr(), x.data_ptr(), slabs.data_ptr<float>(), (int)M, (int)N, (int)K, dy.stride(0),
                   x.stride(0), splits, cur_stream());
  return slabs;
}

// ws[N, K] (fp32) += dy^T x in one pass over the tokens (accumulation across row chunks).
void wgrad2_accum(const Tensor& dy, const Tensor& x, Tensor ws) {
  check_rows2d(dy, "dy");
  check_rows2d(x, "x");
  const int64_t M = dy.size(0), N = dy.size(1), K = x.size(1);
  TORCH_CHECK(x.size(0) == M && dpa::wgrad2_ok((int)M, (int)N, (int)K), "wgrad2: unsupported shape ", M, "x", N, "x", K);
  TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.numel() == N * K);
  dpa::wgrad2_gemm(dy.data_ptr(), x.data_ptr(), ws.data_ptr<float>(), (int)M, (int)N, (int)K, dy.stride(0),
                   x.stride(0), 0, cur_stream());
}

Tensor colsum(const Tensor& x) {
  TORCH_CHECK(x.is_cuda() && x.dim() == 2 && x.stride(1) == 1 &&
                  (x.scalar_type() == at::kBFloat16 || x.scalar_type() == at::kFloat),
              "colsum: bf16/fp32 [M, N] with unit inner stride");
  TORCH_CHECK(x.size(1) % 8 == 0 && x.stride(0) % 8 == 0 && reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0,
              "colsum: 16-byte aligned rows, N % 8 == 0");
  Tensor out = at::zeros({x.size(1)}, x.options().dtype(at::kFloat));
  dpa::colsum(x.data_ptr(), out.data_ptr<float>(), x.size(0), (int)x.size(1), x.stride(0), is_bf16(x), cur_stream());
  return out;
}

// ---------------------------------------------------------------------------------------
// Grouped GEMM over 128-row blocks: c = a * W_g^T (w_is_kn = false, W_g [w_rows = N, w_cols = K]) or
// c = a * W_g (w_is_kn = true, W_g [w_rows = K, w_cols = N]). Rows of skipped blocks are left
// uninitialised.
Tensor grouped_gemm(const Tensor& a, const Tensor& w_ptrs, int64_t w_rows, int64_t w_cols, bool w_is_kn,
                    const Tensor& block_group) {
  check_rows2d(a, "a");
  const int64_t rows = a.size(0), K = a.size(1);
  const int64_t N = w_is_kn ? w_cols : w_rows;
  TORCH_CHECK((w_is_kn ? w_rows : w_cols) == K, "grouped_gemm: contraction mismatch");
  TORCH_CHECK(K % 64 == 0 && N % 8 == 0 && w_cols % 8 == 0, "grouped_gemm: K % 64 == 0 and N % 8 == 0 required");
  TORCH_CHECK(rows % 128 == 0, "grouped_gemm: rows must be padded to a multiple of 128");
  TORCH_CHECK(w_ptrs.is_cuda() && w_ptrs.scalar_type() == at::kLong && w_ptrs.is_contiguous(), "w_ptrs: int64 cuda");
  TORCH_CHECK(block_group.is_cuda() && block_group.scalar_type() == at::kInt && block_group.is_contiguous() &&
                  block_group.numel() == rows / 128,
              "block_group: int32 [rows / 128]");
  Tensor c = at::empty({rows, N}, a.options());
  dpa::gemm_grouped(a.data_ptr(), a.stride(0), (const long long*)w_ptrs.data_ptr<int64_t>(),
                    w_cols, w_is_kn, c.data_ptr(), N, block_group.data_ptr<int>(), (int)w_ptrs.numel(), (int)rows,
                    (int)N, (int)K, cur_stream());
  return c;
}

// The grouped mode of the dense 256 x 256 weight-gradient kernel takes over from 24 K rows on: equal row shares per
// workgroup whatever the routing did (two of eight experts can hold 90 % of the rows early in training), 676-1014 TF
// from 32 K to 512 K rows against 570-680 TF for the 128 x 128 kernel, whose finer tiles win below
// (benchmarks/grouped_wgrad_bench.py). grouped_wgrad_use_wgrad3(false) / DPA_MOE_WGRAD3=0 turns it off.
bool g_grouped_wgrad3 = true;
bool g_grouped_wgrad3_any_size = false;   // tests: take the 256 x 256 kernel whatever the row count
void grouped_wgrad_use_wgrad3(bool on, bool any_size) {
  g_grouped_wgrad3 = on;
  g_grouped_wgrad3_any_size = on && any_size;
}

// fp32 [G, N, K] = per-segment dy^T x.
Tensor grouped_wgrad(const Tensor& dy, const Tensor& x, const Tensor& seg_start, const Tensor& seg_rows) {
  check_rows2d(dy, "dy");
  check_rows2d(x, "x");
  const int64_t M = dy.size(0), N = dy.size(1), K = x.size(1), G = seg_start.numel();
  TORCH_CHECK(x.size(0) == M && K % 128 == 0 && N % 8 == 0, "grouped_wgrad: unsupported shape");
  TORCH_CHECK(seg_start.is_cuda() && seg_start.scalar_type() == at::kLong && seg_start.is_contiguous() &&
                  seg_rows.is_cuda() && seg_rows.scalar_type() == at::kLong && seg_rows.is_contiguous() &&
                  seg_rows.numel() == G,
              "grouped_wgrad: segment tables are int64 cuda [G]");
  if (dpa::wgrad3_grouped_ok(M, (int)N, (int)K) && g_grouped_wgrad3 && (M >= 24576 || g_grouped_wgrad3_any_size)) {
    // the dense layers' 256 x 256 kernel over equal row shares of the segments, partial tiles added atomically
    Tensor out = at::zeros({G, N, K}, dy.options().dtype(at::kFloat));
    dpa::wgrad3_grouped(dy.data_ptr(), x.data_ptr(), out.data_ptr<float>(), (const long long*)seg_start.data_ptr<int64_t>(),
                        (const long long*)seg_rows.data_ptr<int64_t>(), (int)G, (int)M, (int)N, (int)K, dy.stride(0),
                        x.stride(0), cur_stream());
    return out;
  }
  Tensor out = at::empty({G, N, K}, dy.options().dtype(at::kFloat));
  dpa::gemm_grouped_wgrad(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), out.data_ptr<float>(),
                          (const long long*)seg_start.data_ptr<int64_t>(), (const long long*)seg_rows.data_ptr<int64_t>(),
                          (int)G, (int)N, (int)K, (int)M, cur_stream());
  return out;
}

void check_pairs(const Tensor& row_of_pair) {
  TORCH_CHECK(row_of_pair.is_cuda() && row_of_pair.scalar_type() == at::kLong && row_of_pair.is_contiguous(),
              "row_of_pair: int64 cuda");
}

// A padded expert-sorted buffer [rows, C] whose pair-less rows are zero: with the layout's tables (seg_start, counts,
// seg_rows: int64 [E]) only those rows are written here, the caller's kernel writes the others; without them the
// whole buffer is zero-filled.
static Tensor padded_buffer(int64_t rows, int64_t C, const Tensor& like, const std::optional<Tensor>& seg_start,
                            const std::optional<Tensor>& counts, const std::optional<Tensor>& seg_rows) {
  if (!(seg_start.has_value() && counts.has_value() && seg_rows.has_value()))
    return at::zeros({rows, C}, like.options());
  const int64_t E = seg_start->numel();
  TORCH_CHECK(E > 0 && counts->numel() == E && seg_rows->numel() == E && seg_start->is_cuda() && counts->is_cuda() &&
                  seg_rows->is_cuda() && seg_start->scalar_type() == at::kLong && counts->scalar_type() == at::kLong &&
                  seg_rows->scalar_type() == at::kLong && seg_start->is_contiguous() && counts->is_contiguous() &&
                  seg_rows->is_contiguous(),
              "padded buffer: seg_start / counts / seg_rows are contiguous int64 cuda [E]");
  Tensor dst = at::empty({rows, C}, like.options());
  // padding rows per expert are fewer than the layout's block (<= 256); the tail past the last segment fewer than E blocks
  dpa::moe_zero_padding(dst.data_ptr(), (const long long*)seg_start->data_ptr<int64_t>(),
                        (const long long*)counts->data_ptr<int64_t>(), (const long long*)seg_rows->data_ptr<int64_t>(), (int)E,
                        (int)C, rows, 256, cur_stream());
  return dst;
}

// dst[row_of_pair[j]] = src[j / k]; other rows zero.
Tensor moe_scatter_rows(const Tensor& src, const Tensor& row_of_pair, int64_t k, int64_t rows_out,
                        const std::optional<Tensor>& seg_start, const std::optional<Tensor>& counts,
                        const std::optional<Tensor>& seg_rows) {
  check_rows2d(src, "src");
  check_pairs(row_of_pair);
  TORCH_CHECK(src.is_contiguous() && src.size(1) % 8 == 0 && row_of_pair.numel() == src.size(0) * k);
  Tensor dst = padded_buffer(rows_out, src.size(1), src, seg_start, counts, seg_rows);
  dpa::moe_scatter_rows(src.data_ptr(), (const long long*)row_of_pair.data_ptr<int64_t>(), dst.data_ptr(),
                        row_of_pair.numel(), (int)k, (int)src.size(1), cur_stream());
  return dst;
}

// out[t] = sum_s gates[t, s] * y[row_of_pair[t * k + s]]  (gates absent: plain sum).
Tensor moe_gather_combine(const Tensor& y, const Tensor& row_of_pair, const std::optional<Tensor>& gates, int64_t k) {
  check_rows2d(y, "y");
  check_pairs(row_of_pair);
  TORCH_CHECK(y.is_contiguous() && y.size(1) % 8 == 0 && row_of_pair.numel() % k == 0);
  const int64_t ntok = row_of_pair.numel() / k;
  const void* gp = nullptr;
  if (gates.has_value()) {
    TORCH_CHECK(gates->is_cuda() && gates->scalar_type() == at::kBFloat16 && gates->is_contiguous() &&
                    gates->numel() == ntok * k,
                "gates: bf16 [tokens, k]");
    gp = gates->data_ptr();
  }
  Tensor out = at::empty({ntok, y.size(1)}, y.options());
  dpa::moe_gather_combine(y.data_ptr(), (const long long*)row_of_pair.data_ptr<int64_t>(), gp, out.data_ptr(), ntok,
                          (int)k, (int)y.size(1), cur_stream());
  return out;
}

std::tuple<Tensor, Tensor> moe_combine_bwd(const Tensor& dout, const Tensor& y, const Tensor& row_of_pair,
                                           const Tensor& gates, const std::optional<Tensor>& seg_start,
                                           const std::optional<Tensor>& counts, const std::optional<Tensor>& seg_rows) {
  check_rows2d(dout, "dout");
  check_rows2d(y, "y");
  check_pairs(row_of_pair);
  const int64_t ntok = dout.size(0), C = dout.size(1), npairs = row_of_pair.numel();
  TORCH_CHECK(dout.is_contiguous() && y.is_contiguous() && y.size(1) == C && C % 8 == 0 && ntok > 0 &&
              npairs % ntok == 0);
  TORCH_CHECK(gates.is_cuda() && gates.scalar_type() == at::kBFloat16 && gates.is_contiguous() &&
                  gates.numel() == npairs,
              "gates: bf16 [tokens, k]");
  Tensor dy = padded_buffer(y.size(0), C, y, seg_start, counts, seg_rows);
  Tensor dgate = at::empty({ntok, npairs / ntok}, dout.options().dtype(at::kFloat));
  dpa::moe_combine_bwd(dout.data_ptr(), y.data_ptr(), (const long long*)row_of_pair.data_ptr<int64_t>(),
                       gates.data_ptr(), dy.data_ptr(), dgate.data_ptr<float>(), npairs, (int)(npairs / ntok), (int)C,
                       cur_stream());
  return {dy, dgate};
}

// (idx int64 [N, k], gates fp32 [N, k]) = top-k routing of fp32 logits [N, E] (+ optional selection bias [E]).
std::tuple<Tensor, Tensor> moe_topk_gate(const Tensor& logits, const std::optional<Tensor>& bias, int64_t k) {
  TORCH_CHECK(logits.is_cuda() && logits.scalar_type() == at::kFloat && logits.dim() == 2 && logits.is_contiguous(),
              "moe_topk_gate: fp32 contiguous [tokens, experts]");
  const int64_t N = logits.size(0), E = logits.size(1);
  TORCH_CHECK(E >= 1 && E <= 64 && k >= 1 && k <= 8 && k <= E, "moe_topk_gate: E <= 64 and k <= min(8, E)");
  const float* bp = nullptr;
  if (bias.has_value()) {
    TORCH_CHECK(bias->is_cuda() && bias->scalar_type() == at::kFloat && bias->is_contiguous() && bias->numel() == E,
                "moe_topk_gate: bias is fp32 [experts]");
    bp = bias->data_ptr<float>();
  }
  Tensor idx = at::empty({N, k}, logits.options().dtype(at::kLong));
  Tensor gates = at::empty({N, k}, logits.options());
  dpa::moe_topk_gate(logits.data_ptr<float>(), bp, (long long*)idx.data_ptr<int64_t>(), gates.data_ptr<float>(), N,
                     (int)E, (int)k, cur_stream());
  return {idx, gates};
}

// (row_of_pair [npairs] int64, block_group [rows / block] int32, seg_start, seg_rows, counts [E] int64);
// block = rows per GEMM row block of the padded layout (128 or 256), 0 = dense (unpadded) layout
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> moe_plan(const Tensor& idx, int64_t n_experts, int64_t rows,
                                                            int64_t block) {
  TORCH_CHECK(idx.is_cuda() && idx.scalar_type() == at::kLong && idx.is_contiguous(), "moe_plan: int64 expert ids");
  const int64_t npairs = idx.numel();
  const bool dense = block == 0;     // unpadded layout (expert-sorted send order)
  TORCH_CHECK(n_experts >= 1 && n_experts <= 64, "moe_plan: E <= 64");
  TORCH_CHECK(dense ? rows == npairs
                    : ((block == 128 || block == 256) && rows % block == 0 && rows >= npairs + n_experts * (block - 1)),
              "moe_plan: rows is the pair count (block 0) or a multiple of block (128 | 256) >= pairs + (block - 1) * E");
  auto opt = idx.options();
  Tensor row_of_pair = at::empty({npairs}, opt);
  Tensor block_group = at::empty({dense ? 0 : rows / block}, opt.dtype(at::kInt));
  Tensor seg_start = at::empty({n_experts}, opt), seg_rows = at::empty({n_experts}, opt);
  Tensor counts = at::empty({n_experts}, opt);
  const int nb = dpa::moe_plan_blocks(npairs);
  Tensor hist = at::empty({std::max<int64_t>(nb, 1) * n_experts}, opt.dtype(at::kInt));
  dpa::moe_plan((const long long*)idx.data_ptr<int64_t>(), npairs, (int)n_experts, dense ? 0 : (int)(rows / block),
                dense ? 1 : (int)block, hist.data_ptr<int>(), (long long*)row_of_pair.data_ptr<int64_t>(),
                dense ? nullptr : block_group.data_ptr<int>(), (long long*)seg_start.data_ptr<int64_t>(),
                (long long*)seg_rows.data_ptr<int64_t>(), (long long*)counts.data_ptr<int64_t>(), cur_stream());
  return {row_of_pair, block_group, seg_start, seg_rows, counts};
}

// (fp32 slabs [splits, N, K] of dy^T x, fp32 bias gradient [N]): the column sums of dy come out of the
// weight-gradient kernel itself when the shape runs on the 256 x 256 kernel, else from the column-sum kernel.
std::tuple<Tensor, Tensor> wgrad2_slabs_bias(const Tensor& dy, const Tensor& x) {
  check_rows2d(dy, "dy");
  check_rows2d(x, "x");
  const int64_t M = dy.size(0), N = dy.size(1), K = x.size(1);
  TORCH_CHECK(x.size(0) == M && dpa::wgrad2_ok((int)M, (int)N, (int)K), "wgrad2: unsupported shape ", M, "x", N, "x", K);
  const int splits = dpa::wgrad2_splits((int)M, (int)N, (int)K);
  Tensor slabs = at::empty({splits, N, K}, dy.options().dtype(at::kFloat));
  if (dpa::wgrad2_fuses_bias((int)M, (int)N, (int)K)) {
    Tensor bslabs = at::empty({dpa::wgrad2_bias_slabs((int)M, (int)N, (int)K), N}, dy.options().dtype(at::kFloat));
    dpa::wgrad2_gemm(dy.data_ptr(), x.data_ptr(), slabs.data_ptr<float>(), (int)M, (int)N, (int)K, dy.stride(0),
                     x.stride(0), splits, cur_stream(), bslabs.data_ptr<float>());
    return {slabs, bslabs.sum(0)};
  }
  dpa::wgrad2_gemm(dy.data_ptr(), x.data_ptr(), slabs.data_ptr<float>(), (int)M, (int)N, (int)K, dy.stride(0),
                   x.stride(0), splits, cur_stream());
  return {slabs, colsum(dy)};
}

void scale_unless(Tensor x, const Tensor& g, const Tensor& expected) {
  check_act(x, "x");
  TORCH_CHECK(x.is_contiguous() && x.numel() % 8 == 0, "scale_unless: contiguous tensor of a multiple of 8 elements");
  TORCH_CHECK(g.is_cuda() && g.scalar_type() == at::kFloat && g.numel() == 1, "scale_unless: g is one fp32 value on the device");
  TORCH_CHECK(expected.is_cuda() && expected.scalar_type() == at::kFloat && expected.numel() == 1,
              "scale_unless: expected is one fp32 value on the device");
  dpa::scale_unless(x.data_ptr(), x.numel(), is_bf16(x), g.data_ptr<float>(), expected.data_ptr<float>(), cur_stream());
}

// dw_acc [N, K] += dy^T x and, when given, db_acc [N] += column sums of dy: the weight-gradient GEMM
// and the folds of its per-split slabs, with both results accumulated in place (bf16 or fp32
// destinations, typically views of a flat gradient buffer).
void wgrad2_direct(const Tensor& dy, const Tensor& x, Tensor dw_acc, const std::optional<Tensor>& db_acc) {
  check_rows2d(dy, "dy");
  check_rows2d(x, "x");
  const int64_t M = dy.size(0), N = dy.size(1), K = x.size(1);
  TORCH_CHECK(x.size(0) == M && dpa::wgrad2_ok((int)M, (int)N, (int)K), "wgrad2: unsupported shape ", M, "x", N, "x", K);
  check_act(dw_acc, "dw_acc");
  TORCH_CHECK(dw_acc.numel() == N * K, "wgrad2_direct: dw_acc must hold N*K elements");
  if (db_acc.has_value()) {
    check_act(*db_acc, "db_acc");
    TORCH_CHECK(db_acc->numel() == N && N % 8 == 0, "wgrad2_direct: db_acc must hold N elements (N % 8 == 0)");
  }
  const int splits = dpa::wgrad2_splits((int)M, (int)N, (int)K);
  // one split of the 256 x 256 kernel: it adds into the gradient itself (no slab, no fold)
  // (fp32 destinations only: the read-modify-write of a bf16 destination from the accumulator layout is 2-byte
  // accesses, measured 0.5-1 % slower on the GPT-2-1.3B step than slab + fold)
  const bool direct = dpa::wgrad2_direct_ok((int)M, (int)N, (int)K) && dw_acc.is_contiguous() && !is_bf16(dw_acc);
  Tensor slabs = direct ? Tensor() : at::empty({splits, N, K}, dy.options().dtype(at::kFloat));
  float* slab_ptr = direct ? nullptr : slabs.data_ptr<float>();
  void* dptr = direct ? dw_acc.data_ptr() : nullptr;
  if (db_acc.has_value() && dpa::wgrad2_fuses_bias((int)M, (int)N, (int)K)) {
    const int bs = dpa::wgrad2_bias_slabs((int)M, (int)N, (int)K);
    Tensor bslabs = at::empty({bs, N}, dy.options().dtype(at::kFloat));
    dpa::wgrad2_gemm(dy.data_ptr(), x.data_ptr(), slab_ptr, (int)M, (int)N, (int)K, dy.stride(0),
                     x.stride(0), splits, cur_stream(), bslabs.data_ptr<float>(), dptr, is_bf16(dw_acc));
    dpa::wgrad_finish(bslabs.data_ptr<float>(), db_acc->data_ptr(), N, bs, true, is_bf16(*db_acc), cur_stream());
  } else {
    dpa::wgrad2_gemm(dy.data_ptr(), x.data_ptr(), slab_ptr, (int)M, (int)N, (int)K, dy.stride(0),
                     x.stride(0), splits, cur_stream(), nullptr, dptr, is_bf16(dw_acc));
    if (db_acc.has_value()) {
      Tensor cs = colsum(dy);
      dpa::wgrad_finish(cs.data_ptr<float>(), db_acc->data_ptr(), N, 1, true, is_bf16(*db_acc), cur_stream());
    }
  }
  if (!direct)
    dpa::wgrad_finish(slabs.data_ptr<float>(), dw_acc.data_ptr(), N * K, splits, true, is_bf16(dw_acc), cur_stream());
}

// dst_g (+)= ws[g] for every g: ws fp32 [G, ...], dst_ptrs int64 [G] device addresses of equally sized,
// 16-byte aligned destinations (checked by the caller, which owns the tensors behind the addresses).
void ws_to_many(const Tensor& ws, const Tensor& dst_ptrs, bool accumulate, bool out_bf16) {
  TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.dim() >= 2, "ws_to_many: fp32 [G, ...]");
  TORCH_CHECK(dst_ptrs.is_cuda() && dst_ptrs.scalar_type() == at::kLong && dst_ptrs.is_contiguous() &&
                  dst_ptrs.numel() == ws.size(0),
              "ws_to_many: one destination address per slab");
  const int64_t n = ws.numel() / ws.size(0);
  TORCH_CHECK(n % 8 == 0, "ws_to_many: slab size must be a multiple of 8");
  dpa::wgrad_finish_many(ws.data_ptr<float>(), (const long long*)dst_ptrs.data_ptr<int64_t>(), (int)ws.size(0), n,
                         accumulate, out_bf16, cur_stream());
}

}  // namespace

// ---------------------------------------------------------------------------------------
// c[M, N] = epi(a[M, K] * b[N, K]^T + bias), see kernels.h. a may be a row-strided 2-D view; aux is the
// second operand of the multiply / add epilogues. Returns (c, aux_out) with aux_out = GELU'(v) for the
// GELU epilogue and an empty tensor otherwise.
int gemm_cus() {
  static int cus = 0;
  if (cus == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cus;
}

bool gemm_nt_supported(const Tensor& a, const Tensor& b, const std::optional<Tensor>& bias,
                       const std::optional<Tensor>& aux) {
  if (!a.is_cuda() || !b.is_cuda() || a.dim() != 2 || b.dim() != 2) return false;
  if (a.scalar_type() != at::kBFloat16 || b.scalar_type() != at::kBFloat16) return false;
  if (a.stride(1) != 1 || b.stride(1) != 1 || a.size(1) != b.size(1)) return false;
  if (reinterpret_cast<uintptr_t>(a.data_ptr()) % 16 || reinterpret_cast<uintptr_t>(b.data_ptr()) % 16) return false;
  if (a.size(0) < 1) return false;
  int64_t ld_aux = 0;
  if (aux.has_value()) {
    if (!aux->is_cuda() || aux->scalar_type() != at::kBFloat16 || aux->dim() != 2 || aux->stride(1) != 1 ||
        aux->size(0) != a.size(0) || aux->size(1) != b.size(0) || reinterpret_cast<uintptr_t>(aux->data_ptr()) % 16)
      return false;
    ld_aux = aux->stride(0);
  }
  if (bias.has_value() && (!bias->is_cuda() || bias->scalar_type() != at::kBFloat16 || !bias->is_contiguous() ||
                           bias->numel() != b.size(0)))
    return false;
  return dpa::gemm_nt_ok(a.size(0), b.size(0), a.size(1), a.stride(0), b.stride(0), b.size(0),
                         ld_aux ? ld_aux : b.size(0), bias.has_value());
}

// The two SwiGLU forms (see kernels.h): forward b = [gate rows | up rows] -> (act [M, U], [gate | up] [M, 2U]);
// backward b = W_proj^T [U, K], aux = [gate | up] -> [dgate | dup] [M, 2U].
bool gemm_nt_pair_supported(const Tensor& a, const Tensor& b, int64_t epi, const std::optional<Tensor>& aux) {
  if (!a.is_cuda() || !b.is_cuda() || a.dim() != 2 || b.dim() != 2) return false;
  if (a.scalar_type() != at::kBFloat16 || b.scalar_type() != at::kBFloat16) return false;
  if (a.stride(1) != 1 || b.stride(1) != 1 || a.size(1) != b.size(1) || a.size(0) < 1) return false;
  if (reinterpret_cast<uintptr_t>(a.data_ptr()) % 16 || reinterpret_cast<uintptr_t>(b.data_ptr()) % 16) return false;
  const bool fwd = epi == dpa::kEpiSwiglu;
  if (fwd && b.size(0) % 2 != 0) return false;
  const int64_t U = fwd ? b.size(0) / 2 : b.size(0);
  int64_t ld_pair = 2 * U;
  if (fwd != !aux.has_value()) return false;
  if (aux.has_value()) {
    if (!aux->is_cuda() || aux->scalar_type() != at::kBFloat16 || aux->dim() != 2 || aux->stride(1) != 1 ||
        aux->size(0) != a.size(0) || aux->size(1) != 2 * U || reinterpret_cast<uintptr_t>(aux->data_ptr()) % 16)
      return false;
    ld_pair = aux->stride(0);
  }
  return dpa::gemm_nt_pair_ok(a.size(0), U, a.size(1), a.stride(0), b.stride(0), U, ld_pair) &&
         dpa::gemm_nt_pair_ok(a.size(0), U, a.size(1), a.stride(0), b.stride(0), U, 2 * U);
}

std::tuple<Tensor, std::optional<Tensor>> gemm_nt(const Tensor& a, const Tensor& b, const std::optional<Tensor>& bias, int64_t epi,
                                   const std::optional<Tensor>& aux) {
  TORCH_CHECK(epi >= 0 && epi <= 5, "gemm_nt: epilogue 0..5");
  const bool pair = epi == dpa::kEpiSwiglu || epi == dpa::kEpiDswiglu;
  TORCH_CHECK((epi == dpa::kEpiMul || epi == dpa::kEpiAdd || epi == dpa::kEpiDswiglu) == aux.has_value(),
              "gemm_nt: the multiply / add / dSwiGLU epilogues (and only they) take an auxiliary operand");
  if (pair) {
    TORCH_CHECK(!bias.has_value(), "gemm_nt: the SwiGLU epilogues take no bias");
    TORCH_CHECK(gemm_nt_pair_supported(a, b, epi, aux), "gemm_nt: unsupported operands (see gemm_nt_pair_supported)");
  } else {
    TORCH_CHECK(gemm_nt_supported(a, b, bias, aux), "gemm_nt: unsupported operands (see gemm_nt_supported)");
  }
  const int64_t M = a.size(0), K = a.size(1);
  const int64_t N = epi == dpa::kEpiSwiglu ? b.size(0) / 2 : b.size(0);   // output columns before pairing (U for both SwiGLU forms)
  const int64_t c_cols = epi == dpa::kEpiDswiglu ? 2 * N : N;
  Tensor c = at::empty({M, c_cols}, a.options());
  std::optional<Tensor> aux_out;
  dpa::GemmNtArgs g{};
  g.a = a.data_ptr();
  g.b = b.data_ptr();
  g.c = c.data_ptr();
  g.bias = bias.has_value() ? bias->data_ptr() : nullptr;
  g.M = (int)M;
  g.N = (int)N;
  g.K = (int)K;
  g.lda = a.stride(0);
  g.ldb = b.stride(0);
  g.ldc = c_cols;
  g.ld_aux = N;
  if (epi == dpa::kEpiGelu) {
    aux_out = at::empty({M, N}, a.options());
    g.aux_out = aux_out->data_ptr();
  } else if (epi == dpa::kEpiSwiglu) {
    aux_out = at::empty({M, 2 * N}, a.options());
    g.aux_out = aux_out->data_ptr();
    g.ld_aux = 2 * N;
  } else if (aux.has_value()) {
    g.aux_in = aux->data_ptr();
    g.ld_aux = aux->stride(0);
  }
  dpa::gemm_nt(g, (int)epi, gemm_cus(), cur_stream());
  return {c, aux_out};
}

// c[M, N] = rope(a b^T + bias): the fused qkv projection whose first rope_cols columns (the q and k heads, each hs
// wide) leave the GEMM already rotated; row r is position pos0 + r % seq_len.
Tensor gemm_nt_rope(const Tensor& a, const Tensor& b, const std::optional<Tensor>& bias, const Tensor& rope_cos,
                    const Tensor& rope_sin, int64_t pos0, int64_t seq_len, int64_t rope_cols, int64_t hs) {
  TORCH_CHECK(gemm_nt_supported(a, b, bias, std::nullopt), "gemm_nt_rope: unsupported operands (see gemm_nt_supported)");
  const int64_t M = a.size(0), K = a.size(1), N = b.size(0);
  TORCH_CHECK((hs == 64 || hs == 128) && rope_cols % hs == 0 && rope_cols >= 0 && rope_cols <= N && N % hs == 0,
              "gemm_nt_rope: heads of 64 or 128 columns, rope_cols a multiple of the head size within N");
  TORCH_CHECK(seq_len > 0 && pos0 >= 0, "gemm_nt_rope: seq_len > 0, pos0 >= 0");
  check_rope_tables(rope_cos, rope_sin, hs, pos0 + std::min<int64_t>(seq_len, M), "gemm_nt_rope");
  Tensor c = at::empty({M, N}, a.options());
  dpa::GemmNtArgs g{};
  g.a = a.data_ptr();
  g.b = b.data_ptr();
  g.c = c.data_ptr();
  g.bias = bias.has_value() ? bias->data_ptr() : nullptr;
  g.M = (int)M;
  g.N = (int)N;
  g.K = (int)K;
  g.lda = a.stride(0);
  g.ldb = b.stride(0);
  g.ldc = N;
  g.ld_aux = N;
  g.rope_cos = rope_cos.data_ptr<float>();
  g.rope_sin = rope_sin.data_ptr<float>();
  g.rope_cols = (int)rope_cols;
  g.rope_hs = (int)hs;
  g.rope_T = (int)seq_len;
  g.rope_pos0 = (int)pos0;
  TORCH_CHECK(dpa::gemm_nt_rows_per_launch(g, dpa::kEpiRope) > 0, "gemm_nt_rope: rows too wide for one aligned row range");
  dpa::gemm_nt(g, dpa::kEpiRope, gemm_cus(), cur_stream());
  return c;
}

// Decode-step linear layer (csrc/decode.hip): y [B, N] = act(LN(x) w^T + bias) + residual, B <= 16. With the KV caches
// given (contiguous [B, S, heads, hs], row width kv_cols = heads * hs), the output columns from kv_col0 on are the
// new key and value rows and go to row `pos` of the caches; y then holds only the columns below kv_col0.
bool decode_linear_supported(const Tensor& x, const Tensor& w, int64_t act, bool has_ln) {
  if (!x.is_cuda() || !w.is_cuda() || x.dim() != 2 || w.dim() != 2) return false;
  if (x.scalar_type() != at::kBFloat16 || w.scalar_type() != at::kBFloat16) return false;
  if (x.stride(1) != 1 || w.stride(1) != 1 || x.size(1) != w.size(1)) return false;
  if (reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 || reinterpret_cast<uintptr_t>(w.data_ptr()) % 16) return false;
  const bool pair = act == dpa::kDecodeSwiglu;
  if (pair && w.size(0) % 2) return false;
  const int64_t N = pair ? w.size(0) / 2 : w.size(0);
  return dpa::decode_linear_ok(x.size(0), N, x.size(1), x.stride(0), w.stride(0), pair, has_ln);
}

Tensor decode_linear(const Tensor& x, const std::optional<Tensor>& ln_w, const std::optional<Tensor>& ln_b, double eps,
                     const Tensor& w, const std::optional<Tensor>& bias, const std::optional<Tensor>& residual, int64_t act,
                     const std::optional<Tensor>& k_cache, const std::optional<Tensor>& v_cache,
                     const std::optional<Tensor>& pos, int64_t kv_col0, const std::optional<Tensor>& rope_cos,
                     const std::optional<Tensor>& rope_sin, int64_t rope_cols, int64_t rope_hs) {
  TORCH_CHECK(act >= 0 && act <= 2, "decode_linear: act 0 (none), 1 (GELU), 2 (SwiGLU)");
  TORCH_CHECK(decode_linear_supported(x, w, act, ln_w.has_value()),
              "decode_linear: unsupported operands (see decode_linear_supported)");
  const bool pair = act == dpa::kDecodeSwiglu;
  const int64_t B = x.size(0), K = x.size(1), N = pair ? w.size(0) / 2 : w.size(0);
  auto vec_ok = [&](const std::optional<Tensor>& t, int64_t n) {
    return !t.has_value() || (t->is_cuda() && t->scalar_type() == at::kBFloat16 && t->is_contiguous() && t->numel() == n &&
                              reinterpret_cast<uintptr_t>(t->data_ptr()) % 16 == 0);
  };
  TORCH_CHECK(vec_ok(ln_w, K) && vec_ok(ln_b, K) && (!ln_b.has_value() || ln_w.has_value()),
              "decode_linear: LayerNorm weight / bias are contiguous bf16 [K]");
  TORCH_CHECK(vec_ok(bias, w.size(0)), "decode_linear: bias is contiguous bf16 [rows of w]");
  dpa::DecodeLinearArgs p{};
  p.x = x.data_ptr();
  p.ldx = x.stride(0);
  p.ln_w = ln_w.has_value() ? ln_w->data_ptr() : nullptr;
  p.ln_b = ln_b.has_value() ? ln_b->data_ptr() : nullptr;
  p.eps = (float)eps;
  p.w = w.data_ptr();
  p.ldw = w.stride(0);
  p.bias = bias.has_value() ? bias->data_ptr() : nullptr;
  p.B = (int)B;
  p.N = (int)N;
  p.K = (int)K;
  p.act = (int)act;
  int64_t y_cols = N;
  if (k_cache.has_value()) {
    TORCH_CHECK(v_cache.has_value() && pos.has_value() && !pair, "decode_linear: k_cache, v_cache and pos come together");
    TORCH_CHECK(k_cache->is_cuda() && k_cache->scalar_type() == at::kBFloat16 && k_cache->is_contiguous() &&
                    k_cache->dim() == 4 && v_cache->is_cuda() && v_cache->scalar_type() == at::kBFloat16 &&
                    v_cache->is_contiguous() && v_cache->sizes() == k_cache->sizes() && k_cache->size(0) == B,
                "decode_linear: caches are contiguous bf16 [B, S, heads, hs]");
    const int64_t kv_cols = k_cache->size(2) * k_cache->size(3);
    TORCH_CHECK(kv_col0 >= 0 && kv_col0 % 16 == 0 && kv_cols % 16 == 0 && kv_col0 + 2 * kv_cols == N,
                "decode_linear: the projection is [q | k | v] with k and v as wide as a cache row");
    TORCH_CHECK(pos->is_cuda() && pos->scalar_type() == at::kLong && pos->numel() == 1, "decode_linear: pos is int64 [1] on the GPU");
    p.k_cache = k_cache->data_ptr();
    p.v_cache = v_cache->data_ptr();
    p.cache_bs = k_cache->stride(0);
    p.pos = (const long long*)pos->data_ptr<int64_t>();
    p.kv_col0 = (int)kv_col0;
    p.kv_cols = (int)kv_cols;
    y_cols = kv_col0;
  }
  Tensor y = at::empty({B, std::max<int64_t>(y_cols, 1)}, x.options());
  if (residual.has_value()) {
    TORCH_CHECK(residual->is_cuda() && residual->scalar_type() == at::kBFloat16 && residual->dim() == 2 &&
                    residual->size(0) == B && residual->size(1) == N && residual->stride(1) == 1 &&
                    residual->stride(0) % 4 == 0 && reinterpret_cast<uintptr_t>(residual->data_ptr()) % 8 == 0 &&
                    !k_cache.has_value(),
                "decode_linear: residual is bf16 [B, N] with 8-byte aligned rows");
    p.residual = residual->data_ptr();
    p.ldr = residual->stride(0);
  }
  if (rope_cos.has_value()) {
    auto row_ok = [&](const Tensor& t) {
      return t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.numel() == rope_hs / 2 &&
             reinterpret_cast<uintptr_t>(t.data_ptr()) % 8 == 0;
    };
    TORCH_CHECK(rope_sin.has_value() && row_ok(*rope_cos) && row_ok(*rope_sin),
                "decode_linear: rope_cos / rope_sin are the fp32 [hs / 2] rows of the token's position");
    TORCH_CHECK(act == dpa::kDecodeNone && !residual.has_value() && rope_hs >= 8 && rope_hs % 8 == 0 && rope_cols > 0 &&
                    rope_cols % rope_hs == 0 && rope_cols <= N,
                "decode_linear: the rotary epilogue takes whole heads of a plain projection");
    p.rope_cos = rope_cos->data_ptr<float>();
    p.rope_sin = rope_sin->data_ptr<float>();
    p.rope_cols = (int)rope_cols;
    p.rope_hs = (int)rope_hs;
  }
  p.y = y.data_ptr();
  p.ldy = y.stride(0);
  dpa::decode_linear(p, cur_stream());
  return y;
}

// Grouped form of the NT GEMM (csrc/gemm_nt.hip): c[rows, N] = epi(a[rows, K] * W_g^T) with W_g = the [N, K] matrix
// at address w_ptrs[block_group[row / 256]]. epi: 0 none, 1 GELU (also returns GELU'), 2 multiply by aux [rows, N],
// 4 SwiGLU (W_g = [2N, K], gate rows then up rows: returns silu(gate) * up [rows, N] and [gate | up] [rows, 2N]),
// 5 dSwiGLU (aux = [gate | up] [rows, 2N]: returns [dgate | dup] [rows, 2N] from d(act) = a W_g^T).
std::tuple<Tensor, std::optional<Tensor>> gemm_nt_grouped(const Tensor& a, const Tensor& w_ptrs, int64_t N, int64_t ldw,
                                                         int64_t epi, const std::optional<Tensor>& aux,
                                                         const Tensor& block_group) {
  check_rows2d(a, "a");
  const int64_t rows = a.size(0), K = a.size(1);
  const bool pair = epi == dpa::kEpiSwiglu || epi == dpa::kEpiDswiglu;
  TORCH_CHECK(epi == dpa::kEpiNone || epi == dpa::kEpiGelu || epi == dpa::kEpiMul || pair,
              "gemm_nt_grouped: epilogue 0, 1, 2, 4 or 5");
  TORCH_CHECK((epi == dpa::kEpiMul || epi == dpa::kEpiDswiglu) == aux.has_value(),
              "gemm_nt_grouped: the multiply and dSwiGLU epilogues (and only they) take aux");
  TORCH_CHECK(rows % 256 == 0 && rows > 0, "gemm_nt_grouped: rows must be padded to a multiple of 256");
  TORCH_CHECK(w_ptrs.is_cuda() && w_ptrs.scalar_type() == at::kLong && w_ptrs.is_contiguous() && w_ptrs.numel() >= 1,
              "w_ptrs: int64 cuda");
  TORCH_CHECK(block_group.is_cuda() && block_group.scalar_type() == at::kInt && block_group.is_contiguous() &&
                  block_group.numel() == rows / 256,
              "block_group: int32 [rows / 256]");
  const int64_t aux_cols = pair ? 2 * N : N;
  int64_t ld_aux = aux_cols;
  if (aux.has_value()) {
    check_rows2d(*aux, "aux");
    TORCH_CHECK(aux->size(0) == rows && aux->size(1) == aux_cols, "gemm_nt_grouped: aux is [rows, N] ([rows, 2N] for dSwiGLU)");
    ld_aux = aux->stride(0);
  }
  const int64_t c_cols = epi == dpa::kEpiDswiglu ? 2 * N : N;
  if (pair) {
    TORCH_CHECK(dpa::gemm_nt_pair_ok(rows, N, K, a.stride(0), ldw, N, ld_aux) &&
                    dpa::gemm_nt_pair_ok(rows, N, K, a.stride(0), ldw, N, 2 * N),
                "gemm_nt_grouped: unsupported SwiGLU shape ", rows, "x", N, "x", K);
  } else {
    TORCH_CHECK(dpa::gemm_nt_ok(rows, N, K, a.stride(0), ldw, N, ld_aux, false), "gemm_nt_grouped: unsupported shape ", rows,
                "x", N, "x", K);
  }
  Tensor c = at::empty({rows, c_cols}, a.options());
  std::optional<Tensor> aux_out;
  dpa::GemmNtArgs g{};
  g.a = a.data_ptr();
  g.b = nullptr;
  g.c = c.data_ptr();
  g.M = (int)rows;
  g.N = (int)N;
  g.K = (int)K;
  g.lda = a.stride(0);
  g.ldb = ldw;
  g.ldc = c_cols;
  g.ld_aux = ld_aux;
  if (epi == dpa::kEpiGelu || epi == dpa::kEpiSwiglu) {
    aux_out = at::empty({rows, aux_cols}, a.options());
    g.aux_out = aux_out->data_ptr();
  } else if (aux.has_value()) {
    g.aux_in = aux->data_ptr();
  }
  g.w_ptrs = (const long long*)w_ptrs.data_ptr<int64_t>();
  g.block_group = block_group.data_ptr<int>();
  g.ngroups = (int)w_ptrs.numel();
  dpa::gemm_nt(g, (int)epi, gemm_cus(), cur_stream());
  return {c, aux_out};
}

// out[M, N] (rows of a larger buffer allowed: row stride >= N) = a[M, K] * b[N, K]^T, no epilogue.
void gemm_nt_into(const Tensor& a, const Tensor& b, Tensor out) {
  TORCH_CHECK(gemm_nt_supported(a, b, std::nullopt, std::nullopt), "gemm_nt_into: unsupported operands");
  const int64_t M = a.size(0), N = b.size(0), K = a.size(1);
  TORCH_CHECK(out.is_cuda() && out.scalar_type() == at::kBFloat16 && out.dim() == 2 && out.size(0) == M && out.size(1) == N &&
                  out.stride(1) == 1 && reinterpret_cast<uintptr_t>(out.data_ptr()) % 16 == 0,
              "gemm_nt_into: out is bf16 [M, N] with unit inner stride");
  TORCH_CHECK(dpa::gemm_nt_ok(M, N, K, a.stride(0), b.stride(0), out.stride(0), 0, false), "gemm_nt_into: unsupported shape");
  dpa::GemmNtArgs g{};
  g.a = a.data_ptr();
  g.b = b.data_ptr();
  g.c = out.data_ptr();
  g.M = (int)M;
  g.N = (int)N;
  g.K = (int)K;
  g.lda = a.stride(0);
  g.ldb = b.stride(0);
  g.ldc = out.stride(0);
  g.ld_aux = N;
  dpa::gemm_nt(g, dpa::kEpiNone, gemm_cus(), cur_stream());
}

Tensor transpose_bf16(const Tensor& x) {
  TORCH_CHECK(x.is_cuda() && x.dim() == 2 && x.scalar_type() == at::kBFloat16 && x.is_contiguous() &&
                  x.size(0) % 8 == 0 && x.size(1) % 8 == 0 && reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0,
              "transpose_bf16: contiguous bf16 [R, C] with R, C multiples of 8");
  Tensor out = at::empty({x.size(1), x.size(0)}, x.options());
  dpa::transpose_bf16(x.data_ptr(), out.data_ptr(), (int)x.size(0), (int)x.size(1), cur_stream());
  return out;
}

// Equally shaped contiguous bf16 matrices [R, C] at the addresses in `in_ptrs` -> their transposes at `out_ptrs`
// (int64 device tensors; the caller vouches for the addresses: 16-byte aligned, R x C bf16 behind each).
void transpose_bf16_many(const Tensor& in_ptrs, const Tensor& out_ptrs, int64_t R, int64_t C) {
  TORCH_CHECK(in_ptrs.is_cuda() && out_ptrs.is_cuda() && in_ptrs.scalar_type() == at::kLong &&
                  out_ptrs.scalar_type() == at::kLong && in_ptrs.dim() == 1 && in_ptrs.is_contiguous() &&
                  out_ptrs.is_contiguous() && in_ptrs.numel() == out_ptrs.numel(),
              "transpose_bf16_many: two int64 device vectors of equal length");
  TORCH_CHECK(R > 0 && C > 0 && R % 8 == 0 && C % 8 == 0 && in_ptrs.numel() <= 65535,
              "transpose_bf16_many: R, C multiples of 8, at most 65535 matrices");
  dpa::transpose_bf16_many(reinterpret_cast<const long long*>(in_ptrs.data_ptr<int64_t>()),
                           reinterpret_cast<const long long*>(out_ptrs.data_ptr<int64_t>()), (int)in_ptrs.numel(), (int)R,
                           (int)C, cur_stream());
}

TORCH_LIBRARY(dpa, m) {
  m.def("gemm_nt(Tensor a, Tensor b, Tensor? bias, int epi, Tensor? aux) -> (Tensor, Tensor?)");
  m.def("gemm_nt_supported(Tensor a, Tensor b, Tensor? bias, Tensor? aux) -> bool");
  m.def("gemm_nt_into(Tensor a, Tensor b, Tensor(a!) out) -> ()");
  m.def("gemm_nt_grouped(Tensor a, Tensor w_ptrs, int n, int ldw, int epi, Tensor? aux, Tensor block_group) -> (Tensor, Tensor?)");
  m.def("gemm_nt_pair_supported(Tensor a, Tensor b, int epi, Tensor? aux) -> bool");
  m.def("gemm_nt_set_variant(int v) -> ()");
  m.def("grouped_wgrad_use_wgrad3(bool on, bool any_size=False) -> ()");
  m.def("transpose_bf16(Tensor x) -> Tensor");
  m.def("transpose_bf16_many(Tensor in_ptrs, Tensor out_ptrs, int R, int C) -> ()");
  m.def("ws_to_many(Tensor ws, Tensor dst_ptrs, bool accumulate, bool out_bf16) -> ()");
  m.def("wgrad2_slabs_bias(Tensor dy, Tensor x) -> (Tensor, Tensor)");
  m.def("wgrad2_direct(Tensor dy, Tensor x, Tensor(a!) dw_acc, Tensor(b!)? db_acc) -> ()");
  m.def("moe_topk_gate(Tensor logits, Tensor? bias, int k) -> (Tensor, Tensor)");
  m.def("moe_plan(Tensor idx, int n_experts, int rows, int block) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("grouped_gemm(Tensor a, Tensor w_ptrs, int w_rows, int w_cols, bool w_is_kn, Tensor block_group) -> Tensor");
  m.def("grouped_wgrad(Tensor dy, Tensor x, Tensor seg_start, Tensor seg_rows) -> Tensor");
  m.def("moe_scatter_rows(Tensor src, Tensor row_of_pair, int k, int rows_out, Tensor? seg_start=None, Tensor? counts=None, Tensor? seg_rows=None) -> Tensor");
  m.def("moe_gather_combine(Tensor y, Tensor row_of_pair, Tensor? gates, int k) -> Tensor");
  m.def("moe_combine_bwd(Tensor dout, Tensor y, Tensor row_of_pair, Tensor gates, Tensor? seg_start=None, Tensor? counts=None, Tensor? seg_rows=None) -> (Tensor, Tensor)");
  m.def("colsum(Tensor x) -> Tensor");
  m.def("wgrad_into(Tensor dy, Tensor x, Tensor(a!) ws, Tensor(b!)? dbias) -> ()");
  m.def("ws_to(Tensor ws, Tensor(a!) out, bool accumulate) -> ()");
  m.def("wgrad2(Tensor dy, Tensor x) -> Tensor");
  m.def("wgrad2_set_stages(int s) -> ()");
  m.def("scale_unless(Tensor(a!) x, Tensor g, Tensor expected) -> ()");
  m.def("wgrad2_force_splits(int s) -> ()");
  m.def("clear_last_error() -> int");
  m.def("wgrad2_slabs(Tensor dy, Tensor x) -> Tensor");
  m.def("wgrad2_accum(Tensor dy, Tensor x, Tensor(a!) ws) -> ()");
  m.def("layernorm_fwd(Tensor x, Tensor? res, Tensor w, Tensor b, float eps) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("layernorm_bwd(Tensor dy, Tensor x, Tensor w, Tensor mean, Tensor rstd, Tensor? dres) -> (Tensor, Tensor, Tensor)");
  m.def("layernorm_bwd_acc(Tensor dy, Tensor x, Tensor w, Tensor mean, Tensor rstd, Tensor? dres, Tensor(a!) dw_acc, "
        "Tensor(b!) db_acc) -> Tensor");
  m.def("adamw_step(Tensor(a!) master, Tensor(b!) m, Tensor(c!) v, Tensor grad, Tensor? param_out, int[] seg_ends, "
        "float[] seg_wds, float lr, float beta1, float beta2, float eps, int step, Tensor? norm2, float max_norm, "
        "float grad_mul, Tensor? dyn) -> ()");
  m.def("sumsq(Tensor g, Tensor(a!) out, bool accumulate) -> ()");
  m.def("xent_fused(Tensor(a!) logits, Tensor targets, Tensor grad_scale, int ignore_index, bool write_grad) -> Tensor");
  m.def("xent_stats(Tensor logits, Tensor targets, int vocab_start, int ignore_index) -> (Tensor, Tensor, Tensor)");
  m.def("xent_grad(Tensor(a!) logits, Tensor targets, Tensor lse, Tensor tgt_logit, Tensor grad_scale, "
        "int vocab_start, int ignore_index, bool write_grad) -> Tensor");
  m.def("embed_fwd(Tensor idx, Tensor wte, Tensor? pos, int pos0) -> Tensor");
  m.def("embed_bwd(Tensor dout, Tensor ids_sorted, Tensor perm, Tensor(a!) dwte) -> ()");
  m.def("pos_bwd(Tensor dout, Tensor(a!) dpos, int pos0) -> ()");
  m.def("gelu_fwd(Tensor x) -> Tensor");
  m.def("gelu_bwd(Tensor dy, Tensor x) -> Tensor");
  m.def("kv_append(Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor k_new, Tensor v_new, Tensor pos) -> ()");
  m.def("sample_topk(Tensor logits, float temperature, int topk, Tensor seed) -> Tensor");
  m.def("sample_topk_step(Tensor logits, float temperature, int topk, Tensor seed, Tensor(a!) tok, Tensor(b!) history, "
        "Tensor(c!) history_pos, Tensor(d!) pos, Tensor(e!) len_after, Tensor(f!) ticket) -> ()");
  m.def("spin(int blocks, int usec, Tensor(a!) sink) -> ()");
  m.def("act_fwd(Tensor x, int kind) -> Tensor");
  m.def("act_bwd(Tensor dy, Tensor x, int kind) -> Tensor");
  m.def("swiglu_fwd(Tensor h) -> Tensor");
  m.def("swiglu_bwd(Tensor dy, Tensor h) -> Tensor");
  m.def("attn_fwd(Tensor q, Tensor k, Tensor v, float scale, float p_drop=0.0, Tensor? seed=None, Tensor? kv_len=None) -> (Tensor, Tensor)");
  m.def("attn_decode(Tensor q, Tensor k, Tensor v, float scale, Tensor kv_len) -> Tensor");
  m.def("attn_dropout_mask(int B, int nh, int Tq, int Tk, float p_drop, Tensor seed) -> Tensor");
  m.def("attn_bwd(Tensor dout, Tensor q, Tensor k, Tensor v, Tensor o, Tensor lse, Tensor(a!) dq, Tensor(b!) dk, "
        "Tensor(c!) dv, float scale, float p_drop=0.0, Tensor? seed=None, Tensor? rope_cos=None, Tensor? rope_sin=None, "
        "int rope_pos0=0) -> ()");
  m.def("decode_linear(Tensor x, Tensor? ln_w, Tensor? ln_b, float eps, Tensor w, Tensor? bias, Tensor? residual, int act, "
        "Tensor(a!)? k_cache, Tensor(b!)? v_cache, Tensor? pos, int kv_col0, Tensor? rope_cos=None, Tensor? rope_sin=None, "
        "int rope_cols=0, int rope_hs=0) -> Tensor");
  m.def("decode_linear_supported(Tensor x, Tensor w, int act, bool has_ln) -> bool");
  m.def("gemm_nt_rope(Tensor a, Tensor b, Tensor? bias, Tensor rope_cos, Tensor rope_sin, int pos0, int seq_len, "
        "int rope_cols, int hs) -> Tensor");
  m.def("rope_(Tensor(a!) x, Tensor cos_t, Tensor sin_t, int pos0, bool inverse) -> ()");
}

TORCH_LIBRARY_IMPL(dpa, CompositeExplicitAutograd, m) {
  m.impl("wgrad2_set_stages", [](int64_t s) { dpa::wgrad2_set_stages((int)s); });
  m.impl("scale_unless", scale_unless);
  m.impl("wgrad2_force_splits", [](int64_t s) { dpa::wgrad2_force_splits((int)s); });
  // Reads and clears the runtime's sticky "last error" of this thread (a stream capture that was invalidated leaves
  // one behind, and the next launch's error check would report it as its own). Returns the error code read.
  m.impl("clear_last_error", []() -> int64_t { return (int64_t)hipGetLastError(); });
  m.impl("gemm_nt_set_variant", [](int64_t v) { dpa::gemm_nt_set_variant((int)v); });
  m.impl("grouped_wgrad_use_wgrad3", [](bool on, bool any_size) { grouped_wgrad_use_wgrad3(on, any_size); });
  m.impl("gemm_nt_supported", gemm_nt_supported);
  m.impl("decode_linear_supported", decode_linear_supported);
  m.impl("gemm_nt_pair_supported", gemm_nt_pair_supported);
}

TORCH_LIBRARY_IMPL(dpa, CUDA, m) {
  m.impl("ws_to_many", ws_to_many);
  m.impl("gemm_nt", gemm_nt);
  m.impl("gemm_nt_rope", gemm_nt_rope);
  m.impl("decode_linear", decode_linear);
  m.impl("gemm_nt_into", gemm_nt_into);
  m.impl("gemm_nt_grouped", gemm_nt_grouped);
  m.impl("transpose_bf16", transpose_bf16);
  m.impl("transpose_bf16_many", transpose_bf16_many);
  m.impl("wgrad2_slabs_bias", wgrad2_slabs_bias);
  m.impl("wgrad2_direct", wgrad2_direct);
  m.impl("moe_topk_gate", moe_topk_gate);
  m.impl("moe_plan", moe_plan);
  m.impl("grouped_gemm", grouped_gemm);
  m.impl("grouped_wgrad", grouped_wgrad);
  m.impl("moe_scatter_rows", moe_scatter_rows);
  m.impl("moe_gather_combine", moe_gather_combine);
  m.impl("moe_combine_bwd", moe_combine_bwd);
  m.impl("colsum", colsum);
  m.impl("wgrad_into", wgrad_into);
  m.impl("ws_to", ws_to);
  m.impl("wgrad2", wgrad2);
  m.impl("wgrad2_slabs", wgrad2_slabs);
  m.impl("wgrad2_accum", wgrad2_accum);
  m.impl("layernorm_fwd", layernorm_fwd);
  m.impl("layernorm_bwd", layernorm_bwd);
  m.impl("layernorm_bwd_acc", layernorm_bwd_acc);
  m.impl("adamw_step", adamw_step);
  m.impl("sumsq", sumsq);
  m.impl("xent_fused", xent_fused);
  m.impl("xent_stats", xent_stats);
  m.impl("xent_grad", xent_grad);
  m.impl("embed_fwd", embed_fwd);
  m.impl("embed_bwd", embed_bwd);
  m.impl("pos_bwd", pos_bwd);
  m.impl("gelu_fwd", gelu_fwd);
  m.impl("gelu_bwd", gelu_bwd);
  m.impl("kv_append", kv_append);
  m.impl("sample_topk", sample_topk);
  m.impl("sample_topk_step", sample_topk_step);
  m.impl("spin", spin);
  m.impl("act_fwd", act_fwd);
  m.impl("act_bwd", act_bwd);
  m.impl("swiglu_fwd", swiglu_fwd);
  m.impl("swiglu_bwd", swiglu_bwd);
  m.impl("attn_fwd", attn_fwd);
  m.impl("attn_dropout_mask", attn_dropout_mask);
  m.impl("attn_bwd", attn_bwd);
  m.impl("attn_decode", attn_decode);
  m.impl("rope_", rope_);
}
