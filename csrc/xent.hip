// Softmax cross-entropy over a [rows, V] logits block for gfx950, with the gradient written
// in place of the logits (the lm_head backward GEMMs consume it directly, so the fp32
// probability matrix never exists).
//
// One 1024-thread workgroup owns one row; a row of up to 65536 logits is held in registers
// between the statistics pass and the gradient pass, so the logits are read from memory once.
//
// Three modes share the code:
//   FUSED  : statistics + loss + in-place gradient (single-GPU / replicated vocab)
//   STATS  : per-row (max, sum exp(x - max), target logit) for a vocab shard
//   GRAD   : loss/gradient from a given global log-sum-exp (after the cross-rank combine)
// STATS + GRAD are the two halves of vocab-parallel cross-entropy: between them the caller
// all-reduces max / sum-exp / target-logit across the vocab shards.
//
// Replaces `F.cross_entropy(..., ignore_index=-1)` (reference: single-gpu/model.py:689).
#include "common.h"
#include "kernels.h"

namespace dpa {

enum { XENT_FUSED = 0, XENT_STATS = 1, XENT_GRAD = 2 };

constexpr int kXentThreads = 1024;
constexpr int kXentMaxVec = 8;  // vectors of 8 logits cached per thread -> V <= 65536 in registers

__device__ __forceinline__ float exp_diff(float d) { return d != d ? 0.f : __expf(d); }  // (-inf) - (-inf) -> 0

// The per-element work is counted in instructions: at the HBM rate a CU has about 25 vector instructions per element
// (read + write) to spend and the three passes over a row used 18 of them. exp(x - r) is one fused multiply-add and
// one v_exp_f32 (r pre-multiplied by log2(e)); the one-hot subtraction is a single store by the thread that owns
// the target's vector, after the row's gradient has been written, instead of a compare and a select per element.
#ifndef DPA_XENT_NT
#define DPA_XENT_NT 3   // bit 0 loads, bit 1 stores: the row is read once and written once: non-temporal accesses (nothing here is reused from a cache)
#endif
template <typename T>
__device__ __forceinline__ void row_load(Vec8<T>& c, const T* p) { c.template loads<(DPA_XENT_NT & 1) != 0>(p); }
template <typename T>
__device__ __forceinline__ void row_store(T* p, const float (&o)[8]) { store8s<(DPA_XENT_NT & 2) != 0>(p, o); }
constexpr float kL2e = 1.4426950408889634f;
__device__ __forceinline__ float exp2_scaled(float x, float neg_ref_l2e) {
  return __builtin_amdgcn_exp2f(fmaf(x, kL2e, neg_ref_l2e));
}

// Block-wide log-sum-exp combine of per-thread (max, sum of exp(x - max)) pairs: one LDS round.
__device__ __forceinline__ void block_lse(float& m, float& s, float* red /* 32 floats */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
    const float M = fmaxf(m, m2);
    s = s * exp_diff(m - M) + s2 * exp_diff(m2 - M);
    m = M;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) {
    red[wid] = m;
    red[16 + wid] = s;
  }
  __syncthreads();
  float M = red[0];
  for (int i = 1; i < nw; ++i) M = fmaxf(M, red[i]);
  float S = 0.f;
  for (int i = 0; i < nw; ++i) S += red[16 + i] * exp_diff(red[i] - M);
  m = M;
  s = S;
}

template <typename T, int MODE, bool CACHED>
__global__ __launch_bounds__(kXentThreads, sizeof(T) == 2 ? 8 : 4) void xent_kernel(T* __restrict__ logits, const long long* __restrict__ targets,
                                                            float* __restrict__ loss_rows, float* __restrict__ row_max,
                                                            float* __restrict__ row_sum, float* __restrict__ row_tgt,
                                                            const float* __restrict__ lse_in,
                                                            const float* __restrict__ grad_scale, int V, int ld,
                                                            int vocab_start, long long ignore_index, int write_grad) {
  __shared__ float red[32];
  const int row = blockIdx.x;
  T* lr = logits + (size_t)row * ld;
  const long long tgt = targets[row];
  const bool ignored = (tgt == ignore_index);
  const int nvec = V >> 3;  // V is a multiple of 8 (checked on the host)
  const int tloc = (int)(tgt - vocab_start);

  Vec8<T> cache[CACHED ? kXentMaxVec : 1];
  float mx = -INFINITY;
  float sum = 0.f, lse = 0.f;

  if (MODE != XENT_GRAD) {
    if (ignored && MODE == XENT_FUSED) {
      // Ignored rows contribute no loss and a zero gradient.
      if (threadIdx.x == 0) loss_rows[row] = 0.f;
      if (write_grad) {
        float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = threadIdx.x; i < nvec; i += kXentThreads) store8(lr + i * 8, z);
      }
      return;
    }
    if (CACHED) {
      // thread-local max and sum first, then a single block-wide combine
#pragma unroll
      for (int k = 0; k < kXentMaxVec; ++k) {
        const int i = threadIdx.x + k * kXentThreads;
        if (i < nvec) {
          row_load(cache[k], lr + i * 8);
          float x[8];
          cache[k].get(x);
#pragma unroll
          for (int j = 0; j < 8; j += 2) mx = fmaxf(fmaxf(mx, x[j]), x[j + 1]);   // v_max3_f32
        }
      }
      const float nm = -mx * kL2e;
#pragma unroll
      for (int k = 0; k < kXentMaxVec; ++k) {
        const int i = threadIdx.x + k * kXentThreads;
        if (i < nvec) {
          float x[8];
          cache[k].get(x);
#pragma unroll
          for (int j = 0; j < 8; ++j) sum += exp2_scaled(x[j], nm);
        }
      }
      block_lse(mx, sum, red);
    } else {
      for (int i = threadIdx.x; i < nvec; i += kXentThreads) {
        float x[8];
        load8(lr + i * 8, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, x[j]);
      }
      mx = block_max(mx, red);
      const float nm = -mx * kL2e;
      for (int i = threadIdx.x; i < nvec; i += kXentThreads) {
        float x[8];
        load8(lr + i * 8, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += exp2_scaled(x[j], nm);
      }
      sum = block_sum(sum, red);
    }
    if (MODE == XENT_STATS) {
      if (threadIdx.x == 0) {
        row_max[row] = mx;
        row_sum[row] = sum;
        float t = 0.f;
        if (!ignored && tloc >= 0 && tloc < V) t = Cvt<T>::ld(lr + tloc);
        row_tgt[row] = t;
      }
      return;
    }
    lse = mx + __logf(sum);
    // a target outside this shard's columns (corrupt label) reads as logit 0 instead of out of bounds
    if (threadIdx.x == 0) loss_rows[row] = lse - ((tloc >= 0 && tloc < V) ? Cvt<T>::ld(lr + tloc) : 0.f);
  } else {
    lse = lse_in[row];
    if (threadIdx.x == 0 && loss_rows != nullptr) {
      // In the sharded case the caller supplies the all-reduced target logit through row_tgt.
      loss_rows[row] = ignored ? 0.f : (lse - row_tgt[row]);
    }
  }

  if (!write_grad) return;
  const float scale = ignored ? 0.f : *grad_scale;
  const float nl = -lse * kL2e;
  // The target's element: its vector's owner reads the logit now and stores (p - 1) * scale over what the row
  // pass writes there (a thread's stores to one address land in program order).
  const bool tgt_here = !ignored && tloc >= 0 && tloc < V;
  const bool owner = tgt_here && (int)threadIdx.x == ((tloc >> 3) & (kXentThreads - 1));
  float xt = 0.f;
  if (owner) xt = Cvt<T>::ld(lr + tloc);
  __syncthreads();  // every thread has read its logits before anybody overwrites (target read above)
  if (CACHED && MODE == XENT_FUSED) {
#pragma unroll
    for (int k = 0; k < kXentMaxVec; ++k) {
      const int i = threadIdx.x + k * kXentThreads;
      if (i < nvec) {
        float x[8], o[8];
        cache[k].get(x);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = exp2_scaled(x[j], nl) * scale;
        row_store(lr + i * 8, o);
      }
    }
  } else {
    for (int i = threadIdx.x; i < nvec; i += kXentThreads) {
      float x[8], o[8];
      load8(lr + i * 8, x);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = exp2_scaled(x[j], nl) * scale;
      store8(lr + i * 8, o);
    }
  }
  if (owner) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row pass's store of this element has been taken
    Cvt<T>::st(lr + tloc, (exp2_scaled(xt, nl) - 1.f) * scale);
  }
}

template <typename T>
static void xent_launch(int mode, T* logits, const long long* targets, float* loss_rows, float* row_max,
                        float* row_sum, float* row_tgt, const float* lse_in, const float* grad_scale, int rows,
                        int V, int ld, int vocab_start, long long ignore_index, bool write_grad, hipStream_t st) {
  const dim3 grid(rows), block(kXentThreads);
  const bool cached = V <= kXentMaxVec * 8 * kXentThreads;
#define DPA_X(MODE, C)                                                                                       \
  hipLaunchKernelGGL((xent_kernel<T, MODE, C>), grid, block, 0, st, logits, targets, loss_rows, row_max, row_sum, \
                     row_tgt, lse_in, grad_scale, V, ld, vocab_start, ignore_index, write_grad ? 1 : 0)
  if (mode == XENT_FUSED) {
    if (cached) DPA_X(XENT_FUSED, true); else DPA_X(XENT_FUSED, false);
  } else if (mode == XENT_STATS) {
    if (cached) DPA_X(XENT_STATS, true); else DPA_X(XENT_STATS, false);
  } else {
    DPA_X(XENT_GRAD, false);
  }
#undef DPA_X
}

void xent(int mode, void* logits, bool is_bf16, const long long* targets, float* loss_rows, float* row_max,
          float* row_sum, float* row_tgt, const float* lse_in, const float* grad_scale, int rows, int V, int ld,
          int vocab_start, long long ignore_index, bool write_grad, hipStream_t st) {
  if (rows == 0) return;
  if (is_bf16)
    xent_launch<bf16_t>(mode, (bf16_t*)logits, targets, loss_rows, row_max, row_sum, row_tgt, lse_in, grad_scale,
                        rows, V, ld, vocab_start, ignore_index, write_grad, st);
  else
    xent_launch<float>(mode, (float*)logits, targets, loss_rows, row_max, row_sum, row_tgt, lse_in, grad_scale,
                       rows, V, ld, vocab_start, ignore_index, write_grad, st);
}

}  // namespace dpa
