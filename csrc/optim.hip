// Fused AdamW over flat buffers + global-norm reduction for gfx950.
//
// The optimizer state of the whole model (or of this rank's shard of it) lives in three flat
// fp32 buffers (master weights, exp_avg, exp_avg_sq). One launch updates all of it, reads the
// (bf16 or fp32) flat gradient once, applies gradient clipping from a device-resident squared
// norm (no host synchronisation) and writes the low-precision compute copy of the weights in
// the same pass. A non-finite squared norm skips the update (the overflow-skip semantics of a
// gradient scaler, without a separate unscale/inf-check pass).
//
// Replaces `torch.optim.AdamW(fused=True)`, `clip_grad_norm_` and `GradScaler.unscale_/step`
// of the reference trainers (reference: single-gpu/model.py:619-637, single-gpu/train.py:345-352).
#include "common.h"
#include "kernels.h"

namespace dpa {

#ifndef DPA_OPT_NT
#define DPA_OPT_NT 0   // non-temporal accesses of the state and the weights: measured, no change (173 us either way)
#endif
constexpr bool kOptNt = DPA_OPT_NT != 0;

template <typename GT, typename PT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v, const GT* __restrict__ grad,
                                                    PT* __restrict__ param_out, long long n, AdamSegs segs,
                                                    float lr, float beta1, float beta2, float eps, float bc1,
                                                    float bc2_sqrt, const float* __restrict__ norm2,
                                                    float max_norm, float grad_mul,
                                                    const float* __restrict__ dyn) {
  // Step-dependent scalars read from device memory (lr, 1 - beta1^t, sqrt(1 - beta2^t)) when the
  // step is replayed from a captured graph, where kernel arguments are frozen at capture time.
  if (dyn != nullptr) {
    lr = dyn[0];
    bc1 = dyn[1];
    bc2_sqrt = dyn[2];
  }
  float gscale = grad_mul;
  if (norm2 != nullptr) {
    const float n2 = *norm2;
    if (!isfinite(n2)) return;  // overflow: leave weights and state untouched
    if (max_norm > 0.f) {
      const float norm = sqrtf(n2) * fabsf(grad_mul);
      const float clip = max_norm / (norm + 1e-6f);
      if (clip < 1.f) gscale *= clip;
    }
  }
  const long long nvec = n >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i << 3;
    float wd = 0.f;
#pragma unroll 1
    for (int s = 0; s < segs.nseg; ++s) {
      if (e < segs.end[s]) {
        wd = segs.wd[s];
        break;
      }
    }
    float p[8], mm[8], vv[8], g[8];
    load8s<kOptNt>(master + e, p);
    load8s<kOptNt>(m + e, mm);
    load8s<kOptNt>(v + e, vv);
    load8s<kOptNt>(grad + e, g);
    const float decay = 1.f - lr * wd;
    const float step = lr / bc1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gj = g[j] * gscale;
      mm[j] = beta1 * mm[j] + (1.f - beta1) * gj;
      vv[j] = beta2 * vv[j] + (1.f - beta2) * gj * gj;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      p[j] = p[j] * decay - step * (mm[j] / denom);
    }
    store8s<kOptNt>(master + e, p);
    store8s<kOptNt>(m + e, mm);
    store8s<kOptNt>(v + e, vv);
    if (param_out != nullptr) store8s<kOptNt>(param_out + e, p);
  }
}

template <typename GT>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const GT* __restrict__ g, long long n,
                                                            float* __restrict__ partials) {
  __shared__ float red[16];
  float acc = 0.f;
  const long long nvec = n >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float x[8];
    load8(g + (i << 3), x);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += x[j] * x[j];
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void sumsq_final_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ out,
                                   int accumulate) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += partials[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) *out = accumulate ? (*out + tot) : tot;
}

static int flat_grid(long long n) {
  long long b = ((n >> 3) + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  return (int)b;
}

void adamw_step(float* master, float* m, float* v, const void* grad, bool grad_bf16, void* param_out,
                bool param_bf16, long long n, const AdamSegs& segs, float lr, float beta1, float beta2, float eps,
                float bc1, float bc2_sqrt, const float* norm2, float max_norm, float grad_mul, const float* dyn,
                hipStream_t st) {
  const dim3 grid(flat_grid(n)), block(256);
#define DPA_LAUNCH(GT, PT)                                                                                    \
  hipLaunchKernelGGL((adamw_kernel<GT, PT>), grid, block, 0, st, master, m, v, (const GT*)grad, (PT*)param_out, \
                     n, segs, lr, beta1, beta2, eps, bc1, bc2_sqrt, norm2, max_norm, grad_mul, dyn)
  if (grad_bf16) {
    if (param_bf16) DPA_LAUNCH(bf16_t, bf16_t);
    else DPA_LAUNCH(bf16_t, float);
  } else {
    if (param_bf16) DPA_LAUNCH(float, bf16_t);
    else DPA_LAUNCH(float, float);
  }
#undef DPA_LAUNCH
}

int sumsq_parts(long long n) { return flat_grid(n); }

void sumsq(const void* g, bool is_bf16, long long n, float* partials, float* out, bool accumulate,
           hipStream_t st) {
  const int parts = flat_grid(n);
  if (is_bf16)
    hipLaunchKernelGGL((sumsq_partial_kernel<bf16_t>), dim3(parts), dim3(256), 0, st, (const bf16_t*)g, n, partials);
  else
    hipLaunchKernelGGL((sumsq_partial_kernel<float>), dim3(parts), dim3(256), 0, st, (const float*)g, n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, partials, parts, out, accumulate ? 1 : 0);
}

}  // namespace dpa
