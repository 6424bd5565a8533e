// Token/position embedding forward and backward for gfx950.
//
// Forward: out[n, :] = wte[idx[n], :] + pos[(pos0 + n % T), :]   (one fused gather + add)
// Backward (token table): deterministic and atomic-free. The caller sorts the token ids once
// (ids_sorted, perm); every run of equal ids is summed in fp32 by the workgroup that sits on
// the run's first element and added to that row of the (tied lm_head / embedding) weight
// gradient. No two workgroups touch the same row, so bf16 gradients are summed exactly once
// from an fp32 accumulator.
// Backward (learned position table): dpos[t, :] += sum_b dout[b, t, :].
//
// Replaces `nn.Embedding` forward/backward (reference: single-gpu/model.py:541-543,652-666).
#include "common.h"
#include "kernels.h"

namespace dpa {

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ idx, const T* __restrict__ wte,
                                                        const T* __restrict__ pos, T* __restrict__ out, long long N,
                                                        int T_len, int pos0, int C) {
  const int nvec = C >> 3;
  const long long total = N * nvec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / nvec;
    const int c = (int)(i - n * nvec) * 8;
    float a[8];
    load8(wte + (size_t)idx[n] * C + c, a);
    if (pos != nullptr) {
      float p[8];
      load8(pos + (size_t)(pos0 + (int)(n % T_len)) * C + c, p);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += p[j];
    }
    store8(out + (size_t)n * C + c, a);
  }
}

template <typename T>
__global__ __launch_bounds__(128) void embed_bwd_kernel(const T* __restrict__ dout, const long long* __restrict__ ids_sorted,
                                                        const long long* __restrict__ perm, T* __restrict__ dwte,
                                                        long long N, int C) {
  const long long i = blockIdx.x;
  const long long tok = ids_sorted[i];
  if (i > 0 && ids_sorted[i - 1] == tok) return;  // not the head of a run
  const int nvec = C >> 3;
  for (int cv = threadIdx.x; cv < nvec; cv += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long j = i; j < N && ids_sorted[j] == tok; ++j) {
      float d[8];
      load8(dout + (size_t)perm[j] * C + cv * 8, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += d[k];
    }
    float g[8];
    load8(dwte + (size_t)tok * C + cv * 8, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] += acc[k];
    store8(dwte + (size_t)tok * C + cv * 8, g);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pos_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dpos, int B,
                                                      int T_len, int pos0, int C) {
  const int nvec = C >> 3;
  const long long total = (long long)T_len * nvec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i / nvec);
    const int c = (int)(i - (long long)t * nvec) * 8;
    float acc[8];
    load8(dpos + (size_t)(pos0 + t) * C + c, acc);
    for (int b = 0; b < B; ++b) {
      float d[8];
      load8(dout + ((size_t)b * T_len + t) * C + c, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += d[k];
    }
    store8(dpos + (size_t)(pos0 + t) * C + c, acc);
  }
}

static int ew_grid(long long nthreads) {
  long long b = (nthreads + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

void embed_fwd(const long long* idx, const void* wte, const void* pos, void* out, long long N, int T_len, int pos0,
               int C, bool is_bf16, hipStream_t st) {
  if (N == 0) return;
  const dim3 grid(ew_grid(N * (C >> 3))), block(256);
  if (is_bf16)
    hipLaunchKernelGGL((embed_fwd_kernel<bf16_t>), grid, block, 0, st, idx, (const bf16_t*)wte, (const bf16_t*)pos,
                       (bf16_t*)out, N, T_len, pos0, C);
  else
    hipLaunchKernelGGL((embed_fwd_kernel<float>), grid, block, 0, st, idx, (const float*)wte, (const float*)pos,
                       (float*)out, N, T_len, pos0, C);
}

void embed_bwd(const void* dout, const long long* ids_sorted, const long long* perm, void* dwte, long long N, int C,
               bool is_bf16, hipStream_t st) {
  if (N == 0) return;
  const dim3 grid((unsigned)N), block(128);
  if (is_bf16)
    hipLaunchKernelGGL((embed_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dout, ids_sorted, perm,
                       (bf16_t*)dwte, N, C);
  else
    hipLaunchKernelGGL((embed_bwd_kernel<float>), grid, block, 0, st, (const float*)dout, ids_sorted, perm,
                       (float*)dwte, N, C);
}

void pos_bwd(const void* dout, void* dpos, int B, int T_len, int pos0, int C, bool is_bf16, hipStream_t st) {
  const dim3 grid(ew_grid((long long)T_len * (C >> 3))), block(256);
  if (is_bf16)
    hipLaunchKernelGGL((pos_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dout, (bf16_t*)dpos, B, T_len,
                       pos0, C);
  else
    hipLaunchKernelGGL((pos_bwd_kernel<float>), grid, block, 0, st, (const float*)dout, (float*)dpos, B, T_len,
                       pos0, C);
}

}  // namespace dpa
