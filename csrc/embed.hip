// Token/position embedding forward and backward for gfx950.
//
// Forward: out[n, :] = wte[idx[n], :] + pos[(pos0 + n % T), :]   (one fused gather + add)
// Backward (token table): deterministic and atomic-free. The caller sorts the token ids once
// (ids_sorted, perm); runs of equal ids are summed in fp32 in two passes (fixed-size chunks,
// then a fold of the chunk-crossing runs) and added to that row of the (tied lm_head /
// embedding) weight gradient. No two workgroups touch the same row, so bf16 gradients are
// summed exactly once from an fp32 accumulator, and a very frequent token costs no more than
// a rare one.
// Backward (learned position table): dpos[t, :] += sum_b dout[b, t, :].
//
// Token ids are data: an id outside [0, V) (a corrupt shard, a data file of another vocabulary) reads
// as a zero embedding row and receives no gradient; it can never become an out-of-bounds access.
//
// Replaces `nn.Embedding` forward/backward (reference: single-gpu/model.py:541-543,652-666).
#include "common.h"
#include "kernels.h"

namespace dpa {

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ idx, const T* __restrict__ wte,
                                                        const T* __restrict__ pos, T* __restrict__ out, long long N,
                                                        int T_len, int pos0, int C, long long V) {
  const int nvec = C >> 3;
  const long long total = N * nvec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / nvec;
    const int c = (int)(i - n * nvec) * 8;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long id = idx[n];
    if (id >= 0 && id < V) load8(wte + (size_t)id * C + c, a);
    if (pos != nullptr) {
      float p[8];
      load8(pos + (size_t)(pos0 + (int)(n % T_len)) * C + c, p);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += p[j];
    }
    store8(out + (size_t)n * C + c, a);
  }
}

// Token-table backward, pass 1. The sorted positions are cut into chunks of kEmbChunk; one
// workgroup sums the runs of equal ids inside its chunk in fp32. A run that lies completely
// inside the chunk is added to its gradient row right away (no other workgroup touches that
// row). A run that crosses the chunk's left / right edge is parked as the chunk's fp32
// `head` / `tail` partial row for pass 2.
constexpr int kEmbChunk = 32;

template <typename T>
__global__ __launch_bounds__(128) void embed_bwd_chunk_kernel(const T* __restrict__ dout,
                                                              const long long* __restrict__ ids,
                                                              const long long* __restrict__ perm,
                                                              T* __restrict__ dwte, float* __restrict__ head,
                                                              float* __restrict__ tail, long long N, int C,
                                                              long long V) {
  const long long s = (long long)blockIdx.x * kEmbChunk;
  const long long e = (s + kEmbChunk < N) ? s + kEmbChunk : N;
  const bool cont_prev = s > 0 && ids[s - 1] == ids[s];
  const bool cont_next = e < N && ids[e] == ids[e - 1];
  const int nvec = C >> 3;
  for (int cv = threadIdx.x; cv < nvec; cv += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long run_start = s, tok = ids[s];
    for (long long j = s; j <= e; ++j) {
      const long long t = (j < e) ? ids[j] : -1;
      if (t != tok) {  // flush the run [run_start, j)
        if (run_start == s && cont_prev) {
          store8(head + (size_t)blockIdx.x * C + cv * 8, acc);
        } else if (j == e && cont_next) {
          store8(tail + (size_t)blockIdx.x * C + cv * 8, acc);
        } else if (tok >= 0 && tok < V) {
          float g[8];
          load8(dwte + (size_t)tok * C + cv * 8, g);
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] += acc[k];
          store8(dwte + (size_t)tok * C + cv * 8, g);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        tok = t;
        run_start = j;
      }
      if (j < e) {
        float d[8];
        load8(dout + (size_t)perm[j] * C + cv * 8, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += d[k];
      }
    }
  }
}

// Pass 2: the chunk in which a crossing run STARTS folds that run's tail and the head rows of
// the following chunks (found with one binary search for the run's end) and adds the sum to
// the gradient row. Every crossing run has exactly one starting chunk, so rows are written
// once, without atomics, in a fixed order (bitwise reproducible).
template <typename T>
__global__ __launch_bounds__(128) void embed_bwd_fix_kernel(const long long* __restrict__ ids, T* __restrict__ dwte,
                                                            const float* __restrict__ head,
                                                            const float* __restrict__ tail, long long N, int C,
                                                            long long V) {
  const long long s = (long long)blockIdx.x * kEmbChunk;
  const long long e = (s + kEmbChunk < N) ? s + kEmbChunk : N;
  const long long tok = ids[e - 1];
  const bool cont_next = e < N && ids[e] == tok;
  if (!cont_next || tok < 0 || tok >= V) return;
  const bool whole = ids[s] == tok;  // the chunk is a single run
  if (whole && s > 0 && ids[s - 1] == tok) return;  // the run started in an earlier chunk
  long long lo = e, hi = N;  // first position > e-1 whose id differs from tok
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (ids[mid] == tok) lo = mid + 1; else hi = mid;
  }
  const long long last_chunk = (lo - 1) / kEmbChunk;
  const int nvec = C >> 3;
  for (int cv = threadIdx.x; cv < nvec; cv += blockDim.x) {
    float acc[8];
    load8(tail + (size_t)blockIdx.x * C + cv * 8, acc);
    // (eight partial rows per round trip: the run of a frequent token spans hundreds of chunks, and one load per
    // round trip made this fold a 300 us chain of memory latencies in a single workgroup)
    long long j = blockIdx.x + 1;
    for (; j + 8 <= last_chunk + 1; j += 8) {
      float h[8][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) load8(head + (size_t)(j + u) * C + cv * 8, h[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += h[u][k];
    }
    for (; j <= last_chunk; ++j) {
      float h[8];
      load8(head + (size_t)j * C + cv * 8, h);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += h[k];
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
    int b = 0;
    for (; b + 8 <= B; b += 8) {   // eight batch rows per round trip (T * C / 8 threads do not fill the chip)
      float d[8][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) load8(dout + ((size_t)(b + u) * T_len + t) * C + c, d[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += d[u][k];
    }
    for (; b < B; ++b) {
      float d[8];
      load8(dout + ((size_t)b * T_len + t) * C + c, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += d[k];
    }
    store8(dpos + (size_t)(pos0 + t) * C + c, acc);
  }
}

// Decode-time cache append: row `*pos` of k_cache / v_cache [B, S, W] (W = heads * head size) receives the new
// key / value row of every batch element; the position is read from device memory so the launch can be
// replayed from a hipGraph. Positions outside [0, S) write nothing.
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(T* __restrict__ kc, T* __restrict__ vc, const T* __restrict__ kn,
                                                        const T* __restrict__ vn, const long long* __restrict__ pos,
                                                        int B, int S, int W, long long kn_bs, long long vn_bs) {
  const long long p = *pos;
  if (p < 0 || p >= S) return;
  const int wvec = W >> 3;
  const int total = B * wvec;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / wvec, c = (i - b * wvec) * 8;
    const long long dst = ((long long)b * S + p) * W + c;
    *reinterpret_cast<u32x4*>(kc + dst) = *reinterpret_cast<const u32x4*>(kn + b * kn_bs + c);
    *reinterpret_cast<u32x4*>(vc + dst) = *reinterpret_cast<const u32x4*>(vn + b * vn_bs + c);
  }
}

void kv_append(void* kc, void* vc, const void* kn, const void* vn, const long long* pos, int B, int S, int W,
               long long kn_bs, long long vn_bs, hipStream_t st) {
  if (B == 0) return;
  int blocks = (B * (W >> 3) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL((kv_append_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (bf16_t*)kc, (bf16_t*)vc,
                     (const bf16_t*)kn, (const bf16_t*)vn, pos, B, S, W, kn_bs, vn_bs);
}

static int ew_grid(long long nthreads) {
  long long b = (nthreads + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

void embed_fwd(const long long* idx, const void* wte, const void* pos, void* out, long long N, int T_len, int pos0,
               int C, long long V, bool is_bf16, hipStream_t st) {
  if (N == 0) return;
  const dim3 grid(ew_grid(N * (C >> 3))), block(256);
  if (is_bf16)
    hipLaunchKernelGGL((embed_fwd_kernel<bf16_t>), grid, block, 0, st, idx, (const bf16_t*)wte, (const bf16_t*)pos,
                       (bf16_t*)out, N, T_len, pos0, C, V);
  else
    hipLaunchKernelGGL((embed_fwd_kernel<float>), grid, block, 0, st, idx, (const float*)wte, (const float*)pos,
                       (float*)out, N, T_len, pos0, C, V);
}

int embed_bwd_chunks(long long N) { return (int)((N + kEmbChunk - 1) / kEmbChunk); }

void embed_bwd(const void* dout, const long long* ids_sorted, const long long* perm, void* dwte, float* scratch,
               long long N, int C, long long V, bool is_bf16, hipStream_t st) {
  if (N == 0) return;
  const int nch = embed_bwd_chunks(N);
  float* head = scratch;
  float* tail = scratch + (size_t)nch * C;
  const dim3 grid(nch), block(128);
  if (is_bf16) {
    hipLaunchKernelGGL((embed_bwd_chunk_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dout, ids_sorted, perm,
                       (bf16_t*)dwte, head, tail, N, C, V);
    hipLaunchKernelGGL((embed_bwd_fix_kernel<bf16_t>), grid, block, 0, st, ids_sorted, (bf16_t*)dwte, head, tail, N, C, V);
  } else {
    hipLaunchKernelGGL((embed_bwd_chunk_kernel<float>), grid, block, 0, st, (const float*)dout, ids_sorted, perm,
                       (float*)dwte, head, tail, N, C, V);
    hipLaunchKernelGGL((embed_bwd_fix_kernel<float>), grid, block, 0, st, ids_sorted, (float*)dwte, head, tail, N, C, V);
  }
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
