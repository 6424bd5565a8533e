// LayerNorm forward / backward for gfx950.
//
// One wave64 owns one row; the row lives in registers between the statistics pass and the
// normalisation pass, so x is read from HBM exactly once. All global accesses are 16-byte
// vectors. Optional fusions: residual add in the forward (y = LN(x + r), sum written back as
// the new residual stream) and residual-gradient add in the backward (dx += dres).
//
// Replaces the implicit `nn.LayerNorm` kernels of the reference model
// (reference: single-gpu/model.py:514-515,557).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dpa {

// Sum over the LPR lanes (32 or 64) that share a row.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LPR lanes own one row (LPR = 32: two rows per wave, used when the row has an odd multiple of
// 32 sixteen-byte chunks, e.g. C = 768 -> 96 chunks = 3 per lane with no idle lanes).
template <typename T, int NCH, int LPR, bool NT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                     const T* __restrict__ w, const T* __restrict__ b,
                                                     T* __restrict__ y, T* __restrict__ sum_out,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, int C, float eps) {
  constexpr int RPW = 64 / LPR;                       // rows per wave
  const int lane = threadIdx.x & (LPR - 1);
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const T* xr = x + (size_t)row * C;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * LPR;
    if (c < nchunk) {
      load8s<NT>(xr + c * 8, v[i]);
      if (res != nullptr) {
        float r[8];
        load8s<NT>(res + (size_t)row * C + c * 8, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += r[j];
        if (sum_out != nullptr) {
          // Round through the storage dtype so the statistics see what later kernels will read.
          store8s<NT>(sum_out + (size_t)row * C + c * 8, v[i]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  }
  const float mean = row_sum<LPR>(s) / C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * LPR;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[i][j] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = rsqrtf(row_sum<LPR>(ss) / C + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * LPR;
    if (c < nchunk) {
      float wv[8], bv[8], o[8];
      load8(w + c * 8, wv);
      load8(b + c * 8, bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
      store8s<NT>(y + (size_t)row * C + c * 8, o);
    }
  }
}

// Backward. Each wave walks rows with a grid stride and keeps its dgamma/dbeta partial sums in
// registers; the four waves of a block are combined through LDS and written as one partial
// row per block. A second kernel folds the partial rows.
template <typename T, int NCH, int LPR, bool NT>
__global__ __launch_bounds__(256, (NCH * (int)sizeof(T) <= 6) ? 3 : 1) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const T* __restrict__ w, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ part_dw,
                                                     float* __restrict__ part_db, int rows, int C) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & (LPR - 1), wid = threadIdx.x >> 6, sub = (threadIdx.x & 63) / LPR;
  const int nchunk = C >> 3;
  float aw[NCH][8], ab[NCH][8];
  Vec8<T> wv[NCH];   // gamma stays in its storage format: with bf16 that keeps the kernel at 3 waves per SIMD
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * LPR;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      aw[i][j] = 0.f;
      ab[i][j] = 0.f;
    }
    wv[i].load(w + (c < nchunk ? c : 0) * 8);
  }
  for (int row = (blockIdx.x * 4 + wid) * RPW + sub; row < rows; row += gridDim.x * 4 * RPW) {
    const float mu = mean[row], rs = rstd[row];
    float g[NCH][8], xh[NCH][8];
    float s1 = 0.f, s2 = 0.f;
    Vec8<T> rv[NCH];   // residual-stream gradient, requested up front so it overlaps the row reduction
    if (dres != nullptr) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * LPR;
        if (c < nchunk) rv[i].template loads<NT>(dres + (size_t)row * C + c * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * LPR;
      if (c < nchunk) {
        float d[8], xv[8], wf[8];
        load8s<NT>(dy + (size_t)row * C + c * 8, d);
        load8s<NT>(x + (size_t)row * C + c * 8, xv);
        wv[i].pin();
        wv[i].get(wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (xv[j] - mu) * rs;
          g[i][j] = d[j] * wf[j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          aw[i][j] += d[j] * xh[i][j];
          ab[i][j] += d[j];
        }
      }
    }
    s1 = row_sum<LPR>(s1) / C;
    s2 = row_sum<LPR>(s2) / C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * LPR;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
        if (dres != nullptr) {
          float r[8];
          rv[i].get(r);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += r[j];
        }
        store8s<NT>(dx + (size_t)row * C + c * 8, o);
      }
    }
  }
  // Combine the 4*RPW row groups of the block: lds[group][C] for dw, then the same for db.
  constexpr int G = 4 * RPW;
  float* lw = lds;            // G*C
  float* lb = lds + G * C;    // G*C
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * LPR;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lw[(wid * RPW + sub) * C + c * 8 + j] = aw[i][j];
        lb[(wid * RPW + sub) * C + c * 8 + j] = ab[i][j];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float tw = 0.f, tb = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      tw += lw[g * C + c];
      tb += lb[g * C + c];
    }
    part_dw[(size_t)blockIdx.x * C + c] = tw;
    part_db[(size_t)blockIdx.x * C + c] = tb;
  }
}

// Fold `nparts` partial rows into the parameter gradients. One 1024-thread block owns 64
// columns; its 16 wave-rows each walk every 16th partial row with independent (unrolled)
// loads, so the fold is a handful of memory round trips instead of a serial chain. With
// `accumulate` the result is added to what dw/db already hold.
template <typename T>
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ part_dw,
                                                             const float* __restrict__ part_db, T* __restrict__ dw,
                                                             T* __restrict__ db, int nparts, int C, int accumulate) {
  __shared__ float red[2][16][64];
  const int l = threadIdx.x & 63;
  const int col = blockIdx.x * 64 + l;
  const int seg = threadIdx.x >> 6;
  float sw = 0.f, sb = 0.f;
  if (col < C) {
    int p = seg;
    for (; p + 48 < nparts; p += 64) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = part_dw[(size_t)(p + 16 * u) * C + col];
        b[u] = part_db[(size_t)(p + 16 * u) * C + col];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sw += a[u];
        sb += b[u];
      }
    }
    for (; p < nparts; p += 16) {
      sw += part_dw[(size_t)p * C + col];
      sb += part_db[(size_t)p * C + col];
    }
  }
  red[0][seg][l] = sw;
  red[1][seg][l] = sb;
  __syncthreads();
  if (seg == 0 && col < C) {
    float tw = 0.f, tb = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      tw += red[0][k][l];
      tb += red[1][k][l];
    }
    if (accumulate) {
      tw += Cvt<T>::ld(dw + col);
      tb += Cvt<T>::ld(db + col);
    }
    Cvt<T>::st(dw + col, tw);
    Cvt<T>::st(db + col, tb);
  }
}

// Activations of at least this many bytes go through the kernels with the streaming cache policy (benchmark knob:
// DPA_LN_STREAM_MB).
static long long ln_stream_bytes() {
  static const long long v = [] {
    const char* e = getenv("DPA_LN_STREAM_MB");
    return e ? (long long)atoll(e) << 20 : stream_threshold_bytes();
  }();
  return v;
}

// Lanes per row and chunks per lane for a row of C elements (C % 8 == 0).
static void pick_shape(int C, int& lpr, int& nch) {
  const int nchunk = C / 8;
  lpr = (nchunk % 32 == 0 && (nchunk % 64 != 0 || nchunk < 64) && nchunk <= 32 * 8) ? 32 : 64;
  const int need = (nchunk + lpr - 1) / lpr;
  nch = 1;
  while (nch < need) nch <<= 1;
  if (lpr == 32 && nch != need) {       // keep the 32-lane form only when it divides evenly (1,2,3,4,6,8 -> pad to pow2 otherwise)
    if (need == 3) nch = 3;
    else if (need == 6) nch = 6;
  }
}

template <typename T>
static void ln_fwd_launch(const T* x, const T* res, const T* w, const T* b, T* y, T* sum_out, float* mean,
                          float* rstd, int rows, int C, float eps, hipStream_t st) {
  int lpr, nch;
  pick_shape(C, lpr, nch);
  const int rpb = 4 * (64 / lpr);
  const dim3 grid((rows + rpb - 1) / rpb), block(256);
  const bool nt = (long long)rows * C * (long long)sizeof(T) >= ln_stream_bytes();
#define DPA_CASE(N, L)                                                                                              \
  do {                                                                                                              \
    if (nt) hipLaunchKernelGGL((ln_fwd_kernel<T, N, L, true>), grid, block, 0, st, x, res, w, b, y, sum_out, mean, rstd, rows, C, eps); \
    else hipLaunchKernelGGL((ln_fwd_kernel<T, N, L, false>), grid, block, 0, st, x, res, w, b, y, sum_out, mean, rstd, rows, C, eps);   \
  } while (0)
  if (lpr == 32) {
    switch (nch) {
      case 1: DPA_CASE(1, 32); break;
      case 2: DPA_CASE(2, 32); break;
      case 3: DPA_CASE(3, 32); break;
      case 4: DPA_CASE(4, 32); break;
      case 6: DPA_CASE(6, 32); break;
      default: DPA_CASE(8, 32); break;
    }
  } else {
    switch (nch) {
      case 1: DPA_CASE(1, 64); break;
      case 2: DPA_CASE(2, 64); break;
      case 4: DPA_CASE(4, 64); break;
      case 8: DPA_CASE(8, 64); break;
      default: DPA_CASE(16, 64); break;
    }
  }
#undef DPA_CASE
}

template <typename T>
static void ln_bwd_launch(const T* dy, const T* x, const T* w, const float* mean, const float* rstd,
                          const T* dres, T* dx, T* dw, T* db, float* scratch, int nparts, int rows, int C,
                          int accumulate, hipStream_t st) {
  int lpr, nch;
  pick_shape(C, lpr, nch);
  const dim3 grid(nparts), block(256);
  const size_t lds = (size_t)2 * 4 * (64 / lpr) * C * sizeof(float);
  float* pw = scratch;
  float* pb = scratch + (size_t)nparts * C;
  const bool nt = (long long)rows * C * (long long)sizeof(T) >= ln_stream_bytes();
#define DPA_CASE(N, L)                                                                                                 \
  do {                                                                                                                 \
    if (nt) hipLaunchKernelGGL((ln_bwd_kernel<T, N, L, true>), grid, block, lds, st, dy, x, w, mean, rstd, dres, dx, pw, pb, rows, C); \
    else hipLaunchKernelGGL((ln_bwd_kernel<T, N, L, false>), grid, block, lds, st, dy, x, w, mean, rstd, dres, dx, pw, pb, rows, C);   \
  } while (0)
  if (lpr == 32) {
    switch (nch) {
      case 1: DPA_CASE(1, 32); break;
      case 2: DPA_CASE(2, 32); break;
      case 3: DPA_CASE(3, 32); break;
      case 4: DPA_CASE(4, 32); break;
      case 6: DPA_CASE(6, 32); break;
      default: DPA_CASE(8, 32); break;
    }
  } else {
    switch (nch) {
      case 1: DPA_CASE(1, 64); break;
      case 2: DPA_CASE(2, 64); break;
      case 4: DPA_CASE(4, 64); break;
      case 8: DPA_CASE(8, 64); break;
      default: DPA_CASE(16, 64); break;
    }
  }
#undef DPA_CASE
  hipLaunchKernelGGL((ln_bwd_reduce_kernel<T>), dim3((C + 63) / 64), dim3(1024), 0, st, pw, pb, dw, db, nparts,
                     C, accumulate);
}

int layernorm_max_cols() { return 16 * 64 * 8; }  // 8192; LDS in the backward: 8*C*4 B <= 160 KiB -> C <= 5120

int layernorm_bwd_parts(int rows, int C) {
  // One round of resident workgroups: LDS (2 * rows-per-block * C floats) and registers allow three
  // per CU for narrow rows, two from C = 2048 on; a partial second round would run at half occupancy.
  int lpr, nch;
  pick_shape(C, lpr, nch);
  const int rpb = 4 * (64 / lpr);
  const long long lds = 2LL * rpb * C * 4;
  int per_cu = (int)(160 * 1024 / (lds > 0 ? lds : 1));
  if (per_cu > 3) per_cu = 3;
  if (per_cu < 1) per_cu = 1;
  if (nch * 2 > 6 && per_cu > 2) per_cu = 2;   // register budget of the wide-row instantiations
  const int cap = 256 * per_cu;
  const int n = (rows + rpb - 1) / rpb;
  return n < 1 ? 1 : (n > cap ? cap : n);
}

void layernorm_fwd(const void* x, const void* res, const void* w, const void* b, void* y, void* sum_out,
                   float* mean, float* rstd, int rows, int C, float eps, bool is_bf16, hipStream_t st) {
  if (is_bf16)
    ln_fwd_launch<bf16_t>((const bf16_t*)x, (const bf16_t*)res, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y,
                          (bf16_t*)sum_out, mean, rstd, rows, C, eps, st);
  else
    ln_fwd_launch<float>((const float*)x, (const float*)res, (const float*)w, (const float*)b, (float*)y,
                         (float*)sum_out, mean, rstd, rows, C, eps, st);
}

void layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                   const void* dres, void* dx, void* dw, void* db, float* scratch, int nparts, int rows, int C,
                   bool accumulate, bool is_bf16, hipStream_t st) {
  if (is_bf16)
    ln_bwd_launch<bf16_t>((const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, mean, rstd, (const bf16_t*)dres,
                          (bf16_t*)dx, (bf16_t*)dw, (bf16_t*)db, scratch, nparts, rows, C, accumulate ? 1 : 0, st);
  else
    ln_bwd_launch<float>((const float*)dy, (const float*)x, (const float*)w, mean, rstd, (const float*)dres,
                         (float*)dx, (float*)dw, (float*)db, scratch, nparts, rows, C, accumulate ? 1 : 0, st);
}

}  // namespace dpa
