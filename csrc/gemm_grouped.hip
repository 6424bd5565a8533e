// Grouped bf16 GEMM for gfx950 (MoE experts; also usable as a plain GEMM with one group).
//
// Rows of A are grouped into 128-row blocks; block rb multiplies with the weight matrix of
// group g = block_group[rb] (g < 0 or g >= ngroups: the block is padding past the last group
// and is skipped). Token segments are padded to multiples of 128 rows by the caller, so a block
// never straddles two experts and no host-side knowledge of the segment sizes is needed: the
// grid is sized for the static upper bound of rows and every shape decision is read from
// device memory.
//
//   NT form (forward):   C[rows, N] = A[rows, K] * W_g[N, K]^T      (both operands contiguous in K)
//   NN form (dgrad):     C[rows, N] = A[rows, K] * W_g[K, N]        (W rows run along the contraction)
//   TN form (wgrad):     dW_g[N, K] = A[seg_g, N]^T * B[seg_g, K]   (both operands strided; split over tokens)
//
// Tile 128 x 128 x 64, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16
// accumulators. Operand tiles are filled by LDS-DMA with the XOR swizzle applied to the source
// address; K-contiguous operands are read with ds_read_b128, contraction-strided ones with
// ds_read_b64_tr_b16 in natural k order. One 32 KiB LDS buffer per workgroup, four
// workgroups per CU hide each other's DMA latency.
//
// Replaces the Python loop over experts of the reference's MoE layer
// (reference: single-gpu/model.py:490-502).
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_tiles.h"

namespace dpa {

// Operand fragment by transposed read in NATURAL k order (element j of lane half hh is
// contraction index 16*s + 8*hh + j), matching row_frag on the other operand.
__device__ __forceinline__ bfrag tr_frag_nat128(const char* tile, int s, int dt, int lane) {
  const int hh = lane >> 5, g = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  const int chunk = 4 * dt + 2 * g + (pp >> 1);
  s16x4 part[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int row = 16 * s + 8 * hh + 4 * jj + qq;
    const char* a = tile + row * 256 + 16 * swz<128>(row, chunk) + 8 * (pp & 1);
    part[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  }
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = part[0][j];
    v[4 + j] = part[1][j];
  }
  return __builtin_bit_cast(bfrag, v);
}

struct GroupedArgs {
  const bf16_t* a;            // [rows, K]
  long long lda;
  const long long* w_ptrs;    // [ngroups] device addresses of the weight matrices
  long long ldw;              // row stride of every weight matrix
  bf16_t* c;                  // [rows, N]
  long long ldc;
  const int* block_group;     // [rows / 128] group of each row block (outside [0, ngroups): skip)
  int ngroups, rows, N, K;
  int xcd;                    // 1: XCD-aware workgroup order
};

template <bool W_KN>   // false: W_g is [N, K] (NT); true: W_g is [K, N] (NN)
__global__ __launch_bounds__(256, 4) void gemm_grouped_kernel(GroupedArgs p) {
  constexpr int TILE = 128 * 64 * 2;  // 16 KiB per operand tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int ntile_n = (p.N + 127) / 128;
  // XCD-aware order: each XCD (own L2) owns a contiguous run of row blocks with all their column
  // tiles, so the A rows (shared by the column tiles of a row block) and the weight tiles (shared by
  // neighbouring row blocks of one group) are reused through that L2.
  const int bid = p.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int rb = bid / ntile_n, nb = bid % ntile_n;
  const int g = p.block_group ? p.block_group[rb] : 0;
  if (g < 0 || g >= p.ngroups) return;   // padding block (uniform for the whole workgroup)
  const bf16_t* w = reinterpret_cast<const bf16_t*>(p.w_ptrs[g]);
  const int m0 = rb * 128, n0 = nb * 128;
  const int wm = wave >> 1, wn = wave & 1;
  const bf16_t* ap = p.a + (long long)m0 * p.lda;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  char* at = smem;
  char* bt = smem + TILE;
  const int nk = p.K / 64;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    dma_rows128x64(ap + kt * 64, p.lda, p.rows - m0, at, wave, lane);
    if (W_KN) dma_rows64x128(w + (long long)kt * 64 * p.ldw + n0, p.ldw, p.N - n0, bt, wave, lane);
    else dma_rows128x64(w + (long long)n0 * p.ldw + kt * 64, p.ldw, p.N - n0, bt, wave, lane);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bfrag a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = row_frag<64>(at, 64 * wm + 32 * i + r, s, hh);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (W_KN) b[j] = tr_frag_nat128(bt, s, 2 * wn + j, lane);
        else b[j] = row_frag<64>(bt, 64 * wn + 32 * j + r, s, hh);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  }
  // Epilogue through LDS: the accumulators hold one column per lane (2-byte stores, 64 contiguous bytes
  // per half-wave); staged as a [128][128] bf16 image they leave as whole rows, 16 bytes per lane.
  __syncthreads();   // every wave is done reading the operand tiles
  bf16_t* ct = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 64 * wn + 32 * j + r;
#pragma unroll
      for (int q = 0; q < 16; ++q) ct[(64 * wm + 32 * i + acc_row(q, hh)) * 128 + col] = f32_to_bf16(acc[i][j][q]);
    }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int id = tid + 256 * pass, row = id >> 4, ch = id & 15;
    if (m0 + row < p.rows && n0 + ch * 8 < p.N)   // N % 8 == 0 (host-checked): a chunk is inside or outside
      *reinterpret_cast<u32x4*>(p.c + (long long)(m0 + row) * p.ldc + n0 + ch * 8) =
          *reinterpret_cast<const u32x4*>(ct + row * 128 + ch * 8);
  }
}

static int grouped_xcd() {
  static const int v = [] { const char* e = getenv("DPA_GROUPED_XCD"); return e ? atoi(e) : 1; }();
  return v;
}

void gemm_grouped(const void* a, long long lda, const long long* w_ptrs, long long ldw, bool w_is_kn, void* c,
                  long long ldc, const int* block_group, int ngroups, int rows, int N, int K, hipStream_t st) {
  if (rows == 0 || N == 0) return;
  GroupedArgs p;
  p.a = (const bf16_t*)a;
  p.lda = lda;
  p.w_ptrs = w_ptrs;
  p.ldw = ldw;
  p.c = (bf16_t*)c;
  p.ldc = ldc;
  p.block_group = block_group;
  p.ngroups = ngroups;
  p.rows = rows;
  p.N = N;
  p.K = K;
  p.xcd = grouped_xcd();
  const dim3 grid(((rows + 127) / 128) * ((N + 127) / 128)), block(256);
  const size_t lds = 2 * 128 * 64 * 2;
  if (w_is_kn) hipLaunchKernelGGL((gemm_grouped_kernel<true>), grid, block, lds, st, p);
  else hipLaunchKernelGGL((gemm_grouped_kernel<false>), grid, block, lds, st, p);
}

// ------------------------------------------------------------------------------------------
// Grouped weight gradient: dW_g[N, K] = A[seg_g, N]^T * B[seg_g, K], segment bounds on device.
// One workgroup = (group, 128 x 128 output tile, split of the segment); fp32 partial tiles are
// combined with float atomics in the fp32 buffer [ngroups, N, K] (plain stores when a tile has one writer).
// ------------------------------------------------------------------------------------------
struct GroupedWgradArgs {
  const bf16_t* a;   // dY  [rows, N]
  long long lda;
  const bf16_t* b;   // X   [rows, K]
  long long ldb;
  float* out;        // [ngroups, N, K]
  const long long* seg_start;  // [ngroups] first row (multiple of 64)
  const long long* seg_rows;   // [ngroups] rows (multiple of 64, may be 0)
  int ngroups, N, K, splits;
};

__global__ __launch_bounds__(256, 2) void gemm_grouped_wgrad_kernel(GroupedWgradArgs p) {
  constexpr int SUB = 64 * 128 * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int tiles_k = p.K / 128, tiles_n = (p.N + 127) / 128;
  // (Pinning all tiles of a group to one XCD was measured 20 % slower end to end: keep the natural order.)
  const int g = blockIdx.y;
  const int split = blockIdx.x % p.splits, tile = blockIdx.x / p.splits;
  const int tn = tile / tiles_k, tk = tile % tiles_k;
  if (tn >= tiles_n) return;
  const int n0 = tn * 128, k0 = tk * 128;
  const long long rows = p.seg_rows[g], start = p.seg_start[g];
  const long long ntile_all = rows / 64;
  const long long per = (ntile_all + p.splits - 1) / p.splits;
  const long long t0 = split * per;
  long long t1 = t0 + per;
  if (t1 > ntile_all) t1 = ntile_all;
  const bool direct = p.splits == 1;   // sole writer of its tile: plain stores, zeros for an empty segment
  if (t0 >= t1 && !direct) return;
  const int wn = wave >> 1, wk = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  // two LDS stages of (dY sub-tile, X sub-tile); the transfer of stage t+1 (LDS-DMA issued from assembly, so
  // the compiler does not drain it before the next LDS read) overlaps the MFMAs of stage t
  const unsigned lds0 = (unsigned)(size_t)(dma_lptr_t)smem;
  auto issue = [&](long long t, int stage) {
    const long long m = start + t * 64;
    const unsigned base = lds0 + stage * 2 * SUB;
    dma_rows64x128_asm(p.a + m * p.lda + n0, p.lda, p.N - n0, base, wave, lane);
    dma_rows64x128_asm(p.b + m * p.ldb + k0, p.ldb, p.K - k0, base + SUB, wave, lane);
  };
  if (t0 < t1) issue(t0, 0);
  for (long long t = t0; t < t1; ++t) {
    const int stage = (int)((t - t0) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < t1) issue(t + 1, stage ^ 1);
    const char* ya = smem + stage * 2 * SUB;
    const char* xa = ya + SUB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bfrag a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = tr_frag<128>(ya, 0, s, 2 * wn + i, lane);
        b[i] = tr_frag<128>(xa, 0, s, 2 * wk + i, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  }
  float* out = p.out + (long long)g * p.N * p.K;
  const int col = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = k0 + 64 * wk + 32 * j + col;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = n0 + 64 * wn + 32 * i + acc_row(q, hh);
        if (n < p.N && kk < p.K) {
          if (direct) out[(long long)n * p.K + kk] = acc[i][j][q];
          else atomicAdd(out + (long long)n * p.K + kk, acc[i][j][q]);
        }
      }
    }
}

void gemm_grouped_wgrad(const void* a, long long lda, const void* b, long long ldb, float* out,
                        const long long* seg_start, const long long* seg_rows, int ngroups, int N, int K,
                        int max_rows, hipStream_t st) {
  if (ngroups == 0) return;
  GroupedWgradArgs p;
  p.a = (const bf16_t*)a;
  p.lda = lda;
  p.b = (const bf16_t*)b;
  p.ldb = ldb;
  p.out = out;
  p.seg_start = seg_start;
  p.seg_rows = seg_rows;
  p.ngroups = ngroups;
  p.N = N;
  p.K = K;
  const int tiles = ((N + 127) / 128) * (K / 128);
  int splits = 512 / (tiles * ngroups);
  const int max_s = (max_rows / ngroups + 255) / 256;   // ~4 LDS tiles per workgroup for an even load
  if (splits > max_s) splits = max_s;
  if (splits < 1) splits = 1;
  p.splits = splits;
  if (splits > 1) (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)ngroups * N * K, st);
  const dim3 grid(tiles * splits, ngroups), block(256);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_grouped_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_grouped_wgrad_kernel, grid, block, (size_t)(4 * 64 * 128 * 2), st, p);
}

// ------------------------------------------------------------------------------------------
// Row permutation kernels of the MoE dispatch.
//   scatter_rows : dst[row_of_pair[j], :] = src[j / k, :]                       (dispatch)
//   gather_combine: out[t, :] = sum_s gate[t, s] * y[row_of_pair[t*k + s], :]   (combine)
// A negative row marks a pair that has no row in the buffer (dropped by the capacity bound of the expert-parallel
// exchange): it is not written, contributes nothing to the combine and gets a zero gate gradient.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moe_scatter_rows_kernel(const bf16_t* __restrict__ src,
                                                               const long long* __restrict__ row_of_pair,
                                                               bf16_t* __restrict__ dst, long long npairs, int k, int C) {
  const int nvec = C >> 3;
  const long long total = npairs * nvec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long j = i / nvec;
    const int c = (int)(i - j * nvec) * 8;
    const long long row = row_of_pair[j];
    if (row < 0) continue;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (j / k) * C + c);
    *reinterpret_cast<u32x4*>(dst + row * C + c) = v;
  }
}

__global__ __launch_bounds__(256) void moe_gather_combine_kernel(const bf16_t* __restrict__ y,
                                                                 const long long* __restrict__ row_of_pair,
                                                                 const bf16_t* __restrict__ gates,
                                                                 bf16_t* __restrict__ out, long long ntok, int k, int C) {
  const int nvec = C >> 3;
  const long long total = ntok * nvec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / nvec;
    const int c = (int)(i - t * nvec) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < k; ++s) {
      const long long row = row_of_pair[t * k + s];
      if (row < 0) continue;
      const float gt = gates ? bf16_to_f32(gates[t * k + s]) : 1.f;
      float v[8];
      load8(y + row * C + c, v);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += gt * v[q];
    }
    store8(out + t * C + c, acc);
  }
}

// dy[row_of_pair[t*k+s], :] = gate[t,s] * dout[t, :];  dgate[t,s] = <y[row], dout[t]>  (one wave per pair)
__global__ __launch_bounds__(256) void moe_combine_bwd_kernel(const bf16_t* __restrict__ dout,
                                                              const bf16_t* __restrict__ y,
                                                              const long long* __restrict__ row_of_pair,
                                                              const bf16_t* __restrict__ gates,
                                                              bf16_t* __restrict__ dy, float* __restrict__ dgate,
                                                              long long npairs, int k, int C) {
  const int lane = threadIdx.x & 63;
  const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= npairs) return;
  const long long t = pair / k, row = row_of_pair[pair];
  if (row < 0) {   // wave-uniform
    if (lane == 0) dgate[pair] = 0.f;
    return;
  }
  const float gt = bf16_to_f32(gates[pair]);
  float dot = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float d[8], v[8], o[8];
    load8(dout + t * C + c, d);
    load8(y + row * C + c, v);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      dot += d[q] * v[q];
      o[q] = gt * d[q];
    }
    store8(dy + row * C + c, o);
  }
  dot = wave_sum(dot);
  if (lane == 0) dgate[pair] = dot;
}

static int moe_grid(long long n) {
  long long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

void moe_scatter_rows(const void* src, const long long* row_of_pair, void* dst, long long npairs, int k, int C,
                      hipStream_t st) {
  if (npairs == 0) return;
  hipLaunchKernelGGL(moe_scatter_rows_kernel, dim3(moe_grid(npairs * (C >> 3))), dim3(256), 0, st, (const bf16_t*)src,
                     row_of_pair, (bf16_t*)dst, npairs, k, C);
}

// Zeroes the rows of a padded expert-sorted buffer that hold no pair: rows [seg_start[e] + counts[e], seg_start[e] +
// seg_rows[e]) of every expert and everything past the last segment. The buffer's other rows are written by the
// scatter / combine-backward kernels, so the caller can allocate it uninitialised (the padding is at most
// experts x (block - 1) rows of a buffer of hundreds of thousands).
__global__ __launch_bounds__(256) void moe_zero_padding_kernel(bf16_t* __restrict__ dst, const long long* __restrict__ seg_start,
                                                               const long long* __restrict__ counts,
                                                               const long long* __restrict__ seg_rows, int E, int C,
                                                               long long rows_total) {
  const int e = blockIdx.x;
  long long r0, r1;
  if (e < E) {
    r0 = seg_start[e] + counts[e];
    r1 = seg_start[e] + seg_rows[e];
  } else {
    r0 = seg_start[E - 1] + seg_rows[E - 1];
    r1 = rows_total;
  }
  if (r1 > rows_total) r1 = rows_total;
  const int nvec = C >> 3;
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (long long r = r0 + blockIdx.y; r < r1; r += gridDim.y)
    for (int c = threadIdx.x; c < nvec; c += 256) *reinterpret_cast<u32x4*>(dst + r * C + c * 8) = z;
}

void moe_zero_padding(void* dst, const long long* seg_start, const long long* counts, const long long* seg_rows, int E,
                      int C, long long rows_total, int block, hipStream_t st) {
  if (E <= 0 || rows_total == 0) return;
  hipLaunchKernelGGL(moe_zero_padding_kernel, dim3(E + 1, block > 0 ? block : 1), dim3(256), 0, st, (bf16_t*)dst, seg_start,
                     counts, seg_rows, E, C, rows_total);
}

void moe_gather_combine(const void* y, const long long* row_of_pair, const void* gates, void* out, long long ntok,
                        int k, int C, hipStream_t st) {
  if (ntok == 0) return;
  hipLaunchKernelGGL(moe_gather_combine_kernel, dim3(moe_grid(ntok * (C >> 3))), dim3(256), 0, st, (const bf16_t*)y,
                     row_of_pair, (const bf16_t*)gates, (bf16_t*)out, ntok, k, C);
}

void moe_combine_bwd(const void* dout, const void* y, const long long* row_of_pair, const void* gates, void* dy,
                     float* dgate, long long npairs, int k, int C, hipStream_t st) {
  if (npairs == 0) return;
  hipLaunchKernelGGL(moe_combine_bwd_kernel, dim3((unsigned)((npairs + 3) / 4)), dim3(256), 0, st, (const bf16_t*)dout,
                     (const bf16_t*)y, row_of_pair, (const bf16_t*)gates, (bf16_t*)dy, dgate, npairs, k, C);
}

}  // namespace dpa
