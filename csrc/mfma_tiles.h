// MFMA operand / LDS tile helpers shared by the attention and GEMM kernels (gfx950, wave64,
// v_mfma_f32_32x32x16_bf16). A [rows][HS] bf16 tile (HS = 64 or 128 columns) lives in LDS with
// one XOR swizzle of its 16-byte chunks that is conflict-free both for row reads
// (ds_read_b128, one row per lane) and for transposed 4-row block reads (ds_read_b64_tr_b16).
#pragma once
#include "common.h"

namespace dpa {

typedef __attribute__((ext_vector_type(8))) __bf16 bfrag;  // one MFMA A/B operand (8 bf16)
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ f32x16 mfma32(bfrag a, bfrag b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 16-byte-chunk swizzle of a row-major [rows][HS] bf16 LDS tile. Conflict-free for
// ds_read_b128 row reads (lane = row) and for ds_read_b64_tr_b16 4-row block reads.
template <int HS>
__device__ __forceinline__ int swz(int row, int chunk) {
  if (HS == 64) return chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
  return chunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
}

// Accumulator register i of lane-half hh holds row (i&3) + 8*(i>>2) + 4*hh of a 32x32 tile.
__device__ __forceinline__ int acc_row(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

__device__ __forceinline__ bfrag zero_frag() {
  u32x4 z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bfrag, z);
}

// Operand-A fragment by row read: lane (r, hh) gets tile[row][16*ks + 8*hh .. +7].
template <int HS>
__device__ __forceinline__ bfrag row_frag(const char* tile, int row, int ks, int hh) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(tile + row * (HS * 2) + 16 * swz<HS>(row, 2 * ks + hh));
  return __builtin_bit_cast(bfrag, v);
}

// Operand-A fragment by transposed read for a product that sums over the tile's ROWS:
// lane (r, hh) gets tile[rbase + 16*s + 8*(j>>2) + 4*hh + (j&3)][32*dt + r] in element j, which
// is the k order of an accumulator tile used as operand B.
template <int HS>
__device__ __forceinline__ bfrag tr_frag(const char* tile, int rbase, int s, int dt, int lane) {
  const int hh = lane >> 5, g = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  const int chunk = 4 * dt + 2 * g + (pp >> 1);
  s16x4 part[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int row = rbase + 16 * s + 8 * jj + 4 * hh + qq;
    const char* a = tile + row * (HS * 2) + 16 * swz<HS>(row, chunk) + 8 * (pp & 1);
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

// Registers 8*s .. 8*s+7 of an accumulator tile -> operand-B fragment of k-step s.
__device__ __forceinline__ bfrag acc_frag(const f32x16& a, int s) {
  u32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(a[8 * s + 2 * j], a[8 * s + 2 * j + 1]);
  return __builtin_bit_cast(bfrag, v);
}

// Global -> register -> LDS staging of a [64][HS] tile by 256 threads.
template <int HS>
struct Stage64 {
  static constexpr int CPR = HS / 8;
  static constexpr int NLD = 64 * CPR / 256;
  u32x4 reg[NLD];
  __device__ __forceinline__ void issue(const bf16_t* base, int row0, int nrows, long long ts, int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + 256 * i, row = id / CPR, ch = id % CPR;
      u32x4 z = {0u, 0u, 0u, 0u};
      if (row0 + row < nrows)
        z = *reinterpret_cast<const u32x4*>(base + (long long)(row0 + row) * ts + ch * 8);
      reg[i] = z;
    }
  }
  __device__ __forceinline__ void commit(char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + 256 * i, row = id / CPR, ch = id % CPR;
      *reinterpret_cast<u32x4*>(tile + row * (HS * 2) + 16 * swz<HS>(row, ch)) = reg[i];
    }
  }
};

}  // namespace dpa
