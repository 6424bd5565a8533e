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

// 8 bf16 (one MFMA operand fragment) at LDS byte address addr + imm (imm: a constant that goes into the offset field).
__device__ __forceinline__ bfrag lds_frag_at(unsigned addr, int imm) {
  typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
  const u32x4 v = *(lds_u32x4*)(size_t)(addr + (unsigned)imm);
  return __builtin_bit_cast(bfrag, v);
}

// The same fragment from two precomputed LDS byte addresses (the 8-row halves jj = 0, 1 of k-step 0 of a tile) plus a
// constant: a ring of tiles whose slot changes at run time keeps one such address pair per (operand, dt) in registers,
// adds the slot's offset once per slot and reaches the tile's second k-step through the instruction's offset field,
// instead of re-deriving row, swizzle and slot for every read.
__device__ __forceinline__ bfrag tr_frag_at(const unsigned (&addr)[2], int imm) {
  s16x4 part[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
    part[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(addr[jj] + (unsigned)imm));
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = part[0][j];
    v[4 + j] = part[1][j];
  }
  return __builtin_bit_cast(bfrag, v);
}
// LDS byte offset (inside a [rows][HS] tile) of half jj of the fragment that tr_frag(tile, 0, 0, dt, lane) reads.
template <int HS>
__device__ __forceinline__ unsigned tr_frag_offset(int dt, int jj, int lane) {
  const int hh = lane >> 5, g = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  const int chunk = 4 * dt + 2 * g + (pp >> 1);
  const int row = 8 * jj + 4 * hh + qq;
  return (unsigned)(row * (HS * 2) + 16 * swz<HS>(row, chunk) + 8 * (pp & 1));
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


// ---- global -> LDS tile transfers (LDS-DMA: global_load_lds, 16 bytes per lane, no registers) --------
// A piece is the 1 KiB that one wave instruction moves (64 lanes x 16 bytes, contiguous in LDS); the
// XOR swizzle is applied on the global side: LDS slot `sl` of a row receives logical chunk swz(row, sl).
typedef __attribute__((address_space(1))) const void* dma_gptr_t;
typedef __attribute__((address_space(3))) void* dma_lptr_t;

// [128 rows][64 cols] tile of a K-contiguous operand (128-byte rows, 16 pieces of 8 rows), 4 waves.
// Rows at or past `row_limit` re-read row 0 (their results are never stored).
__device__ __forceinline__ void dma_rows128x64(const bf16_t* g, long long ld, int row_limit, char* lds, int wave,
                                               int lane) {
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    const int piece = wave * 4 + pc;
    int row = piece * 8 + (lane >> 3);
    const int chunk = swz<64>(row, lane & 7);
    if (row >= row_limit) row = 0;
    __builtin_amdgcn_global_load_lds((dma_gptr_t)(g + (long long)row * ld + chunk * 8),
                                     (dma_lptr_t)(lds + piece * 1024), 16, 0, 0);
  }
}

// [64 rows][128 cols] tile of a contraction-strided operand (256-byte rows, 16 pieces of 4 rows), 4
// waves. Columns at or past `col_limit` re-read chunk 0 (never stored).
__device__ __forceinline__ void dma_rows64x128(const bf16_t* g, long long ld, int col_limit, char* lds, int wave,
                                               int lane) {
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    const int piece = wave * 4 + pc;
    const int row = piece * 4 + (lane >> 4);
    int chunk = swz<128>(row, lane & 15);
    if (chunk * 8 >= col_limit) chunk = 0;
    __builtin_amdgcn_global_load_lds((dma_gptr_t)(g + (long long)row * ld + chunk * 8),
                                     (dma_lptr_t)(lds + piece * 1024), 16, 0, 0);
  }
}

// The [64][128] transfer issued from inline assembly. hipcc treats an LDS-DMA builtin as an LDS write
// that any later ds_read may alias and drains it (vmcnt(0)) before the next LDS read, which
// serialises a multi-stage pipeline. Issued from asm the transfer is invisible to the compiler's
// wait insertion and is retired by hand with s_waitcnt vmcnt before the barrier that publishes the
// stage. M0 (LDS base of the piece) is written and restored inside the same statement.
__device__ __forceinline__ void dma_rows64x128_asm(const bf16_t* g, long long ld, int col_limit, unsigned lds_addr,
                                                   int wave, int lane) {
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    const int piece = wave * 4 + pc;
    const int row = piece * 4 + (lane >> 4);
    int chunk = swz<128>(row, lane & 15);
    if (chunk * 8 >= col_limit) chunk = 0;
    const bf16_t* src = g + (long long)row * ld + chunk * 8;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr + piece * 1024);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst)
        : "memory");
  }
}

// The same transfer for a [32 rows][128 cols] tile (8 pieces of 4 rows, 2 per wave). No "memory" clobber: the
// compiler may move its own LDS reads across the statement (they touch other ring slots; the ordering that
// matters is made by the caller's s_waitcnt + barrier statements, which do carry the clobber), so that a
// transfer can sit in the middle of an MFMA / LDS-read stream.
__device__ __forceinline__ void dma_rows32x128_asm(const bf16_t* g, long long ld, int col_limit, unsigned lds_addr,
                                                   int wave, int lane) {
#pragma unroll
  for (int pc = 0; pc < 2; ++pc) {
    const int piece = wave * 2 + pc;
    const int row = piece * 4 + (lane >> 4);
    int chunk = swz<128>(row, lane & 15);
    if (chunk * 8 >= col_limit) chunk = 0;
    const bf16_t* src = g + (long long)row * ld + chunk * 8;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr + piece * 1024);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst));
  }
}

// One 1 KiB piece of such a transfer with the row base in scalar registers: global address = sbase + voff (the lane's
// byte offset inside the 32-row block, computed once per kernel), LDS address = dst (a scalar). M0 is written and
// not restored: the kernels that use this touch M0 nowhere else (no other instruction of theirs reads it on gfx950),
// and the save / restore pair plus the per-lane 64-bit address arithmetic were a third of the ring loop's instructions.
__device__ __forceinline__ void dma_piece_sbase(const void* sbase, unsigned voff, unsigned dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %2"
      :
      : "v"(voff), "s"(dst), "s"(sbase)
      : "m0");
}

}  // namespace dpa
