// Weight-gradient GEMM for gfx950:  dW[N, K] (+)= dY[M, N]^T * X[M, K]   (+ dbias[N] = sum_m dY)
//
// This is the GEMM shape that the backward pass of every Linear layer needs: the contraction
// runs over the M = batch*seq tokens (tens of thousands) while the output is a small weight
// matrix, and BOTH operands are "contraction-strided" (row-major activations, contraction
// down the rows). Library kernels launch a few dozen workgroups for it and crawl.
//
// Common design: split-M (the token range is divided over workgroups so that a small output
// matrix still fills 256 CUs), dY and X tiles staged row-major into LDS exactly as they sit in
// memory under an XOR swizzle, both MFMA operands fetched with the transposed LDS read
// (ds_read_b64_tr_b16) so no transposed copy of an activation is ever made, fp32 partial results
// per split folded into the bf16 (or fp32) gradient by a small finishing pass.
//
// Three kernels, fastest applicable one first:
//  * wgrad3_kernel: 256x256 output tile, 4 waves of 128x128 (accumulators in AGPRs), one workgroup
//    per CU, two 64 KiB LDS stages filled by LDS-DMA from inline assembly; per-split fp32 slabs;
//    optional bias gradient from the dY fragments. Needs N % 128 == 0 (a half-empty last row of tiles is
//    allowed) and K % 256 == 0.
//  * wgrad2_kernel: 128x256 tile, 4 waves of 128x64, single 48 KiB LDS buffer filled by LDS-DMA,
//    two workgroups per CU hide each other's transfer latency. Needs K % 256 == 0, M % 64 == 0.
//  * wgrad_kernel: 128x128 tile, register-staged double buffer, fp32 workspace with float atomics
//    across splits, bias gradient folded into the staging pass. Needs N, K % 128 == 0.
//
// Replaces the dW GEMMs that `nn.Linear`'s backward issues implicitly in the reference
// (reference: single-gpu/model.py:112-114,373-384,559).
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "mfma_tiles.h"
#include "host_logic.h"

namespace dpa {

// Timing ablations of wgrad3_kernel (benchmark builds only; results are wrong): 1 = no wait for the LDS-DMA,
// 2 = no LDS-DMA after the first tile, 3 = LDS-DMA only (no fragment reads, no MFMAs)
#ifndef DPA_WG3_ABLATE
#define DPA_WG3_ABLATE 0
#endif
// 1 = ring of four 32-token LDS slots with fragments requested one token step ahead; 0 = two 64-token stages
#ifndef DPA_WG3_RING
#define DPA_WG3_RING 1
#endif
#ifndef DPA_WG3_SLOTS
#define DPA_WG3_SLOTS 5     // ring slots of 32 KiB (4 or 5: 128 or 160 KiB of LDS)
#endif

constexpr int kWgTile = 128;   // output tile edge
constexpr int kWgStep = 64;    // tokens per LDS tile

template <bool ATOMIC, bool BIAS>
__global__ __launch_bounds__(256) void wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                    float* __restrict__ ws, float* __restrict__ dbias, int M, int N,
                                                    int K, long long ldy, long long ldx, int splits, int mper) {
  constexpr int TILE = kWgStep * kWgTile * 2;  // 16 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int tiles_k = K / kWgTile;
  const int id = blockIdx.x;
  const int split = id % splits, tile = id / splits;
  const int tn = tile / tiles_k, tk = tile % tiles_k;
  const int n0 = tn * kWgTile, k0 = tk * kWgTile;
  const int wn = wave >> 1, wk = wave & 1;
  const int m_begin = split * mper;
  int m_end = m_begin + mper;
  if (m_end > M) m_end = M;
  const int ntiles = (m_end > m_begin) ? (m_end - m_begin + kWgStep - 1) / kWgStep : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool do_bias = BIAS && (tk == 0);

  Stage64<128> ys, xs;
  const bf16_t* yp = dy + n0;
  const bf16_t* xp = x + k0;
  auto issue = [&](int t) {
    ys.issue(yp, m_begin + t * kWgStep, m_end, ldy, tid);
    xs.issue(xp, m_begin + t * kWgStep, m_end, ldx, tid);
  };
  auto commit = [&](int buf) {
    char* base = smem + buf * 2 * TILE;
    ys.commit(base, tid);
    xs.commit(base + TILE, tid);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < Stage64<128>::NLD; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bsum[2 * j] += lo_bf16(ys.reg[i][j]);
          bsum[2 * j + 1] += hi_bf16(ys.reg[i][j]);
        }
    }
  };
  if (ntiles > 0) {
    issue(0);
    commit(0);
  }
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) issue(t + 1);
    __syncthreads();
    const char* yb = smem + (t & 1) * 2 * TILE;
    const char* xb = yb + TILE;
#pragma unroll
    for (int s = 0; s < kWgStep / 16; ++s) {
      bfrag a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = tr_frag<128>(yb, 0, s, 2 * wn + i, lane);
        b[i] = tr_frag<128>(xb, 0, s, 2 * wk + i, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
    if (t + 1 < ntiles) commit((t + 1) & 1);
  }

  // accumulate the 64x64 wave tile into the fp32 workspace
  const int col = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* base = ws + (long long)(n0 + 64 * wn + 32 * i) * K + k0 + 64 * wk + 32 * j + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* p = base + (long long)acc_row(r, hh) * K;
        if (ATOMIC) atomicAdd(p, acc[i][j][r]);
        else *p += acc[i][j][r];
      }
    }
  if (do_bias) {
    // every thread owns the same 8 columns (chunk tid % 16) in all of its staged rows
    const int c0 = (tid & 15) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(dbias + n0 + c0 + j, bsum[j]);
  }
}

// ------------------------------------------------------------------------------------------
// Second-generation kernel: 128 x 256 output tile, 64-token LDS tiles filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write instructions — at 64 tokens per
// tile the ds_write_b128 path, ~79 B/clk/CU, is what bounded the register-staged kernel
// above). The XOR swizzle that the transposed reads need is applied to the per-lane SOURCE
// address, the LDS image itself is written linearly. Each of the 4 waves owns 128 x 64 of the
// tile (4 x 2 accumulators: 6 operand fragments feed 8 MFMAs per 16-token step). One LDS
// buffer of 48 KiB; two or three workgroups per CU overlap each other's DMA waits.
// Partial results go to per-split fp32 slabs with plain stores (no atomics, bitwise
// reproducible); the finishing pass sums the slabs.
// ------------------------------------------------------------------------------------------
constexpr int kW2N = 128, kW2K = 256;
constexpr int kWg3Lds = (DPA_WG3_RING && DPA_WG3_SLOTS == 5) ? 163840 : 131072;
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool ACCUM>
__global__ __launch_bounds__(256) void wgrad2_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                     float* __restrict__ slabs, int M, int N, int K, long long ldy,
                                                     long long ldx, int splits, int mper) {
  constexpr int SUB = 64 * 128 * 2;  // 16 KiB
  constexpr int STAGE = 3 * SUB;     // 48 KiB: dY sub-tile + two X sub-tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int tiles_k = K / kW2K;
  const int id = blockIdx.x;
  const int split = id % splits, tile = id / splits;
  const int tn = tile / tiles_k, tk = tile % tiles_k;
  const int n0 = tn * kW2N, k0 = tk * kW2K;
  const int m_begin = split * mper;
  int m_end = m_begin + mper;
  if (m_end > M) m_end = M;
  const int ntiles = (m_end > m_begin) ? (m_end - m_begin) / 64 : 0;   // M % 64 == 0 (host-checked)

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  auto issue = [&](int t, int stage) {
    const long long m = m_begin + (long long)t * 64;
    char* base = smem + stage * STAGE;
    dma_rows64x128(dy + m * ldy + n0, ldy, N - n0, base, wave, lane);
    dma_rows64x128(x + m * ldx + k0, ldx, K - k0, base + SUB, wave, lane);
    dma_rows64x128(x + m * ldx + k0 + 128, ldx, K - k0 - 128, base + 2 * SUB, wave, lane);
  };
  auto compute = [&](int stage) {
    const char* ya = smem + stage * STAGE;
    const char* xb = ya + SUB + (wave >> 1) * SUB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bfrag a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tr_frag<128>(ya, 0, s, i, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = tr_frag<128>(xb, 0, s, 2 * (wave & 1) + j, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  };

  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();          // everybody is done reading the previous tile
    issue(t, 0);
    __syncthreads();          // waits for this wave's DMA (vmcnt) and for everybody else's
    compute(0);
  }

  float* out = slabs + (ACCUM ? 0 : (long long)split * N * K);
  const int col = lane & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = k0 + 64 * wave + 32 * j + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * i + acc_row(r, hh);
        if (n < N && kk < K) {
          float* p = out + (long long)n * K + kk;
          if (ACCUM) *p += acc[i][j][r];
          else *p = acc[i][j][r];
        }
      }
    }
}

// 256 x 256 output tile, 4 waves in 2 x 2, each wave 128 x 128 (4 x 4 accumulator tiles in AGPRs).
// Transposed LDS reads deliver half the bytes per cycle of row reads, and with both operands
// token-major every MFMA operand of this product needs one; a 128 x 128 wave tile feeds 16 MFMAs
// from 8 operand fragments (the 128 x 64 wave tile of wgrad2_kernel: 8 from 6), which moves the
// LDS ceiling from about half to about three quarters of the MFMA rate. One workgroup per CU;
// two 64 KiB stages, filled by LDS-DMA issued from inline assembly (see dma_rows64x128_asm in mfma_tiles.h) so that
// the transfer of tile t+1 stays in flight during the MFMAs of tile t.
//
// BIAS: the waves also sum their dY fragments over the tokens on the otherwise idle VALU (each lane adds
// the 8 tokens of its column; the waves that hold the same fragments share the token steps), giving partial
// bias gradients without a second pass over dY.
//
// GROUPED: one product per row segment (the experts of a MoE layer in the padded expert-sorted layout), the
// segment bounds read from device memory. Rows are counted in chunks of `mper`, numbered through the segments
// (segment s owns ceil(rows_s / mper) of them); a work item = (output tile, an equal share of consecutive chunks),
// so every workgroup has the same amount of work whatever the routing did — early in training two of eight
// experts can hold 90 % of the rows. A workgroup walks the segments its share overlaps: one accumulation per
// segment, added into the zero-initialised fp32 [segments, N, K] result with float atomics (one 256 KiB tile per
// workgroup and segment boundary crossed, not one per chunk).
// (one wave per SIMD by construction — 128 KiB of LDS per workgroup —, stated to the register allocator so that all
// 256 accumulators stay in AGPRs in every instantiation)
template <bool ACCUM, bool BIAS, bool GROUPED = false>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad3_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                        float* __restrict__ slabs, float* __restrict__ bias_slabs,
                                                        int M, int N, int K, long long ldy, long long ldx,
                                                        int splits, int mper, const long long* __restrict__ seg_start = nullptr,
                                                        const long long* __restrict__ seg_rows = nullptr, int nseg = 1,
                                                        WgradWalk walk = WgradWalk(), void* direct = nullptr,
                                                        int direct_bf16 = 0) {
  constexpr int SUB = 64 * 128 * 2;  // 16 KiB
  constexpr int STAGE = 4 * SUB;     // 64 KiB: two dY sub-tiles + two X sub-tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int tiles_k = K / 256;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2. Re-label them so that an
  // XCD holds (about) all output tiles of ONE token split: those workgroups walk the same token
  // stages at the same pace and share every dY / X stage through that L2 instead of each pulling
  // its own copy over the fabric (5-8 % on the GPT-2-124M shapes).
  const int ntile = ((N + 255) / 256) * tiles_k;   // N % 128 == 0: the last row of tiles may be half empty
  int split, tile;
  bool atomic_out = GROUPED;
  if (ACCUM) {
    // accumulation into one fp32 result across launches: see WgradWalk
    const WgradItem it = walk.at(blockIdx.x);
    if (it.parts == 0) return;
    tile = it.tile;
    split = it.part;
    atomic_out = it.parts > 1;
    mper = (M / 64 + it.parts - 1) / it.parts * 64;
  } else {
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    split = id / ntile;
    tile = id % ntile;
  }
  int tn, tk;
  wgrad_tile_coord(tile, (N + 255) / 256, tiles_k, tn, tk);
  const int n0 = tn * 256, k0 = tk * 256;
  const int wn = wave >> 1, wk = wave & 1;
  const int ycol1 = n0 + 128 < N ? 128 : 0;   // half-empty last tile: the second dY sub-tile re-reads the first (never stored)
  const bool rows_ok = n0 + 128 * wn < N;     // this wave's 128 output rows exist
  int seg = 0, m_begin = 0, m_end = 0;
  // grouped mode: this workgroup's chunks [c_lo, c_hi) of the chunk numbering that runs through the segments;
  // c0 = number of the first chunk of segment `seg`, seg_nch = chunks of that segment
  int c_lo = 0, c_hi = 0, c0 = 0, seg_nch = 0;
  if (GROUPED) {
    int total = 0;
    for (int s = 0; s < nseg; ++s) total += ((int)seg_rows[s] + mper - 1) / mper;
    const int nitems = gridDim.x / ntile;
    const int cper = (total + nitems - 1) / nitems;
    c_lo = split * cper;
    c_hi = c_lo + cper < total ? c_lo + cper : total;
    split = 0;
    if (c_lo >= c_hi) return;
  } else {
    m_begin = split * mper;
    m_end = m_begin + mper;
    if (m_end > M) m_end = M;
  }
  for (;;) {   // one pass per row range: the only one in the dense modes, one per overlapped segment in grouped mode
  if (GROUPED) {
    bool found = false;
    for (; seg < nseg; ++seg) {
      const int len = (int)seg_rows[seg];
      seg_nch = (len + mper - 1) / mper;
      const int lo = c_lo > c0 ? c_lo : c0, hi = c_hi < c0 + seg_nch ? c_hi : c0 + seg_nch;
      if (lo < hi) {
        const int base = (int)seg_start[seg];
        m_begin = base + (lo - c0) * mper;
        m_end = base + (hi - c0) * mper;
        if (m_end > base + len) m_end = base + len;
        found = true;
        break;
      }
      c0 += seg_nch;
      if (c0 >= c_hi) break;
    }
    if (!found) break;
  }
  const int ntiles = (m_end > m_begin) ? (m_end - m_begin) / 64 : 0;

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  // bias gradient: the 2 * tiles_k waves that hold the same dY fragments (both wk of every K-tile column) take
  // the token steps in turn, so the column sums cost every wave a sixth (K = 768) of what one wave would pay
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const int bias_q = 2 * tiles_k, bias_me = 2 * tk + wk;
  int bias_cnt = bias_me;      // token steps until this wave's next turn

  const unsigned lds0 = (unsigned)(size_t)(dma_lptr_t)smem;
#if DPA_WG3_RING
  // Ring of four 32-token slots (32 KiB each: two dY and two X sub-tiles of 32 x 128). Slot t+1 is published (its
  // transfers waited for, then the workgroup barrier) in the MIDDLE of the work on slot t, and the operand
  // fragments of a token step are requested one step ahead, across slots: the MFMA stream never waits for a
  // slot to land or for a first LDS read. Transfers run three slots ahead (up to 96 KiB in flight).
  constexpr int RSUB = 32 * 128 * 2, RSLOT = 4 * RSUB, NSLOT = DPA_WG3_SLOTS, AHEAD = NSLOT - 1;
  const int nst = 2 * ntiles;      // 32-token stages
  // Transfers: the row block's base address and the LDS destination are scalars, the lane's byte offsets inside a
  // 32-row block are computed once (rows piece * 4 + lane / 16, swizzled 16-byte chunk lane % 16).
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  unsigned voff[4][2];   // sub-tile q (dY columns 0 / ycol1, X columns 0 / 128), piece pc
#pragma unroll
  for (int pc = 0; pc < 2; ++pc) {
    const int row = (wave_s * 2 + pc) * 4 + (lane >> 4);
    const int chunk = swz<128>(row, lane & 15);
    const unsigned vy = (unsigned)(((long long)row * ldy + chunk * 8) * 2), vx = (unsigned)(((long long)row * ldx + chunk * 8) * 2);
    voff[0][pc] = vy;
    voff[1][pc] = vy + (unsigned)ycol1 * 2u;
    voff[2][pc] = vx;
    voff[3][pc] = vx + 256u;
  }
  // row bases of the stage that is transferred next: advanced by 32 rows per stage and held at the last stage (the
  // transfers past the end re-read it and are never used), so the loop has no 64-bit multiply
  const bf16_t* gy = dy + (long long)m_begin * ldy + n0;
  const bf16_t* gx = x + (long long)m_begin * ldx + k0;
  int t_dma = 0;   // stage the bases point at
  auto issue_sub = [&](int slot, int q) {   // sub-tile q of stage t_dma into ring slot `slot`
    const unsigned base = lds0 + slot * RSLOT + q * RSUB + wave_s * 2048;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) dma_piece_sbase(q < 2 ? gy : gx, voff[q][pc], base + pc * 1024);
  };
  auto next_stage = [&]() {
    if (t_dma < nst - 1) {
      gy += 32 * ldy;
      gx += 32 * ldx;
    }
    ++t_dma;
  };
  auto issue = [&](int slot) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_sub(slot, q);
    next_stage();
  };
  // Fragment reads: the lane's LDS addresses of the eight half-fragments of each operand inside slot 0 (k-step 0) are
  // computed once; a slot's offset is added once per slot (16 additions), the second k-step of a slot is the
  // instruction's offset field (16 rows x 256 bytes).
  constexpr int KSTEP = 16 * 128 * 2;
  unsigned pa[4][2], pb[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const unsigned o = tr_frag_offset<128>(dt, jj, lane);
      pa[dt][jj] = lds0 + wn * RSUB + o;
      pb[dt][jj] = lds0 + (2 + wk) * RSUB + o;
    }
  auto bias_step = [&](const bfrag (&a)[4]) {
    if (BIAS) {
      if (bias_cnt == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 v = __builtin_bit_cast(u32x4, a[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[i] += lo_bf16(v[e]) + hi_bf16(v[e]);
        }
        bias_cnt = bias_q;
      }
      --bias_cnt;
    }
  };
  // One token step: the 16 MFMAs on (a, b) in four groups of four; behind group g go the reads of one A and one B
  // fragment of the next step (into an, bn) and, with DMA, the transfer of sub-tile g of stage `tdma`. The
  // sched_group_barrier sequence pins that order (left alone the scheduler sinks every read down to its first use).
  // (fa, fb: the address sets of the slot the NEXT step's fragments come from, imm: 0 or KSTEP)
  auto step = [&](const bfrag (&a)[4], const bfrag (&b)[4], bfrag (&an)[4], bfrag (&bn)[4], const unsigned (&fa)[4][2],
                  const unsigned (&fb)[4][2], int imm, bool dma, int sdma) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[g][j] = mfma32(a[g], b[j], acc[g][j]);
      // read in the order of first use by the next step: a0 b0 | b1 b2 | b3 a1 | a2 a3
      if (g == 0) { an[0] = tr_frag_at(fa[0], imm); bn[0] = tr_frag_at(fb[0], imm); }
      if (g == 1) { bn[1] = tr_frag_at(fb[1], imm); bn[2] = tr_frag_at(fb[2], imm); }
      if (g == 2) { bn[3] = tr_frag_at(fb[3], imm); an[1] = tr_frag_at(fa[1], imm); }
      if (g == 3) { an[2] = tr_frag_at(fa[2], imm); an[3] = tr_frag_at(fa[3], imm); }
      if (dma) issue_sub(sdma, g);
    }
    if (dma) next_stage();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 LDS read
    }
    bias_step(a);
  };
  if (nst > 0) {
    bfrag a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) issue(i);   // (stages 0 .. AHEAD - 1 into slots of the same numbers)
    wait_vm<8 * (AHEAD - 1)>();                         // stage 0 (8 transfers per wave and stage, retired in order)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    unsigned xa[4][2], xb[4][2], ya[4][2], yb[4][2];    // address sets of two slots, used in turn
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        xa[dt][jj] = pa[dt][jj];
        xb[dt][jj] = pb[dt][jj];
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = tr_frag_at(xa[i], 0);
      b0[i] = tr_frag_at(xb[i], 0);
    }
    int cur = 0, nxt = 1;   // ring slots of stages t and t + 1 (kept as running indices: no division in the loop)
    // one stage: `ca, cb` address the slot of stage t, `na, nb` receive the addresses of stage t + 1's slot
    auto stage = [&](int t, const unsigned (&ca)[4][2], const unsigned (&cb)[4][2], unsigned (&na)[4][2], unsigned (&nb)[4][2]) {
      step(a0, b0, a1, b1, ca, cb, KSTEP, false, 0);
      // stage t+1 has landed; the barrier also proves that every wave is done reading slot t-1, which stage t+3 overwrites
      wait_vm<8 * (AHEAD - 2)>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned off = (unsigned)nxt * RSLOT;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          na[dt][jj] = pa[dt][jj] + off;
          nb[dt][jj] = pb[dt][jj] + off;
        }
      const int sd = cur == 0 ? NSLOT - 1 : cur - 1;   // (t + AHEAD) % NSLOT with AHEAD = NSLOT - 1
      step(a1, b1, a0, b0, na, nb, 0, true, sd);
      cur = nxt;
      nxt = nxt + 1 == NSLOT ? 0 : nxt + 1;
    };
    int t = 0;
    for (; t + 1 < nst; t += 2) {
      stage(t, xa, xb, ya, yb);
      stage(t + 1, ya, yb, xa, xb);
    }
    if (t < nst) stage(t, xa, xb, ya, yb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the transfers past the end
  }
#else
  auto issue = [&](int t, int stage) {
    const long long m = m_begin + (long long)t * 64;
    const unsigned base = lds0 + stage * STAGE;
    dma_rows64x128_asm(dy + m * ldy + n0, ldy, 128, base, wave, lane);
    dma_rows64x128_asm(dy + m * ldy + n0 + ycol1, ldy, 128, base + SUB, wave, lane);
    dma_rows64x128_asm(x + m * ldx + k0, ldx, 128, base + 2 * SUB, wave, lane);
    dma_rows64x128_asm(x + m * ldx + k0 + 128, ldx, 128, base + 3 * SUB, wave, lane);
  };
  if (ntiles > 0) issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed (this wave's DMAs drained, then everybody's through the barrier); the same
    // barrier proves that every wave is done reading the stage that tile t+1 is about to overwrite
#if DPA_WG3_ABLATE != 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#if DPA_WG3_ABLATE != 2
    if (t + 1 < ntiles) issue(t + 1, (t + 1) & 1);
#endif
#if DPA_WG3_ABLATE == 3
    continue;
#endif
    const char* ya = smem + (t & 1) * STAGE + wn * SUB;
    const char* xb = smem + (t & 1) * STAGE + (2 + wk) * SUB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bfrag a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tr_frag<128>(ya, 0, s, i, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = tr_frag<128>(xb, 0, s, j, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
      if (BIAS) {
        if (bias_cnt == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const u32x4 v = __builtin_bit_cast(u32x4, a[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[i] += lo_bf16(v[e]) + hi_bf16(v[e]);
          }
          bias_cnt = bias_q;
        }
        --bias_cnt;
      }
    }
  }
#endif
  if (BIAS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      if (hh == 0 && rows_ok) bias_slabs[((long long)split * bias_q + bias_me) * N + n0 + 128 * wn + 32 * i + (lane & 31)] = tot;
    }
  }

  float* out = slabs + (GROUPED ? (long long)seg * N * K : (ACCUM ? 0 : (long long)split * N * K));
  const int col = lane & 31;
  if (rows_ok) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + 128 * wk + 32 * j + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 128 * wn + 32 * i + acc_row(r, hh);
        float* p = out + (long long)n * K + kk;
        if (GROUPED) unsafeAtomicAdd(p, acc[i][j][r]);
        else if (ACCUM) {
          if (atomic_out) unsafeAtomicAdd(p, acc[i][j][r]);
          else *p += acc[i][j][r];
        } else if (direct != nullptr) {
          // one split: the gradient itself is the destination (bf16 or fp32, added to), no slab and no fold
          const long long o = (long long)n * K + kk;
          if (direct_bf16) {
            bf16_t* d = (bf16_t*)direct + o;
            *d = f32_to_bf16(bf16_to_f32(*d) + acc[i][j][r]);
          } else {
            ((float*)direct)[o] += acc[i][j][r];
          }
        } else *p = acc[i][j][r];
      }
    }
  }
  if (!GROUPED) break;
  // next segment; every wave is done with the LDS stages before the next range's first transfer lands in them
  c0 += seg_nch;
  ++seg;
  __builtin_amdgcn_s_barrier();
  }
}

// out (bf16 or fp32) = [out +] sum over `nslabs` slabs of ws
template <typename T>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ ws, T* __restrict__ out,
                                                           long long nvec, int nslabs, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    // Eight slabs are requested together: a small weight has fewer vectors than the chip has threads and up to 64
    // slabs, and one load per round trip made the fold a chain of memory latencies (19 us for 65 MB).
    float a[8];
    load8(ws + i * 8, a);
    int s = 1;
    for (; s + 8 <= nslabs; s += 8) {
      float b[8][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) load8(ws + (long long)(s + u) * nvec * 8 + i * 8, b[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[u][j];
    }
    if (s + 4 <= nslabs) {
      float b[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) load8(ws + (long long)(s + u) * nvec * 8 + i * 8, b[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[u][j];
      s += 4;
    }
    {
      float b[3][8];   // the last 0-3 slabs, requested together as well
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (s + u < nslabs) load8(ws + (long long)(s + u) * nvec * 8 + i * 8, b[u]);
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (s + u < nslabs)
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += b[u][j];
    }
    if (accumulate) {
      float o[8];
      load8(out + i * 8, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += o[j];
    }
    store8(out + i * 8, a);
  }
}

// The same fold for a batch of equally sized destinations given by address (the experts of a MoE
// layer live at different places of the flat gradient buffer): out_g (+)= ws[g], one launch.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_finish_many_kernel(const float* __restrict__ ws,
                                                                const long long* __restrict__ dst_ptrs,
                                                                long long nvec, int accumulate) {
  const float* src = ws + (long long)blockIdx.y * nvec * 8;
  T* out = reinterpret_cast<T*>(dst_ptrs[blockIdx.y]);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float a[8];
    load8(src + i * 8, a);
    if (accumulate) {
      float o[8];
      load8(out + i * 8, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += o[j];
    }
    store8(out + i * 8, a);
  }
}

void wgrad_finish_many(const float* ws, const long long* dst_ptrs, int count, long long n, bool accumulate,
                       bool out_bf16, hipStream_t st) {
  const long long nvec = n >> 3;
  if (nvec == 0 || count == 0) return;
  long long blocks = (nvec + 255) / 256;
  if (blocks > 512) blocks = 512;
  const dim3 grid((unsigned)blocks, (unsigned)count), block(256);
  if (out_bf16)
    hipLaunchKernelGGL((wgrad_finish_many_kernel<bf16_t>), grid, block, 0, st, ws, dst_ptrs, nvec, accumulate ? 1 : 0);
  else
    hipLaunchKernelGGL((wgrad_finish_many_kernel<float>), grid, block, 0, st, ws, dst_ptrs, nvec, accumulate ? 1 : 0);
}

int wgrad_splits(int M, int N, int K) {
  const int tiles = (N / kWgTile) * (K / kWgTile);
  int s = (512 + tiles - 1) / tiles;          // aim at >= 2 workgroups per CU
  const int max_s = (M + 4 * kWgStep - 1) / (4 * kWgStep);   // at least 4 LDS tiles per workgroup
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

void wgrad_gemm(const void* dy, const void* x, float* ws, float* dbias, int M, int N, int K, long long ldy,
                long long ldx, int splits, hipStream_t st) {
  if (M == 0) return;
  int mper = (M + splits - 1) / splits;
  mper = (mper + kWgStep - 1) / kWgStep * kWgStep;
  const dim3 grid((N / kWgTile) * (K / kWgTile) * splits), block(256);
  const size_t lds = 4 * kWgStep * kWgTile * 2;  // 64 KiB
#define DPA_WG(A, B)                                                                                            \
  hipLaunchKernelGGL((wgrad_kernel<A, B>), grid, block, lds, st, (const bf16_t*)dy, (const bf16_t*)x, ws, dbias, M, \
                     N, K, ldy, ldx, splits, mper)
  if (splits > 1) {
    if (dbias) DPA_WG(true, true); else DPA_WG(true, false);
  } else {
    if (dbias) DPA_WG(false, true); else DPA_WG(false, false);
  }
#undef DPA_WG
}

bool wgrad2_ok(int M, int N, int K) { return M % 64 == 0 && M >= 64 && N % 8 == 0 && K % kW2K == 0; }

int g_wgrad2_stages = 4;   // 4 = 256x256 kernel where the shape allows, else (and with 1) the 128x256 kernel (benchmark knob)

static bool use_wgrad3(int M, int N, int K) {
  return g_wgrad2_stages == 4 && N % 128 == 0 && K % 256 == 0 && M >= 1024;
}

int g_wgrad2_force_splits = 0;   // benchmark knob: > 0 overrides the split count of the 256x256 kernel
void wgrad2_force_splits(int s) { g_wgrad2_force_splits = s; }

int wgrad2_splits(int M, int N, int K) {
  if (use_wgrad3(M, N, K) && g_wgrad2_force_splits > 0) return g_wgrad2_force_splits;
  if (use_wgrad3(M, N, K)) {
    // Row splits of the 256 x 256 kernel: the count that minimises (rounds of 256 workgroups) x (time of one split of
    // a tile) + (traffic of the per-split fp32 slabs and their fold). 192 tiles (dW[6144, 2048]) run as 4 splits =
    // 3 full rounds instead of one round on three quarters of the CUs; 36 tiles as 7 splits; 256 tiles as one.
    const int t3 = ((N + 255) / 256) * (K / 256);
    int smax = M / 512;                  // at least 16 LDS slots per workgroup
    if (smax > 64) smax = 64;
    if (smax < 1) smax = 1;
    const double tile_s = 2.0 * M * 256.0 * 256.0 / 4.5e12;            // one tile over all rows on one CU (1150 TF chip-wide)
    const double slab_s = 2.0 * 4.0 * (double)N * (double)K / 4.0e12;  // a slab written and read back at 4 TB/s
    int best = 1;
    double best_t = 1e30;
    for (int sp = 1; sp <= smax; ++sp) {
      const int rounds = (t3 * sp + 255) / 256;
      const double t = rounds * tile_s / sp + sp * slab_s;
      if (t < best_t * 0.995) {          // ties go to the smaller count
        best_t = t;
        best = sp;
      }
    }
    return best;
  }
  const int tiles = ((N + kW2N - 1) / kW2N) * (K / kW2K);
  int s = 512 / tiles;   // one round of the workgroups that fit (two per CU)
  const int max_s = M / 256;           // at least 4 LDS tiles per workgroup
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}

// splits > 0: writes `splits` slabs (ws holds splits*N*K floats). splits == 0: a single pass
// over all tokens that ADDS into ws[N*K] (accumulation across calls, e.g. lm_head row chunks).
void wgrad2_set_stages(int s) { g_wgrad2_stages = s; }

bool wgrad2_fuses_bias(int M, int N, int K) { return use_wgrad3(M, N, K); }
// rows of the fp32 [rows, N] partial bias sums the fused form writes (every one is written: sum them all)
int wgrad2_bias_slabs(int M, int N, int K) { return wgrad2_splits(M, N, K) * 2 * (K / 256); }

bool wgrad2_direct_ok(int M, int N, int K) {
  return use_wgrad3(M, N, K) && wgrad2_splits(M, N, K) == 1;
}

void wgrad2_gemm(const void* dy, const void* x, float* ws, int M, int N, int K, long long ldy, long long ldx,
                 int splits, hipStream_t st, float* bias_slabs, void* direct, bool direct_bf16) {
  if (M == 0) return;
  const int sp = splits > 0 ? splits : 1;
  int mper = (M / 64 + sp - 1) / sp * 64;
  if (use_wgrad3(M, N, K)) {
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute((const void*)wgrad3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kWg3Lds);
      (void)hipFuncSetAttribute((const void*)wgrad3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kWg3Lds);
      (void)hipFuncSetAttribute((const void*)wgrad3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kWg3Lds);
      attr3 = true;
    }
    const int tiles = ((N + 255) / 256) * (K / 256);
    const dim3 grid3(tiles * sp), block3(256);
#define DPA_W3(ACC, BI)                                                                                          \
  hipLaunchKernelGGL((wgrad3_kernel<ACC, BI>), grid3, block3, kWg3Lds, st, (const bf16_t*)dy, (const bf16_t*)x, ws, \
                     bias_slabs, M, N, K, ldy, ldx, sp, mper, (const long long*)nullptr, (const long long*)nullptr, 1, \
                     WgradWalk(), splits == 1 ? direct : nullptr, direct_bf16 ? 1 : 0)
    if (splits > 0) {
      if (bias_slabs != nullptr) DPA_W3(false, true); else DPA_W3(false, false);
    } else {
      WgradWalk walk;
      walk.tiles = tiles;
      walk.max_parts = M / 2048 < 1 ? 1 : (M / 2048 > 8 ? 8 : M / 2048);
      hipLaunchKernelGGL((wgrad3_kernel<true, false>), dim3(walk.grid()), block3, kWg3Lds, st, (const bf16_t*)dy,
                         (const bf16_t*)x, ws, (float*)nullptr, M, N, K, ldy, ldx, 1, mper, (const long long*)nullptr,
                         (const long long*)nullptr, 1, walk);
    }
#undef DPA_W3
    return;
  }
  const dim3 grid(((N + kW2N - 1) / kW2N) * (K / kW2K) * sp), block(256);
  const size_t lds = (size_t)3 * 64 * 128 * 2;  // 48 KiB: two workgroups per CU hide each other's DMA latency
  if (splits > 0)
    hipLaunchKernelGGL((wgrad2_kernel<false>), grid, block, lds, st, (const bf16_t*)dy, (const bf16_t*)x, ws, M, N, K,
                       ldy, ldx, sp, mper);
  else
    hipLaunchKernelGGL((wgrad2_kernel<true>), grid, block, lds, st, (const bf16_t*)dy, (const bf16_t*)x, ws, M, N, K,
                       ldy, ldx, sp, mper);
}

constexpr int kGroupedChunk = 256;   // granularity of the row split (4 LDS tiles); a workgroup takes a run of chunks

bool wgrad3_grouped_ok(long long rows, int N, int K) { return rows % 64 == 0 && rows > 0 && N % 256 == 0 && K % 256 == 0; }

// out: fp32 [nseg, N, K], zero-initialised by the caller.
void wgrad3_grouped(const void* dy, const void* x, float* out, const long long* seg_start, const long long* seg_rows,
                    int nseg, int rows, int N, int K, long long ldy, long long ldx, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad3_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kWg3Lds);
    attr = true;
  }
  // one workgroup per CU: row shares so that tiles * shares fills the 256 CUs once (at least 1024 rows per share)
  const int tiles = (N / 256) * (K / 256);
  int items = 256 / tiles;
  const int max_items = rows / 1024;
  if (items > max_items) items = max_items;
  if (items < 1) items = 1;
  const dim3 grid(tiles * items), block(256);
  hipLaunchKernelGGL((wgrad3_kernel<false, false, true>), grid, block, kWg3Lds, st, (const bf16_t*)dy, (const bf16_t*)x, out,
                     (float*)nullptr, rows, N, K, ldy, ldx, 1, kGroupedChunk, seg_start, seg_rows, nseg);
}

void wgrad_finish(const float* ws, void* out, long long n, int nslabs, bool accumulate, bool out_bf16,
                  hipStream_t st) {
  const long long nvec = n >> 3;
  if (nvec == 0) return;
  long long blocks = (nvec + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (out_bf16)
    hipLaunchKernelGGL((wgrad_finish_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, ws, (bf16_t*)out, nvec,
                       nslabs, accumulate ? 1 : 0);
  else
    hipLaunchKernelGGL((wgrad_finish_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, ws, (float*)out, nvec,
                       nslabs, accumulate ? 1 : 0);
}

}  // namespace dpa
