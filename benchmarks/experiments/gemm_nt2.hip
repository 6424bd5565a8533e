// NT GEMM with fused epilogues, two-workgroups-per-CU form for gfx950:
//   C[M, N] = epi( A[M, K] * B[N, K]^T [+ bias] )   (bf16 in, fp32 accumulate, bf16 out)
//
// csrc/gemm_nt.hip runs one 512-thread workgroup per CU on a 256 x 256 tile; its epilogue (conversion,
// activation arithmetic, stores) is serial with its K loop, which on the K = 768 products of a
// GPT-2-sized layer is 25-45 % of a tile's time (twelve K-tiles per 128 KiB of output). This kernel
// halves the workgroup instead of shortening the epilogue: 256 threads = 4 waves (one per SIMD) own a
// 128 x 256 tile out of 80 KiB of LDS, so TWO workgroups are resident per CU and, started half a tile
// apart, one runs its K loop on the matrix pipe while the other converts and stores. The per-wave work
// is unchanged (128 x 64 outputs in 32 accumulators of v_mfma_f32_16x16x32_bf16, transposed product so
// that a lane owns 16 consecutive output columns), so the LDS fragment traffic per FLOP is the same;
// the LDS-DMA traffic per FLOP is 1.5x (384 operand rows per 128 x 256 tile instead of 512 per 256 x 256).
//
// Operand ring: nine 8 KiB blocks (64 rows x 64 k, 128-byte rows, 16-byte chunks XOR-swizzled by the row
// exactly as in gemm_nt.hip). A K-tile is six blocks in the fixed order
//     A0 (rows 0-63), A1 (rows 64-127), B00, B01 (the first 32 columns of wave columns 0-1 / 2-3),
//     B10, B11 (their other 32 columns),
// consumed in two phases of 32 MFMAs: P0 reads A0, A1 and a B0x block (20 fragment reads; the sixteen A
// fragments stay in registers), P1 reads a B1x block (4 reads). One barrier per phase: a block is waited
// for (counted vmcnt, never 0) at the end of the phase before its first read and refilled right after
// the barrier that follows its last read. The refill stream runs on across output tiles; every block
// is issued two phases ahead of its first read except B01, for which the ring is one block short
// (issued one phase ahead; weights, served from L2).
//
// MEASURED AND NOT USED (kept as an experiment, not part of the extension): on MI355X this form is 15-25 % slower than
// gemm_nt.hip on every layer shape (1046 vs 1254 TF on [65536, 3072, 768]; GELU form 782 vs 873 TF), and the epilogue
// is as exposed as before (+133 us for the GELU form against +108 us). The two waves of a SIMD share its issue
// bandwidth: the K loop's own address / LDS-DMA / wait instructions plus the epilogue's vector instructions fill it,
// so "matrix pipe of one workgroup under the vector work of the other" does not materialise; the barrier-locked
// ping-pong of the two wave groups in gemm_nt.hip keeps the matrix pipe 72 % busy, the free-running pair here 56 %.
// Numbers: profiles/r3_gemm_nt_two_wg_per_cu_experiment.txt. Build with the driver:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Icsrc -DDPA_WITH_FORM2 benchmarks/gemm_nt_driver.hip csrc/gemm_nt.hip \
//         benchmarks/experiments/gemm_nt2.hip -o benchmarks/gemm_nt_driver.bin
#include <type_traits>

#include "common.h"
#include "host_logic.h"
#include "kernels.h"

namespace dpa {
bool gemm_nt2_ok(int epi, bool grouped);
void gemm_nt2(GemmNtArgs a, int epi, int num_cus, hipStream_t st);
void gemm_nt2_set_skew(int ticks);
}

namespace dpa {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bfrag8;
typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;

#ifndef DPA_GEMM2_STORE_AUX
#define DPA_GEMM2_STORE_AUX 2   // non-temporal epilogue stores, as in gemm_nt.hip
#endif
#ifndef DPA_GEMM2_SPREAD_ISSUE
#define DPA_GEMM2_SPREAD_ISSUE 1   // LDS-DMA pieces of a phase issued between its MFMA groups instead of ahead of them
#endif
constexpr int kStoreAux2 = DPA_GEMM2_STORE_AUX;
constexpr bool kSpreadIssue = DPA_GEMM2_SPREAD_ISSUE != 0;
constexpr int kBlk = 8192;                    // one ring block: 64 rows x 64 k x 2 bytes
constexpr int kRing = 9;
constexpr int kXpose2 = kRing * kBlk;         // 2 KiB per wave: one 16-row x 64-column bf16 piece of the output tile
constexpr int kLds2 = kXpose2 + 4 * 2048;     // 80 KiB: two workgroups per CU

__device__ __forceinline__ f32x4 mfma16(bfrag8 a, bfrag8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

// One LDS-DMA piece: 64 lanes x 16 bytes from rs[voff] to LDS [lds_addr, lds_addr + 1024).
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, rsrc_t rs) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(rs)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// Exact-erf GELU and its derivative (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7), two values at a time.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void gelu_and_grad2(f32x2 x, f32x2& act, f32x2& dact) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 one = {1.0f, 1.0f};
  const f32x2 u = pk_fma(ax, f32x2{0.3275911f * 0.70710678f, 0.3275911f * 0.70710678f}, one);
  const f32x2 t = {__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
  const f32x2 xx = x * x * f32x2{-0.5f * 1.4426950409f, -0.5f * 1.4426950409f};
  const f32x2 e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  f32x2 poly = pk_fma(t, f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
  poly = pk_fma(poly, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
  poly = pk_fma(poly, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
  poly = pk_fma(poly, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
  const f32x2 h = poly * t * e;
  const f32x2 hm = one - h;
  const f32x2 phi = {x[0] >= 0.f ? hm[0] : h[0], x[1] >= 0.f ? hm[1] : h[1]};
  act = x * phi;
  dact = pk_fma(x * e, f32x2{0.3989422804f, 0.3989422804f}, phi);
}

}  // namespace

template <int EPI, bool BIAS>
__global__ __launch_bounds__(256, 2) void gemm_nt2_kernel(GemmNtArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = wave column (64 output columns)
  const int q = lane >> 4, l15 = lane & 15;
  const int nk = p.K / 64;

  const rsrc_t rsA = make_rsrc(p.a, p.a_bytes);
  const rsrc_t rsB = make_rsrc(p.b, p.b_bytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

  // ---- LDS-DMA source offsets of this lane (bytes inside the 128-row A / 256-row B panel of a tile) ----
  // Piece pc = 2 * wave + pp of a block covers block rows 8 pc .. 8 pc + 7; lane l -> row 8 pc + (l >> 3),
  // chunk slot l & 7, which must receive logical chunk (l & 7) ^ ((row >> 1) & 7).
  unsigned voffA0, voffA1, voffB0, voffB1, pieceLds0, pieceLds1;
  {
    auto piece = [&](int pp, unsigned& va, unsigned& vb, unsigned& pl) {
      const int pc = 2 * wc + pp;
      const int rho = 8 * pc + (lane >> 3);
      const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
      // A block A{mh}: block row rho <-> tile row 64 mh + rho
      va = (unsigned)rho * (unsigned)p.lda * 2u + chunk * 16;
      // B block B{nh}{half}: block row rho = 32 (wave column & 1) + 16 nt + i <-> tile column
      // 128 half + 64 (rho >> 5) + 16 (i >> 2) + 4 (2 nh + nt) + (i & 3): a lane of the consuming wave ends up with
      // 16 consecutive output columns
      const int i = rho & 15, nt = (rho >> 4) & 1, wcs = rho >> 5;
      const int colB = 64 * wcs + 16 * (i >> 2) + 4 * nt + (i & 3);
      vb = (unsigned)colB * (unsigned)p.ldb * 2u + chunk * 16;
      pl = lds0 + pc * 1024;
    };
    piece(0, voffA0, voffB0, pieceLds0);
    piece(1, voffA1, voffB1, pieceLds1);
    asm volatile("" : "+v"(voffA0), "+v"(voffA1), "+v"(voffB0), "+v"(voffB1));
    pieceLds0 = __builtin_amdgcn_readfirstlane(pieceLds0);
    pieceLds1 = __builtin_amdgcn_readfirstlane(pieceLds1);
  }
  const unsigned aStepMh = 64u * (unsigned)p.lda * 2u;     // A1 rows = A0 rows + 64
  const unsigned bStepNh = 8u * (unsigned)p.ldb * 2u;      // B1x columns = B0x columns + 8
  const unsigned bStepHalf = 128u * (unsigned)p.ldb * 2u;  // Bx1 columns = Bx0 columns + 128

  // ---- fragment read offsets (bytes inside a block) ---------------------------------------------------
  unsigned aoff0, aoff1, boff0, boff1;
  {
    const int rhoA = l15, rhoB = 32 * (wc & 1) + l15;
    aoff0 = rhoA * 128 + (((0 + q) ^ ((rhoA >> 1) & 7)) << 4);
    aoff1 = rhoA * 128 + (((4 + q) ^ ((rhoA >> 1) & 7)) << 4);
    boff0 = rhoB * 128 + (((0 + q) ^ ((rhoB >> 1) & 7)) << 4);
    boff1 = rhoB * 128 + (((4 + q) ^ ((rhoB >> 1) & 7)) << 4);
  }
  const int bhalf = wc >> 1;   // which of the Bx0 / Bx1 blocks this wave reads

  // The two workgroups of a CU should run half a tile apart (one in its K loop, the other in its epilogue).
  // Workgroups are dealt to the CUs in rounds, so the second half of the grid is (in practice; nothing
  // depends on it) the second round: it starts late by about half a tile's time.
  if (p.skew_ticks > 0 && blockIdx.x >= (gridDim.x >> 1)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.skew_ticks) __builtin_amdgcn_s_sleep(32);
  }
  GemmWalk walk{p.tiles_m, p.tiles_n, (int)gridDim.x, 128, 8};
  const int wg = blockIdx.x;
  const int ntiles = walk.count(wg);
  if (ntiles == 0) return;

  int cur_m0, cur_n0, nxt_m0, nxt_n0;
  {
    const TileCoord t0 = walk.at(wg, 0);
    cur_m0 = t0.m0;
    cur_n0 = t0.n0;
    nxt_m0 = cur_m0;
    nxt_n0 = cur_n0;
  }
  bool have_nxt = ntiles > 1;
  if (have_nxt) {
    const TileCoord t1 = walk.at(wg, 1);
    nxt_m0 = t1.m0;
    nxt_n0 = t1.n0;
  }
  const unsigned kOob = 0x80000000u;   // a dead refill (past the last tile) reads zeros that nobody reads

  // ---- refill stream -----------------------------------------------------------------------------------
  // Blocks are issued strictly in ring order, so the write slot is a running counter (scalar registers).
  unsigned wslot = 0, rslot = 0;
  // TY: 0 = A0, 1 = A1, 2 = B00, 3 = B01, 4 = B10, 5 = B11 of K-tile kk (kk may run into the next output tile)
  auto issue = [&](auto type_tag, int kk) {
    constexpr int TY = decltype(type_tag)::value;
    constexpr bool IS_A = TY < 2;
    const bool over = kk >= nk;
    const unsigned k = (unsigned)(over ? kk - nk : kk) * 128u;
    unsigned base;
    if (IS_A) base = (unsigned)(over ? nxt_m0 : cur_m0) * (unsigned)p.lda * 2u + (TY == 1 ? aStepMh : 0u) + k;
    else
      base = (unsigned)(over ? nxt_n0 : cur_n0) * (unsigned)p.ldb * 2u + (((TY - 2) >> 1) ? bStepNh : 0u) +
             (((TY - 2) & 1) ? bStepHalf : 0u) + k;
    if (over && !have_nxt) base = kOob;
    const unsigned ldsu = wslot * kBlk;
    wslot = wslot == kRing - 1 ? 0u : wslot + 1u;
    const rsrc_t rs = IS_A ? rsA : rsB;
    dma16(pieceLds0 + ldsu, base + (IS_A ? voffA0 : voffB0), rs);
    dma16(pieceLds1 + ldsu, base + (IS_A ? voffA1 : voffB1), rs);
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>;
  using T4 = std::integral_constant<int, 4>;
  using T5 = std::integral_constant<int, 5>;

  // Epilogue rows regrouped through a wave-private 2 KiB LDS buffer (see gemm_nt.hip): every 8 consecutive
  // lanes store (load) one whole 128-byte line.
  char* const tbuf = smem + kXpose2 + wc * 2048;
  const unsigned tw0 = l15 * 128 + (((2 * q) ^ (l15 & 7)) << 4), tw1 = l15 * 128 + (((2 * q + 1) ^ (l15 & 7)) << 4);
  const int trow = lane >> 3, tch = lane & 7;
  const unsigned tr0 = trow * 128 + ((tch ^ (trow & 7)) << 4), tr1 = tr0 + 1024;
  auto store_rows16 = [&](__amdgpu_buffer_rsrc_t rs, u32x4 lo, u32x4 hi, int row0, int col0, int lim, unsigned ld) {
    *reinterpret_cast<u32x4*>(tbuf + tw0) = lo;
    *reinterpret_cast<u32x4*>(tbuf + tw1) = hi;
    asm volatile("" ::: "memory");
    const u32x4 st1 = *reinterpret_cast<const u32x4*>(tbuf + tr0), st2 = *reinterpret_cast<const u32x4*>(tbuf + tr1);
    asm volatile("" ::: "memory");
    const int r1 = row0 + trow, r2 = r1 + 8;
    const bool cok = 8 * tch < lim;
    const unsigned cb = ((unsigned)col0 + 8u * tch) * 2u;
    const unsigned o1 = (cok && r1 < p.M) ? (unsigned)r1 * ld * 2u + cb : 0x80000000u;
    const unsigned o2 = (cok && r2 < p.M) ? (unsigned)r2 * ld * 2u + cb : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(st1, rs, o1, 0, kStoreAux2);
    __builtin_amdgcn_raw_buffer_store_b128(st2, rs, o2, 0, kStoreAux2);
  };

  f32x4 acc[2][4][4];
  bfrag8 af[2][4][2] = {}, bf[2][2] = {};

  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto ring = [](unsigned s, unsigned d) -> unsigned { return s + d >= (unsigned)kRing ? s + d - kRing : s + d; };
  // Fragment reads in the order the MFMAs consume them (k-step major), so that the first MFMA group waits for the
  // first reads only and the rest of the reads travel under it.
  auto read_a_ks = [&](int ks) {
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      const char* u = smem + ring(rslot, mh) * kBlk;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        af[mh][mt][ks] = *reinterpret_cast<const bfrag8*>(u + (ks ? aoff1 : aoff0) + mt * 2048);
    }
  };
  auto read_b_ks = [&](int nh, int ks) {
    const char* u = smem + ring(rslot, 2 + 2 * nh + bhalf) * kBlk;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bf[nt][ks] = *reinterpret_cast<const bfrag8*>(u + (ks ? boff1 : boff0) + nt * 2048);
  };
  auto read_p0 = [&]() {
    read_b_ks(0, 0);
    read_a_ks(0);
    read_b_ks(0, 1);
    read_a_ks(1);
  };
  auto read_b = [&](int nh) {
    read_b_ks(nh, 0);
    read_b_ks(nh, 1);
  };
  // The 32 MFMAs of a phase in four groups of 8 (one k-step of one row half each); `between(g)` runs after
  // group g (LDS-DMA issue of the phase when kSpreadIssue: the matrix pipe stays busy while they go out).
  auto mfma_phase = [&](int nh, auto between) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mh][mt][2 * nh + nt] = mfma16(bf[nt][ks], af[mh][mt][ks], acc[mh][mt][2 * nh + nt]);
        between(2 * ks + mh);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto nothing = [](int) {};

  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(p.aux_out, 0, (int)p.aux_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.aux_in), 0, (int)p.aux_bytes, 0x00020000);
  constexpr int kStores = EPI == kEpiGelu ? 32 : 16;   // epilogue stores per wave and tile

  // Phase bodies. P0 of a tile's first K-tile has no barrier and issues nothing: its blocks were waited for before
  // the barrier that precedes the previous epilogue, and its two refills went out there.
  auto phase0 = [&](int kt, auto first_tag, auto wait_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int W = decltype(wait_tag)::value;
    if (!FIRST) barrier();
    if (!FIRST && !kSpreadIssue) {
      issue(T1{}, kt + 1);
      issue(T2{}, kt + 1);
    }
    read_p0();
    if (!FIRST && kSpreadIssue) {
      mfma_phase(0, [&](int g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g == 0) issue(T1{}, kt + 1);
        if (g == 1) issue(T2{}, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      mfma_phase(0, nothing);
    }
    wait_vm<W>();
  };
  auto phase1 = [&](int kt) {
    barrier();
    if (!kSpreadIssue) {
      issue(T3{}, kt + 1);
      issue(T4{}, kt + 1);
      issue(T5{}, kt + 1);
      issue(T0{}, kt + 2);
    } else {
      issue(T3{}, kt + 1);   // one phase ahead of its first read: goes out first
    }
    read_b(1);
    if (kSpreadIssue) {
      mfma_phase(1, [&](int g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g == 0) issue(T4{}, kt + 1);
        if (g == 1) issue(T5{}, kt + 1);
        if (g == 2) issue(T0{}, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      mfma_phase(1, nothing);
    }
    wait_vm<6>();
    rslot = ring(rslot, 6);
  };
  using First = std::true_type;
  using Later = std::false_type;

  // ---- prologue: the first K-tile and the first block of the second, then what every epilogue is preceded by
  issue(T0{}, 0);
  issue(T1{}, 0);
  issue(T2{}, 0);
  issue(T3{}, 0);
  issue(T4{}, 0);
  issue(T5{}, 0);
  issue(T0{}, 1);
  wait_vm<6>();
  barrier();
  issue(T1{}, 1);
  issue(T2{}, 1);
  {
    // as many out-of-range (dropped) stores as an epilogue issues: the first tile then counts its outstanding
    // operations exactly like every later one
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < kStores; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, rsC, 0x80000000u + 16u * i, 0, 0);
  }

  for (int it = 0; it < ntiles; ++it) {
    zero_acc();
    // first K-tile: its B1x blocks are older than the previous epilogue's stores and the two refills before them
    phase0(0, First{}, std::integral_constant<int, 6 + kStores>{});
    phase1(0);
    for (int kt = 1; kt < nk; ++kt) {
      phase0(kt, Later{}, std::integral_constant<int, 6>{});
      phase1(kt);
    }
    // all six blocks of the next tile's first K-tile and A0 of its second are issued; A0, A1, B00, B01 have landed
    barrier();
    issue(T1{}, nk + 1);
    issue(T2{}, nk + 1);

    // ---- epilogue (no barriers, no LDS-DMA: only loads and stores, counted in the next tile's first wait) ----
    {
      const int colq = cur_n0 + 64 * wc + 16 * q;   // first of this lane's 16 columns
      const bool col_ok = colq < p.N;
      const int col0 = cur_n0 + 64 * wc, lim = p.N - col0;   // the wave's 64 columns
      u32x4 bp0 = {0u, 0u, 0u, 0u}, bp1 = {0u, 0u, 0u, 0u};
      if (BIAS) {
        // N is a multiple of 16 and colq of 16: the lane's 32 bytes are inside the bias row or skipped
        const u32x4* gb = reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.bias) + (col_ok ? colq : 0));
        if (col_ok) {
          bp0 = gb[0];
          bp1 = gb[1];
        }
      }
      auto bias_at = [&](int c) -> float {
        if (!BIAS) return 0.f;
        const unsigned w = c < 8 ? bp0[c >> 1] : bp1[(c - 8) >> 1];
        return (c & 1) ? hi_bf16(w) : lo_bf16(w);
      };
      // The auxiliary operand (multiply / add forms) is read the way the output is written (8 lanes per 128-byte
      // line), every piece ahead of the first store, and turned into the accumulator layout through the
      // transposition buffer. Rows / columns past the end read as zeros.
      constexpr bool kAuxX = EPI == kEpiMul || EPI == kEpiAdd;
      u32x4 ax[2][4][2];
      if (kAuxX) {
        const bool cok = 8 * tch < lim;
        const unsigned cb = ((unsigned)col0 + 8u * tch) * 2u;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int r1 = cur_m0 + 64 * mh + 16 * mt + trow, r2 = r1 + 8;
            const unsigned o1 = (cok && r1 < p.M) ? (unsigned)r1 * (unsigned)p.ld_aux * 2u + cb : 0x80000000u;
            const unsigned o2 = (cok && r2 < p.M) ? (unsigned)r2 * (unsigned)p.ld_aux * 2u + cb : 0x80000000u;
            ax[mh][mt][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, o1, 0, 0);
            ax[mh][mt][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, o2, 0, 0);
          }
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int row0 = cur_m0 + 64 * mh + 16 * mt;
          float v[16];
#pragma unroll
          for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * T + e] = acc[mh][mt][T][e] + bias_at(4 * T + e);
          if (kAuxX) {
            *reinterpret_cast<u32x4*>(tbuf + tr0) = ax[mh][mt][0];
            *reinterpret_cast<u32x4*>(tbuf + tr1) = ax[mh][mt][1];
            asm volatile("" ::: "memory");
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(tbuf + tw0);
            const u32x4 x1 = *reinterpret_cast<const u32x4*>(tbuf + tw1);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a0 = lo_bf16(x0[j]), a1 = hi_bf16(x0[j]), a2 = lo_bf16(x1[j]), a3 = hi_bf16(x1[j]);
              if (EPI == kEpiMul) {
                v[2 * j] *= a0; v[2 * j + 1] *= a1; v[8 + 2 * j] *= a2; v[8 + 2 * j + 1] *= a3;
              } else {
                v[2 * j] += a0; v[2 * j + 1] += a1; v[8 + 2 * j] += a2; v[8 + 2 * j + 1] += a3;
              }
            }
          }
          u32x4 o0, o1;
          if (EPI == kEpiGelu) {
            u32x4 g0, g1;
            float a[16], d[16];
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
              f32x2 a2, d2;
              gelu_and_grad2(f32x2{v[j], v[j + 1]}, a2, d2);
              a[j] = a2[0]; a[j + 1] = a2[1];
              d[j] = d2[0]; d[j + 1] = d2[1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              o0[j] = pack_bf16x2(a[2 * j], a[2 * j + 1]);
              o1[j] = pack_bf16x2(a[8 + 2 * j], a[8 + 2 * j + 1]);
              g0[j] = pack_bf16x2(d[2 * j], d[2 * j + 1]);
              g1[j] = pack_bf16x2(d[8 + 2 * j], d[8 + 2 * j + 1]);
            }
            store_rows16(rsG, g0, g1, row0, col0, lim, (unsigned)p.ld_aux);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              o0[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
              o1[j] = pack_bf16x2(v[8 + 2 * j], v[8 + 2 * j + 1]);
            }
          }
          store_rows16(rsC, o0, o1, row0, col0, lim, (unsigned)p.ldc);
        }
      }
    }
    cur_m0 = nxt_m0;
    cur_n0 = nxt_n0;
    have_nxt = it + 2 < ntiles;
    if (have_nxt) {
      const TileCoord t2 = walk.at(wg, it + 2);
      nxt_m0 = t2.m0;
      nxt_n0 = t2.n0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (zero-fill) LDS-DMA before the LDS is released
}

bool gemm_nt2_ok(int epi, bool grouped) {
  return !grouped && (epi == kEpiNone || epi == kEpiGelu || epi == kEpiMul || epi == kEpiAdd);
}

int g_gemm_nt2_skew = -1;   // benchmark knob: start skew of the second half of the grid in 10 ns ticks (-1: from the shape)
void gemm_nt2_set_skew(int ticks) { g_gemm_nt2_skew = ticks; }

// One launch over a row range whose byte offsets stay below 2 GiB (callers: gemm_nt2()).
static void gemm_nt2_launch(GemmNtArgs a, int epi, int num_cus, hipStream_t st) {
  a.tiles_m = (a.M + 127) / 128;
  a.tiles_n = (a.N + 255) / 256;
  a.a_bytes = (unsigned)(((long long)a.M - 1) * a.lda * 2 + (long long)a.K * 2);
  a.b_bytes = (unsigned)(((long long)a.N - 1) * a.ldb * 2 + (long long)a.K * 2);
  a.c_bytes = (unsigned)(((long long)a.M - 1) * a.ldc * 2 + (long long)a.N * 2);
  a.aux_bytes = (unsigned)(((long long)a.M - 1) * a.ld_aux * 2 + (long long)a.N * 2);
  // Half a tile's time: a tile is 2 * 128 * 256 * K FLOP at about 2.3 TFLOP/s per workgroup (two per CU share
  // 4.7 TFLOP/s), plus the epilogue; in 10 ns ticks of s_memrealtime.
  if (g_gemm_nt2_skew >= 0) a.skew_ticks = g_gemm_nt2_skew;
  else a.skew_ticks = (int)(0.5 * (2.0 * 128 * 256 * a.K / 2.3e12) * 1e8) + 100;
  const int grid = gemm_nt_grid(a.tiles_m, a.tiles_n, num_cus, 2);
  static bool attr = false;
#define DPA_GNT2_ALL(F) \
  F(kEpiNone, false) F(kEpiNone, true) F(kEpiGelu, false) F(kEpiGelu, true) F(kEpiMul, false) F(kEpiMul, true) \
  F(kEpiAdd, false) F(kEpiAdd, true)
  if (!attr) {
#define DPA_GNT2_ATTR(E, B) \
  (void)hipFuncSetAttribute((const void*)gemm_nt2_kernel<E, B>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
    DPA_GNT2_ALL(DPA_GNT2_ATTR)
#undef DPA_GNT2_ATTR
    attr = true;
  }
  const bool bias = a.bias != nullptr;
#define DPA_GNT2_LAUNCH(E, B)                                                                \
  if (epi == E && bias == B) {                                                               \
    hipLaunchKernelGGL((gemm_nt2_kernel<E, B>), dim3(grid), dim3(256), kLds2, st, a);        \
    return;                                                                                  \
  }
  DPA_GNT2_ALL(DPA_GNT2_LAUNCH)
#undef DPA_GNT2_LAUNCH
#undef DPA_GNT2_ALL
}

void gemm_nt2(GemmNtArgs a, int epi, int num_cus, hipStream_t st) {
  if (a.M == 0 || a.N == 0) return;
  const long long chunk = gemm_nt_rows_per_launch(a, epi);
  if (chunk <= 0 || !gemm_nt2_ok(epi, a.w_ptrs != nullptr)) {
    fprintf(stderr, "gemm_nt2: unsupported form or operands past 2 GiB per row range (callers check)\n");
    abort();
  }
  const int M = a.M;
  for (long long m0 = 0; m0 < M; m0 += chunk) {
    GemmNtArgs s = a;
    s.M = (int)(M - m0 < chunk ? M - m0 : chunk);
    s.a = static_cast<const char*>(a.a) + m0 * a.lda * 2;
    s.c = static_cast<char*>(a.c) + m0 * a.ldc * 2;
    if (a.aux_in != nullptr) s.aux_in = static_cast<const char*>(a.aux_in) + m0 * a.ld_aux * 2;
    if (a.aux_out != nullptr) s.aux_out = static_cast<char*>(a.aux_out) + m0 * a.ld_aux * 2;
    gemm_nt2_launch(s, epi, num_cus, st);
  }
}

}  // namespace dpa
