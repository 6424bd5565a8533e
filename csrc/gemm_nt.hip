// NT GEMM with fused epilogues for gfx950:   C[M, N] = epi( A[M, K] * B[N, K]^T )   (bf16 in, fp32
// accumulate, bf16 out). Both operands are contraction-contiguous, which is the layout of every
// forward product of a Linear layer (x * W^T) and, with the weight transposed once per optimizer
// step, of every input-gradient product (dy * (W^T)^T) as well.
//
// Structure (one persistent 512-thread workgroup per CU, 256 x 256 output tile, 64-deep K-tiles):
//  * 8 waves as 2 (M) x 4 (N), each owning 128 x 64 of the tile in 32 accumulators of
//    v_mfma_f32_16x16x32_bf16. The product is formed transposed (the B rows are the MFMA's row
//    operand), so a lane ends up with 16 CONSECUTIVE output columns of one output row and stores
//    them as two 16-byte pieces; a quad of lanes writes one whole 128-byte line. No LDS transpose.
//  * The two wave groups (rows 0-127 / 128-255) run the same program one barrier apart: while one
//    group's 16 MFMAs of a phase occupy the matrix pipe of a SIMD, its partner on that SIMD issues
//    the LDS reads and LDS-DMA of its own next phase ("ping-pong").
//  * Operand tiles travel global -> LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane, issued
//    from inline assembly and retired with counted s_waitcnt vmcnt, never vmcnt(0)); out-of-range rows
//    read as zeros through the buffer descriptor's range check. A K-tile is four 16 KiB units
//    (A rows 0-63 / 64-127 of both wave groups, B columns 0-31 / 32-63 of all four wave columns),
//    two copies of each: one unit is refilled per phase, four phases (about a microsecond) ahead of
//    its first read. The refill stream runs on across output tiles, so the first K-tiles of the next
//    output tile land while this tile's epilogue runs.
//  * LDS rows are 128 bytes (64 k); 16-byte chunk c of row r sits at chunk slot c ^ ((r >> 1) & 7),
//    applied on the global source address of the DMA and again on the ds_read_b128 address:
//    conflict-free for the 16-row x 4-chunk fragment reads.
//  * Epilogues on the accumulators: + bias (staged once in LDS as fp32); exact-erf GELU that also
//    emits GELU'(x) for the backward pass; x auxiliary tensor (the dGELU of the backward pass);
//    + auxiliary tensor (residual); rotary embedding of the leading (q, k) columns of a fused qkv projection.
//
// Replaces the GEMMs that nn.Linear issues implicitly in the reference
// (reference: single-gpu/model.py:112-114,153,373-374,388-398,559).
#include <type_traits>

#include "common.h"
#include "host_logic.h"
#include "kernels.h"

namespace dpa {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bfrag8;
typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;

#ifndef DPA_GEMM_STORE_AUX
#define DPA_GEMM_STORE_AUX 2   // non-temporal: measured 4-6 % faster than the default policy on the K = 768 shapes
#endif
#ifndef DPA_GEMM_AUX_LOAD
#define DPA_GEMM_AUX_LOAD 2   // cache policy of the epilogue loads of the auxiliary operand (read once): non-temporal, -2 % on the dGELU form, neutral on the residual form
#endif
#ifndef DPA_GEMM_PREFETCH_A
#define DPA_GEMM_PREFETCH_A 1   // phase 3 (no fragment reads of its own) reads half of the next K-tile's first A fragments
#endif
#ifndef DPA_GEMM_LATE_BARRIER
#define DPA_GEMM_LATE_BARRIER 1   // the second wave group's last barrier of a tile comes after its epilogue
#endif
#ifndef DPA_GEMM_PHASES
#define DPA_GEMM_PHASES 2   // phases (memory section + compute section) per K-tile: 2 x 32 MFMAs or 4 x 16
#endif
constexpr bool kTwoPhase = DPA_GEMM_PHASES == 2;
constexpr bool kPrefetchA = DPA_GEMM_PREFETCH_A != 0 && !kTwoPhase;   // (the 2-phase order leaves no landed unit to read early)
constexpr int kStoreAux = DPA_GEMM_STORE_AUX;   // cache policy bits of the epilogue stores (1 sc0, 2 nt, 16 sc1)
constexpr int kUnit = 16384;            // one LDS unit: 128 rows x 64 k x 2 bytes
constexpr int kCopy = 4 * kUnit;        // a K-tile: Am0, Am1, Bn0, Bn1
#ifndef DPA_GEMM_XPOSE
#define DPA_GEMM_XPOSE 1   // epilogue rows regrouped through LDS: 8 consecutive lanes store (load) one 128-byte line
#endif
constexpr bool kXpose = DPA_GEMM_XPOSE != 0;
constexpr int kXposeLds = 2 * kCopy;    // 2 KiB per wave: one 16-row x 64-column bf16 piece of the output tile
constexpr int kBiasLds = kXposeLds + 8 * 2048;   // bf16 bias staged behind them
constexpr int kMaxBiasN = kGemmNtMaxBiasN;
constexpr int kLds = kBiasLds + kMaxBiasN * 2;   // 160 KiB

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
// (M0 is written and not restored: nothing else in this kernel reads it)
#ifndef DPA_GEMM_M0_RESTORE
#define DPA_GEMM_M0_RESTORE 0
#endif
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, rsrc_t rs) {
#if DPA_GEMM_M0_RESTORE
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
#else
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds"
      :
      : "s"(lds_addr), "v"(voff), "s"(rs)
      : "memory", "m0");
#endif
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

// Exact-erf GELU and its derivative. erfc by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7):
// Phi(-|x|) = 0.5 * t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2).
#ifndef DPA_GELU_VARIANT
#define DPA_GELU_VARIANT 2   // 0: one value at a time; 1: pairs in packed fp32 arithmetic; 2: pairs, sign by bit copy; 3: 2 + one reciprocal per pair
#endif
__device__ __forceinline__ void gelu_and_grad(float x, float& act, float& dact) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678f, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950409f));
  float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  poly = fmaf(poly, t, 0.5f * 1.421413741f);
  poly = fmaf(poly, t, 0.5f * -0.284496736f);
  poly = fmaf(poly, t, 0.5f * 0.254829592f);
  const float h = poly * t * e;
  const float phi = x >= 0.f ? 1.0f - h : h;
  act = x * phi;
  dact = fmaf(x * e, 0.3989422804f, phi);
}

// The same on two values with the plain arithmetic in packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ void gelu_and_grad2(f32x2 x, f32x2& act, f32x2& dact) {
#if DPA_GELU_VARIANT >= 10
  // timing-only ablations: 10 = the transcendental instructions replaced by copies, 11 = no arithmetic at all
  if (DPA_GELU_VARIANT == 11) { act = x; dact = x; return; }
#endif
#if DPA_GELU_VARIANT == 1
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
#else
  // Phi(x) = 0.5 + sign(x) (0.5 - Phi(-|x|)): the sign goes in by a bit copy instead of a compare and a select per
  // value, |x| rides in the fused multiply-add's operand modifier, the exponent's factor is split over both
  // factors of the square.
  const float k1 = 0.3275911f * 0.70710678f;
  const f32x2 u = {fmaf(fabsf(x[0]), k1, 1.0f), fmaf(fabsf(x[1]), k1, 1.0f)};
#if DPA_GELU_VARIANT == 3
  const float r = __builtin_amdgcn_rcpf(u[0] * u[1]);   // one reciprocal for both: 1 / u0 = u1 / (u0 u1)
  const f32x2 t = f32x2{u[1], u[0]} * splat2(r);
#elif DPA_GELU_VARIANT == 10
  const f32x2 t = u;
#else
  const f32x2 t = {__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
#endif
  // e = exp(-x^2 / 2) / sqrt(2 pi), the density itself (the constant rides in the exponent; the polynomial's
  // coefficients carry its inverse)
  const f32x2 s = x * splat2(0.84932180028f);   // sqrt(0.5 log2(e))
  const f32x2 xx = pk_fma(s, s, splat2(1.32574806473f));   // + log2(sqrt(2 pi))
#if DPA_GELU_VARIANT == 10
  const f32x2 e = xx;
#else
  const f32x2 e = {__builtin_amdgcn_exp2f(-xx[0]), __builtin_amdgcn_exp2f(-xx[1])};
#endif
  constexpr float kc = 0.5f * 2.50662827463f;   // 0.5 sqrt(2 pi)
  f32x2 poly = pk_fma(t, splat2(kc * 1.061405429f), splat2(kc * -1.453152027f));
  poly = pk_fma(poly, t, splat2(kc * 1.421413741f));
  poly = pk_fma(poly, t, splat2(kc * -0.284496736f));
  poly = pk_fma(poly, t, splat2(kc * 0.254829592f));
  const f32x2 tp = poly * t;
  const f32x2 q = pk_fma(-e, tp, splat2(0.5f));   // 0.5 - Phi(-|x|) >= 0
  const f32x2 phi = f32x2{__builtin_copysignf(q[0], x[0]), __builtin_copysignf(q[1], x[1])} + splat2(0.5f);
  act = x * phi;
  dact = pk_fma(x, e, phi);
#endif
}

// vmcnt immediates of the four phases of a K-tile, and whether phases 0 and 1 issue a unit refill
template <int W0, int W1, int W2, int W3, bool ISSUE01>
struct KtCfg {
  static constexpr int kW0 = W0, kW1 = W1, kW2 = W2, kW3 = W3;
  static constexpr bool kIssue01 = ISSUE01;
};

}  // namespace

// Timing trace (benchmark builds with -DDPA_GEMM_TRACE only): every wave of one workgroup stamps the clock
// after each phase of two consecutive output tiles into the (unused, bias-free) tail of the LDS; the
// stamps are copied out when the workgroup ends. LDS writes only, so no vector-memory count moves.
#ifdef DPA_GEMM_TRACE
__device__ unsigned long long g_gemm_trace[8 * 128];
constexpr int kTraceWg = 37, kTraceTile = 5;
void gemm_nt_read_trace(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 8 * 128);
}
#define DPA_TR(slot)                                                                                              \
  do {                                                                                                            \
    if (trace_base >= 0) {                                                                                        \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                 \
      if (lane == 0) reinterpret_cast<unsigned long long*>(smem + kBiasLds)[wave * 128 + trace_base + (slot)] = t_; \
    }                                                                                                             \
  } while (0)
#else
#define DPA_TR(slot) do { } while (0)
#endif

// ABL (timing-only ablations, results wrong): 1 = no LDS-DMA in the K-loop, 2 = no fragment reads, 4 = no
// barriers, 8 = no MFMAs.
template <int EPI, bool BIAS, int SYNC, int ABL = 0, bool GROUPED = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel(GemmNtArgs p) {
  constexpr bool kPair = EPI == kEpiSwiglu;   // output tile = 128 gate columns paired with their 128 up columns
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int q = lane >> 4, l15 = lane & 15;
  const int nk = p.K / 64;

  const rsrc_t rsB = make_rsrc(p.b, p.b_bytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

  if (BIAS) {
    bf16_t* sb = reinterpret_cast<bf16_t*>(smem + kBiasLds);
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(p.bias);
    for (int i = tid; i < p.N; i += 512) sb[i] = bias[i];
    __syncthreads();
  }

  // ---- LDS-DMA source offsets of this lane (bytes inside the 256-row A / B panel of a tile) -------
  // Piece pc = 2 * wave + pp of a unit covers unit rows 8 pc .. 8 pc + 7; lane l -> row 8 pc + (l >> 3),
  // chunk slot l & 7, which must receive logical chunk (l & 7) ^ ((row >> 1) & 7).
  unsigned voffA0, voffA1, voffB0, voffB1, pieceLds0, pieceLds1;
  {
    auto piece = [&](int pp, unsigned& va, unsigned& vb, unsigned& pl) {
      const int pc = 2 * wave + pp;
      const int rho = 8 * pc + (lane >> 3);
      const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
      // A unit Am{mh}: unit row rho <-> tile row 128 (rho >> 6) + 64 mh + (rho & 63)
      const int rowA = 128 * (rho >> 6) + (rho & 63);
      va = (unsigned)rowA * (unsigned)p.lda * 2u + chunk * 16;
      // B unit Bn{nh}: unit row rho = 32 wc + 16 nt + i <-> tile column 64 wc + 16 (i >> 2) + 4 (2 nh + nt) + (i & 3)
      const int i = rho & 15, nt = (rho >> 4) & 1, wcs = rho >> 5;
      // paired (SwiGLU) tiles: a lane's columns 0-7 are gate columns, 8-15 the matching up columns (unit
      // Bn1 = the same rows of the up half of the weight), so 32 output columns per wave column
      const int colB = kPair ? 32 * wcs + 8 * (i >> 2) + 4 * nt + (i & 3) : 64 * wcs + 16 * (i >> 2) + 4 * nt + (i & 3);
      vb = (unsigned)colB * (unsigned)p.ldb * 2u + chunk * 16;
      pl = lds0 + pc * 1024;
    };
    piece(0, voffA0, voffB0, pieceLds0);
    piece(1, voffA1, voffB1, pieceLds1);
    // opaque from here on: keeps every issue at one scalar base + one v_add per piece
    asm volatile("" : "+v"(voffA0), "+v"(voffA1), "+v"(voffB0), "+v"(voffB1));
    pieceLds0 = __builtin_amdgcn_readfirstlane(pieceLds0);
    pieceLds1 = __builtin_amdgcn_readfirstlane(pieceLds1);
  }
  const unsigned aStepMh = 64u * (unsigned)p.lda * 2u;   // Am1 rows = Am0 rows + 64
  // Bn1 columns = Bn0 columns + 8; paired tiles: Bn1 = the up rows, N weight rows after the gate rows
  const unsigned bStepNh = (kPair ? (unsigned)p.N : 8u) * (unsigned)p.ldb * 2u;

  // ---- fragment read offsets (bytes inside a unit) -------------------------------------------------
  unsigned aoff0, aoff1, boff0, boff1;
  {
    const int rhoA = 64 * wr + l15, rhoB = 32 * wc + l15;
    aoff0 = rhoA * 128 + (((0 + q) ^ ((rhoA >> 1) & 7)) << 4);
    aoff1 = rhoA * 128 + (((4 + q) ^ ((rhoA >> 1) & 7)) << 4);
    boff0 = rhoB * 128 + (((0 + q) ^ ((rhoB >> 1) & 7)) << 4);
    boff1 = rhoB * 128 + (((4 + q) ^ ((rhoB >> 1) & 7)) << 4);
  }

  if (p.skew_ticks > 0) {   // benchmark aid: start the workgroups in four groups a quarter of a tile apart
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 3) * p.skew_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  GemmWalk walk{p.tiles_m, p.tiles_n, (int)gridDim.x};
  const int wg = blockIdx.x;
  const int ntiles = walk.count(wg);
  if (ntiles == 0) return;

  // ---- refill stream ---------------------------------------------------------------------------
  // Units are issued in the fixed order Am0, Bn0 | Bn1, Am1 of consecutive K-tiles, running on into
  // the next output tile; past the last tile the offsets point outside the buffers (zeros, never read).
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
  // Grouped mode: the weight matrix of a tile is the one of its 256-row block's group; a block without a
  // group (padding past the last segment) gets an empty buffer range, i.e. multiplies by zeros. The
  // descriptor of the tile after next is formed a whole tile ahead of its first use.
  auto group_rsrc = [&](int m0) -> rsrc_t {
    const int g = p.block_group[m0 >> 8];
    const bool live = g >= 0 && g < p.ngroups;
    return make_rsrc(reinterpret_cast<const void*>(p.w_ptrs[live ? g : 0]), live ? p.b_bytes : 0u);
  };
  // The A operand is addressed per output tile: the descriptor starts at the tile's first row and covers its 256
  // rows (fewer at the bottom edge: rows past M read zeros), so offsets stay below 2^30 bytes however tall A is
  // (dlogits [65536, 50304] of the head's input gradient is 6.6 GB). One 64-bit multiply-add per tile.
  auto a_rsrc = [&](int m0) -> rsrc_t {
    int rows = p.M - m0;
    if (rows > 256) rows = 256;
    if (rows < 1) rows = 1;
    return make_rsrc(static_cast<const char*>(p.a) + (long long)m0 * p.lda * 2,
                     (unsigned)((rows - 1) * (long long)p.lda * 2 + (long long)p.K * 2));
  };
  rsrc_t rsA_cur = a_rsrc(cur_m0), rsA_nxt = have_nxt ? a_rsrc(nxt_m0) : rsA_cur;
  rsrc_t rsB_cur = rsB, rsB_nxt = rsB;
  if (GROUPED) {
    rsB_cur = group_rsrc(cur_m0);
    rsB_nxt = have_nxt ? group_rsrc(nxt_m0) : rsB_cur;
  }
  // a dead refill (past the last tile) is aimed at offsets >= 2 GiB, beyond every buffer: it reads zeros
  const unsigned kOob = 0x80000000u;

  // IS_A / HALF are compile-time tags (std::integral_constant): unit Am{HALF} or Bn{HALF} of K-tile kk
  // (kk may run into the next output tile) goes to LDS copy `copy`.
  auto issue = [&](auto is_a_tag, auto half_tag, int kk, int copy) {
    constexpr bool IS_A = decltype(is_a_tag)::value;
    constexpr int HALF = decltype(half_tag)::value;
    const bool over = kk >= nk;
    const unsigned k = (unsigned)(over ? kk - nk : kk) * 128u;
    unsigned base;
    if (IS_A) base = (HALF ? aStepMh : 0u) + k;
    else base = ((unsigned)(over ? nxt_n0 : cur_n0) >> (kPair ? 1 : 0)) * (unsigned)p.ldb * 2u + (HALF ? bStepNh : 0u) + k;
    if (over && !have_nxt) base = kOob;
    const unsigned ldsu = copy * kCopy + ((IS_A ? 0 : 2) + HALF) * kUnit;
    const unsigned v0 = base + (IS_A ? voffA0 : voffB0);
    const unsigned v1 = base + (IS_A ? voffA1 : voffB1);
    if (ABL & 1) {
      asm volatile("" ::"v"(v0), "v"(v1), "s"(ldsu));
      return;
    }
    const rsrc_t rs = IS_A ? (over ? rsA_nxt : rsA_cur) : (GROUPED ? (over ? rsB_nxt : rsB_cur) : rsB);
    dma16(pieceLds0 + ldsu, v0, rs);
    dma16(pieceLds1 + ldsu, v1, rs);
  };
  using TA = std::true_type;
  using TB = std::false_type;
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  // A lane holds columns colq .. colq + 15 of row row0 + l15 as two 16-byte halves, so neighbouring lanes hold
  // different rows and a store instruction issued from that layout touches 64 separate lines (measured: about
  // 72 clocks per 1 KiB store instruction, which made the epilogue cost 9 K clocks per tile). The 16-row x
  // 64-column piece therefore goes through a wave-private 2 KiB LDS buffer (16-byte slots XOR-swizzled by
  // the row, conflict-free both ways) and comes back with lane L holding chunk L & 7 of rows L >> 3 and
  // (L >> 3) + 8: every 8 consecutive lanes write one whole 128-byte line. LDS operations of one wave execute
  // in order, so the buffer needs no barrier and is reused immediately.
  char* const tbuf = smem + kXposeLds + wave * 2048;
  const unsigned tw0 = l15 * 128 + (((2 * q) ^ (l15 & 7)) << 4), tw1 = l15 * 128 + (((2 * q + 1) ^ (l15 & 7)) << 4);
  const int trow = lane >> 3, tch = lane & 7;
  const unsigned tr0 = trow * 128 + ((tch ^ (trow & 7)) << 4), tr1 = tr0 + 1024;
  // col0 = first column of the piece in the tensor (a multiple of 64), lim = number of valid columns from col0 on
  auto store_rows16 = [&](__amdgpu_buffer_rsrc_t rs, u32x4 lo, u32x4 hi, int row0, int col0, int lim, unsigned ld) {
    if (kXpose) {
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
      __builtin_amdgcn_raw_buffer_store_b128(st1, rs, o1, 0, kStoreAux);
      __builtin_amdgcn_raw_buffer_store_b128(st2, rs, o2, 0, kStoreAux);
      return;
    }
    // register-only form: lanes l15 and l15 ^ 8 trade one half each (DPP row rotate by 8), after which a store
    // instruction writes 8 whole lines, though still one row per lane
    const bool low = l15 < 8;
    const bool col_ok = 16 * q < lim;
    u32x4 recv;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      recv[j] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(low ? hi[j] : lo[j]), 0x128, 0xf, 0xf, false);
    const u32x4 st1 = low ? lo : recv, st2 = low ? recv : hi;
    const int r1 = row0 + (l15 & 7), r2 = r1 + 8;
    const unsigned cb = ((unsigned)col0 + 16u * q + (low ? 0u : 8u)) * 2u;
    const unsigned o1 = (col_ok && r1 < p.M) ? (unsigned)r1 * ld * 2u + cb : 0x80000000u;
    const unsigned o2 = (col_ok && r2 < p.M) ? (unsigned)r2 * ld * 2u + cb : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(st1, rs, o1, 0, kStoreAux);
    __builtin_amdgcn_raw_buffer_store_b128(st2, rs, o2, 0, kStoreAux);
  };

  f32x4 acc[2][4][4];
  bfrag8 af[4][2] = {}, bf0[2][2] = {}, bf1[2][2] = {};
  bfrag8 afn[2][2] = {};   // rows 0-31 of the NEXT K-tile's first A unit, read one phase early (kPrefetchA)

  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto read_a = [&](int copy, int mh) {
    if (ABL & 2) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(af[mt][ks]));
      return;
    }
    const char* u = smem + copy * kCopy + mh * kUnit;
#pragma unroll
    for (int mt = (kPrefetchA && mh == 0) ? 2 : 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[mt][ks] = *reinterpret_cast<const bfrag8*>(u + (ks ? aoff1 : aoff0) + mt * 2048);
  };
  // The memory sections of a K-tile read 12, 4, 8 and 0 fragments; with this, 8, 4, 8 and 4: the first
  // A unit of K-tile kt + 1 has landed (and passed a barrier) by phase 3 of K-tile kt under the same waits.
  auto prefetch_a = [&](int copy_next) {
    if (!kPrefetchA) return;
    if (ABL & 2) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(afn[mt][ks]));
      return;
    }
    const char* u = smem + copy_next * kCopy;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        afn[mt][ks] = *reinterpret_cast<const bfrag8*>(u + (ks ? aoff1 : aoff0) + mt * 2048);
  };
  auto read_b = [&](int copy, int nh, bfrag8 (&bf)[2][2]) {
    if (ABL & 2) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(bf[nt][ks]));
      return;
    }
    const char* u = smem + copy * kCopy + (2 + nh) * kUnit;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        bf[nt][ks] = *reinterpret_cast<const bfrag8*>(u + (ks ? boff1 : boff0) + nt * 2048);
  };
  auto quadrant = [&](int mh, int nh, const bfrag8 (&bf)[2][2]) {
    if (ABL & 8) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("" ::"v"(af[mt][ks]));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) asm volatile("" ::"v"(bf[nt][ks]));
      }
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mh][mt][2 * nh + nt] = mfma16(bf[nt][ks], (kPrefetchA && mh == 0 && mt < 2) ? afn[mt][ks] : af[mt][ks],
                                            acc[mh][mt][2 * nh + nt]);
    __builtin_amdgcn_s_setprio(0);
  };

  // Phase = memory section (fragment reads, one unit refill, counted wait) + compute section (16 MFMAs).
  // SYNC 0: a barrier after each section, the second wave group one barrier behind the first, so the
  // groups strictly alternate between the sections. SYNC 1: one barrier per phase, after the compute
  // section for the first group and between the sections for the second, so that inside every barrier
  // interval one group runs memory -> compute and the other compute -> memory. Either way a unit is
  // waited for one phase before its first read and refilled two phases after its last.
  auto sync_mc = [&]() {
    if (ABL & 4) return;
    if (SYNC == 0 || wr == 1) barrier();
  };
  auto sync_cm = [&]() {
    if (ABL & 4) return;
    if (SYNC == 0 || wr == 0) barrier();
  };
  // One K-tile = four phases; W = vector-memory operations allowed to stay in flight at each wait
  // (8 = the four most recent units; + the epilogue's stores while they are younger than the unit
  // being waited for).
  int trace_base = -1;
  (void)trace_base;
  constexpr bool kLateBarrier = SYNC == 0 && DPA_GEMM_LATE_BARRIER != 0;
  // Two phases per K-tile: A = rows 0-63 x 64 columns (16 fragment reads, 32 MFMAs), B = rows 64-127 (8 reads,
  // 32 MFMAs). Each section boundary costs a latency tail in which neither the LDS pipe nor the matrix pipe
  // works, so half as many sections per K-tile are faster although the LDS traffic is the same. Refill order
  // of the stream: Am0, Bn0, Bn1 of K-tile kt + 2 in phase B of kt (their slots were read in phase A), Am1 of
  // kt + 1 in phase A of kt (read in phase B of kt - 1); every unit is waited for one phase before its first
  // read, a K-tile after its issue, with the 8 youngest pieces (4 units) left in flight.
  auto ktile2 = [&](int kt, int copy, auto cfg, bool last = false) {
    using Cfg = decltype(cfg);
    read_a(copy, 0);
    read_b(copy, 0, bf0);
    read_b(copy, 1, bf1);
    if (Cfg::kIssue01) issue(TA{}, H1{}, kt + 1, copy ^ 1);
    wait_vm<Cfg::kW0>();
    sync_mc();
    DPA_TR(kt * 4 + 0);
    quadrant(0, 0, bf0);
    quadrant(0, 1, bf1);
    sync_cm();
    DPA_TR(kt * 4 + 1);
    read_a(copy, 1);
    issue(TA{}, H0{}, kt + 2, copy);
    issue(TB{}, H0{}, kt + 2, copy);
    issue(TB{}, H1{}, kt + 2, copy);
    wait_vm<Cfg::kW2>();
    sync_mc();
    DPA_TR(kt * 4 + 2);
    quadrant(1, 1, bf1);
    quadrant(1, 0, bf0);
    if (!(kLateBarrier && last && wr == 1)) sync_cm();   // see ktile4
    DPA_TR(kt * 4 + 3);
  };
  auto ktile4 = [&](int kt, int copy, auto cfg, bool last = false) {
    using Cfg = decltype(cfg);
    // phase 0: quadrant (rows 0-63, columns 0-31)
    read_a(copy, 0);
    read_b(copy, 0, bf0);
    if (Cfg::kIssue01) issue(TB{}, H1{}, kt + 1, copy ^ 1);
    wait_vm<Cfg::kW0>();
    sync_mc();
    quadrant(0, 0, bf0);
    sync_cm();
    DPA_TR(kt * 4 + 0);
    // phase 1: (rows 0-63, columns 32-63)
    read_b(copy, 1, bf1);
    if (Cfg::kIssue01) issue(TA{}, H1{}, kt + 1, copy ^ 1);
    wait_vm<Cfg::kW1>();
    sync_mc();
    quadrant(0, 1, bf1);
    sync_cm();
    DPA_TR(kt * 4 + 1);
    // phase 2: (rows 64-127, columns 32-63)
    read_a(copy, 1);
    issue(TA{}, H0{}, kt + 2, copy);
    wait_vm<Cfg::kW2>();
    sync_mc();
    quadrant(1, 1, bf1);
    sync_cm();
    DPA_TR(kt * 4 + 2);
    // phase 3: (rows 64-127, columns 0-31), operands already in registers
    prefetch_a(copy ^ 1);
    issue(TB{}, H0{}, kt + 2, copy);
    wait_vm<Cfg::kW3>();
    sync_mc();
    quadrant(1, 0, bf0);
    // After the last K-tile the second wave group runs its epilogue BEFORE this barrier: the first group
    // (one barrier ahead) is in its own epilogue by then, so the two epilogues overlap instead of each
    // group waiting out the other's at the next barrier. The epilogue touches no operand LDS.
    if (!(kLateBarrier && last && wr == 1)) sync_cm();
    DPA_TR(kt * 4 + 3);
  };

  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(p.aux_out, 0, (int)p.aux_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.aux_in), 0, (int)p.aux_bytes, 0x00020000);
  // epilogue stores per wave and tile
  constexpr int kStores = (EPI == kEpiGelu || EPI == kEpiDswiglu) ? 32 : (EPI == kEpiSwiglu ? 24 : 16);

  // ---- prologue: both K-tile copies of the first tile (the order of the stream) ----------------------
  issue(TA{}, H0{}, 0, 0);
  issue(TB{}, H0{}, 0, 0);
  issue(TB{}, H1{}, 0, 0);
  issue(TA{}, H1{}, 0, 0);
  issue(TA{}, H0{}, 1, 1);
  issue(TB{}, H0{}, 1, 1);
  issue(TB{}, H1{}, 1, 1);
  issue(TA{}, H1{}, 1, 1);
  // as many out-of-range (dropped) stores as an epilogue issues: the first tile then counts its
  // outstanding operations exactly like every later one
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < kStores; ++i)   // distinct offsets: identical stores would be merged into one
      __builtin_amdgcn_raw_buffer_store_b128(z, rsC, 0x80000000u + 16u * i, 0, 0);
  }
  wait_vm<(kTwoPhase ? 10 : 12) + kStores>();   // the units the first memory section reads
  barrier();
  if (SYNC == 0 && wr == 1) barrier();   // the second wave group runs one barrier behind the first
  prefetch_a(0);


  int copy = 0;
  for (int it = 0; it < ntiles; ++it) {
#ifdef DPA_GEMM_TRACE
    trace_base = (!BIAS && blockIdx.x == kTraceWg && nk <= 12 && (it == kTraceTile || it == kTraceTile + 1)) ? 60 * (it - kTraceTile) : -1;
    DPA_TR(58);
#endif
    zero_acc();
    // Waits: a unit is waited for one phase before its first read, leaving in flight the vector-memory
    // operations issued after it: 8 LDS-DMA pieces in the steady state; in the first K-tile of a tile
    // two more (its phases 0 and 1 issue nothing: both copies were filled ahead, the last two units
    // at the start of the previous epilogue), plus that epilogue's stores while they are younger than
    // the unit in question (through phase 1 of the second K-tile).
    if (kTwoPhase) {
      // 2-phase waits: phase A of the first K-tile issues nothing (its unit went out ahead of the previous
      // epilogue); the previous epilogue's stores are younger than the awaited unit through phase A of the
      // second K-tile.
      ktile2(0, copy, KtCfg<8 + kStores, 0, 8 + kStores, 0, false>{});
      copy ^= 1;
      ktile2(1, copy, KtCfg<8 + kStores, 0, 8, 0, true>{}, nk == 2);
      copy ^= 1;
      for (int kt = 2; kt < nk; ++kt) {
        ktile2(kt, copy, KtCfg<8, 0, 8, 0, true>{}, kt == nk - 1);
        copy ^= 1;
      }
      issue(TA{}, H1{}, nk + 1, copy ^ 1);   // ahead of the stores: the slot's last reader was the last phase B
    } else {
      ktile4(0, copy, KtCfg<10 + kStores, 8 + kStores, 8 + kStores, 8 + kStores, false>{});
      copy ^= 1;
      ktile4(1, copy, KtCfg<8 + kStores, 8 + kStores, 8, 8, true>{}, nk == 2);
      copy ^= 1;
      for (int kt = 2; kt < nk; ++kt) {
        ktile4(kt, copy, KtCfg<8, 8, 8, 8, true>{}, kt == nk - 1);
        copy ^= 1;
      }
      // the two units of the next tile's second K-tile whose slots the last K-tile has finished reading:
      // issued ahead of the stores, so that nothing the next tile needs soon queues behind them
      issue(TB{}, H1{}, nk + 1, copy ^ 1);
      issue(TA{}, H1{}, nk + 1, copy ^ 1);
    }
    DPA_TR(48);

    // ---- epilogue (no barriers, no LDS-DMA: only stores, counted in the next tile's first waits) ----
    {
      // first of this lane's 16 columns (paired tiles: of its 8 output columns)
      const int colq = kPair ? (cur_n0 >> 1) + 32 * wc + 8 * q : cur_n0 + 64 * wc + 16 * q;
      const bool col_ok = colq < p.N;
      const int col0 = cur_n0 + 64 * wc, lim = p.N - col0;   // the wave's 64 columns (unpaired tiles)
      // bias of the lane's 16 columns: as floats, or (forms that also hold the auxiliary pieces in registers)
      // packed as loaded and unpacked at each use
      constexpr bool kBiasPacked = BIAS && kXpose && (EPI == kEpiMul || EPI == kEpiAdd || EPI == kEpiRope);
      float bv[16];
      u32x4 bp0 = {0u, 0u, 0u, 0u}, bp1 = {0u, 0u, 0u, 0u};
      if (BIAS) {
        // N is a multiple of 16 and colq of 16: the 32 bytes are inside the staged row or skipped
        const u32x4* sb = reinterpret_cast<const u32x4*>(smem + kBiasLds + (col_ok ? colq * 2 : 0));
        if (col_ok) {
          bp0 = sb[0];
          bp1 = sb[1];
        }
        if (!kBiasPacked) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bv[2 * j] = lo_bf16(bp0[j]);
            bv[2 * j + 1] = hi_bf16(bp0[j]);
            bv[8 + 2 * j] = lo_bf16(bp1[j]);
            bv[8 + 2 * j + 1] = hi_bf16(bp1[j]);
          }
        }
      }
      auto bias_at = [&](int c) -> float {
        if (!BIAS) return 0.f;
        if (!kBiasPacked) return bv[c];
        const unsigned w = c < 8 ? bp0[c >> 1] : bp1[(c - 8) >> 1];
        return (c & 1) ? hi_bf16(w) : lo_bf16(w);
      };
      // The auxiliary operand (multiply / add forms) is read the way the output is written: lane L fetches
      // chunk L & 7 of rows L >> 3 and (L >> 3) + 8 of each 16-row piece (8 lanes per 128-byte line), all
      // pieces ahead of the first store so that no load queues behind a store; the pieces are turned into the
      // accumulator layout through the transposition buffer. Rows / columns past the end read as zeros.
      constexpr bool kAuxX = kXpose && (EPI == kEpiMul || EPI == kEpiAdd);
      u32x4 ax[2][4][2];
      if (kAuxX) {
        const bool cok = 8 * tch < lim;
        const unsigned cb = ((unsigned)col0 + 8u * tch) * 2u;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int r1 = cur_m0 + 128 * wr + 64 * mh + 16 * mt + trow, r2 = r1 + 8;
            const unsigned o1 = (cok && r1 < p.M) ? (unsigned)r1 * (unsigned)p.ld_aux * 2u + cb : 0x80000000u;
            const unsigned o2 = (cok && r2 < p.M) ? (unsigned)r2 * (unsigned)p.ld_aux * 2u + cb : 0x80000000u;
            ax[mh][mt][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, o1, 0, DPA_GEMM_AUX_LOAD);
            ax[mh][mt][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, o2, 0, DPA_GEMM_AUX_LOAD);
          }
      }
      // Rotary form: the lane's 16 columns are 8 (even, odd) pairs of one head (heads are 64 or 128 wide and
      // start at multiples of their width, colq is a multiple of 16), the same 8 table columns for every row of
      // the tile. A wave's 64 columns lie on one side of rope_cols: the test is on scalars, a plain branch.
      // The lane's rows are 16 apart inside one 128-row block, which never straddles two sequences
      // (rope_T % 128 == 0): cos / sin of the first row come from the tables, every later row's by the angle
      // addition with the tables' row 16 (32 registers instead of 128; seven steps in fp32).
      const bool rot = EPI == kEpiRope && cur_n0 + 64 * wc < p.rope_cols;
      f32x4 rc[2] = {}, rs[2] = {}, dc[2] = {}, ds[2] = {};   // (defined on both paths: nothing carried across tiles)
      if (EPI == kEpiRope) {
        if (rot) {
          int c0 = (colq & (p.rope_hs - 1)) >> 1;
          const int half = p.rope_hs >> 1;
          // opaque: c0 is the same for every tile of a lane, and the step constants loaded from it would
          // otherwise be hoisted out of the tile loop into registers the K loop has no room for
          asm volatile("" : "+v"(c0));
          const unsigned t = (unsigned)(cur_m0 + 128 * wr + l15) % (unsigned)p.rope_T;
          const long long o = (long long)(p.rope_pos0 + (int)t) * half + c0, o16 = 16ll * half + c0;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            rc[h2] = reinterpret_cast<const f32x4*>(p.rope_cos + o)[h2];
            rs[h2] = reinterpret_cast<const f32x4*>(p.rope_sin + o)[h2];
            dc[h2] = reinterpret_cast<const f32x4*>(p.rope_cos + o16)[h2];
            ds[h2] = reinterpret_cast<const f32x4*>(p.rope_sin + o16)[h2];
          }
        }
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int row0 = cur_m0 + 128 * wr + 64 * mh + 16 * mt;
          const int row = row0 + l15;
          const bool ok = col_ok && row < p.M;
          float v[16];
#pragma unroll
          for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * T + e] = BIAS ? acc[mh][mt][T][e] + bias_at(4 * T + e) : acc[mh][mt][T][e];   // (x + 0.0f is an instruction: -0.0)
          if (EPI == kEpiRope) {
            if (rot) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float c = rc[j >> 2][j & 3], sn = rs[j >> 2][j & 3];
                const float re = v[2 * j], im = v[2 * j + 1];
                v[2 * j] = re * c - im * sn;
                v[2 * j + 1] = re * sn + im * c;
                rc[j >> 2][j & 3] = c * dc[j >> 2][j & 3] - sn * ds[j >> 2][j & 3];
                rs[j >> 2][j & 3] = sn * dc[j >> 2][j & 3] + c * ds[j >> 2][j & 3];
              }
            }
          }
          if (EPI == kEpiMul || EPI == kEpiAdd) {
            u32x4 x0, x1;
            if (kAuxX) {
              *reinterpret_cast<u32x4*>(tbuf + tr0) = ax[mh][mt][0];
              *reinterpret_cast<u32x4*>(tbuf + tr1) = ax[mh][mt][1];
              asm volatile("" ::: "memory");
              x0 = *reinterpret_cast<const u32x4*>(tbuf + tw0);
              x1 = *reinterpret_cast<const u32x4*>(tbuf + tw1);
              asm volatile("" ::: "memory");
            } else {
              const unsigned xoff = ok ? ((unsigned)row * (unsigned)p.ld_aux + (unsigned)colq) * 2u : 0u;
              const u32x4* xp = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.aux_in) + xoff);
              x0 = xp[0];
              x1 = xp[1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // (pairs in packed fp32 arithmetic: half the multiply / add instructions)
              const f32x2 xa = {lo_bf16(x0[j]), hi_bf16(x0[j])}, xb = {lo_bf16(x1[j]), hi_bf16(x1[j])};
              f32x2 va = {v[2 * j], v[2 * j + 1]}, vb = {v[8 + 2 * j], v[8 + 2 * j + 1]};
              if (EPI == kEpiMul) {
                va *= xa; vb *= xb;
              } else {
                va += xa; vb += xb;
              }
              v[2 * j] = va[0]; v[2 * j + 1] = va[1]; v[8 + 2 * j] = vb[0]; v[8 + 2 * j + 1] = vb[1];
            }
          }
          if (EPI == kEpiSwiglu) {
            // v[0..7] = gate, v[8..15] = up of output columns colq .. colq + 7
            u32x4 o, hg, hu;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // silu(g) * u on a pair of columns in packed fp32 arithmetic
              const f32x2 g = {v[2 * j], v[2 * j + 1]}, u = {v[8 + 2 * j], v[8 + 2 * j + 1]};
              const f32x2 z = g * splat2(-1.4426950408889634f);
              const f32x2 den = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + splat2(1.0f);
              const f32x2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
              const f32x2 a2 = g * sg * u;
              o[j] = pack_bf16x2(a2[0], a2[1]);
              hg[j] = pack_bf16x2(g[0], g[1]);
              hu[j] = pack_bf16x2(u[0], u[1]);
            }
            const unsigned oc = ok ? ((unsigned)row * (unsigned)p.ldc + (unsigned)colq) * 2u : 0x80000000u;
            const unsigned og = ok ? ((unsigned)row * (unsigned)p.ld_aux + (unsigned)colq) * 2u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(o, rsC, oc, 0, kStoreAux);
            __builtin_amdgcn_raw_buffer_store_b128(hg, rsG, og, 0, kStoreAux);
            __builtin_amdgcn_raw_buffer_store_b128(hu, rsG, ok ? og + (unsigned)p.N * 2u : 0x80000000u, 0, kStoreAux);
            continue;
          }
          if (EPI == kEpiDswiglu) {
            // v = d(loss)/d(act) for columns colq .. colq + 15; [gate | up] from aux_in, [dgate | dup] to c
            const unsigned xoff = ok ? ((unsigned)row * (unsigned)p.ld_aux + (unsigned)colq) * 2u : 0u;
            const u32x4* gp = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.aux_in) + xoff);
            const u32x4* up = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.aux_in) + xoff + (ok ? (unsigned)p.N * 2u : 0u));
            const u32x4 g0 = gp[0], g1 = gp[1], u0 = up[0], u1 = up[1];
            u32x4 dg0, dg1, du0, du1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // pairs in packed fp32 arithmetic: d(up) = ds g sg, d(gate) = ds u sg (1 + g (1 - sg)), sg = sigmoid(g)
              const f32x2 gs[2] = {{lo_bf16(g0[j]), hi_bf16(g0[j])}, {lo_bf16(g1[j]), hi_bf16(g1[j])}};
              const f32x2 us[2] = {{lo_bf16(u0[j]), hi_bf16(u0[j])}, {lo_bf16(u1[j]), hi_bf16(u1[j])}};
              const f32x2 ds[2] = {{v[2 * j], v[2 * j + 1]}, {v[8 + 2 * j], v[8 + 2 * j + 1]}};
              f32x2 dgv[2], duv[2];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const f32x2 z = gs[e] * splat2(-1.4426950408889634f);
                const f32x2 den = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + splat2(1.0f);
                const f32x2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                const f32x2 dsg = ds[e] * sg;
                duv[e] = dsg * gs[e];
                const f32x2 w = pk_fma(gs[e], splat2(1.0f) - sg, splat2(1.0f));   // 1 + g (1 - sg)
                dgv[e] = dsg * us[e] * w;
              }
              dg0[j] = pack_bf16x2(dgv[0][0], dgv[0][1]);
              dg1[j] = pack_bf16x2(dgv[1][0], dgv[1][1]);
              du0[j] = pack_bf16x2(duv[0][0], duv[0][1]);
              du1[j] = pack_bf16x2(duv[1][0], duv[1][1]);
            }
            store_rows16(rsC, dg0, dg1, row0, col0, lim, (unsigned)p.ldc);
            store_rows16(rsC, du0, du1, row0, col0 + p.N, lim, (unsigned)p.ldc);
            continue;
          }
          u32x4 o0, o1;
          if (EPI == kEpiGelu) {
            u32x4 g0, g1;
            float a[16], d[16];
#if DPA_GELU_VARIANT >= 1
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
              f32x2 a2, d2;
              gelu_and_grad2(f32x2{v[j], v[j + 1]}, a2, d2);
              a[j] = a2[0]; a[j + 1] = a2[1];
              d[j] = d2[0]; d[j + 1] = d2[1];
            }
#else
#pragma unroll
            for (int j = 0; j < 16; ++j) gelu_and_grad(v[j], a[j], d[j]);
#endif
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
    DPA_TR(49);
    if (kLateBarrier && wr == 1) barrier();
    cur_m0 = nxt_m0;
    cur_n0 = nxt_n0;
    have_nxt = it + 2 < ntiles;
    if (have_nxt) {
      const TileCoord t2 = walk.at(wg, it + 2);
      nxt_m0 = t2.m0;
      nxt_n0 = t2.n0;
    }
    rsA_cur = rsA_nxt;
    if (have_nxt) rsA_nxt = a_rsrc(nxt_m0);
    if (GROUPED) {
      rsB_cur = rsB_nxt;
      if (have_nxt) rsB_nxt = group_rsrc(nxt_m0);
    }
  }
  if (SYNC == 0 && wr == 0) barrier();   // pairs with the extra barrier the second wave group started with
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (zero-fill) LDS-DMA before the LDS is released
#ifdef DPA_GEMM_TRACE
  if (blockIdx.x == kTraceWg && !BIAS) {
    __syncthreads();
    for (int i = tid; i < 8 * 128; i += 512) g_gemm_trace[i] = reinterpret_cast<unsigned long long*>(smem + kBiasLds)[i];
  }
#endif
}

// out[C, R] = in[R, C]^T through a padded 64 x 64 LDS tile (16-byte global accesses both ways).
__device__ __forceinline__ void transpose_tile(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
  __shared__ bf16_t tile[64][72];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i, r = id >> 3, ch = id & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < R && c0 + ch * 8 < C) v = *reinterpret_cast<const u32x4*>(in + (long long)(r0 + r) * C + c0 + ch * 8);
    *reinterpret_cast<u32x4*>(&tile[r][ch * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i, c = id >> 3, ch = id & 7;
    if (c0 + c < C && r0 + ch * 8 < R) {
      bf16_t t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = tile[ch * 8 + j][c];
      u32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (unsigned)t[2 * j] | ((unsigned)t[2 * j + 1] << 16);
      *reinterpret_cast<u32x4*>(out + (long long)(c0 + c) * R + r0 + ch * 8) = v;
    }
  }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int R, int C) {
  transpose_tile(in, out, R, C);
}

// The same for a list of equally shaped matrices in one launch (blockIdx.z = matrix): the transposed copies of
// all layers' weights are refreshed together after an optimizer step.
__global__ __launch_bounds__(256) void transpose_bf16_many_kernel(const long long* __restrict__ in,
                                                                  const long long* __restrict__ out, int R, int C) {
  transpose_tile(reinterpret_cast<const bf16_t*>(in[blockIdx.z]), reinterpret_cast<bf16_t*>(out[blockIdx.z]), R, C);
}

void transpose_bf16(const void* in, void* out, int R, int C, hipStream_t st) {
  if (R == 0 || C == 0) return;
  const dim3 grid((C + 63) / 64, (R + 63) / 64), block(256);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, block, 0, st, (const bf16_t*)in, (bf16_t*)out, R, C);
}

void transpose_bf16_many(const long long* in_ptrs, const long long* out_ptrs, int count, int R, int C, hipStream_t st) {
  if (R == 0 || C == 0 || count == 0) return;
  const dim3 grid((C + 63) / 64, (R + 63) / 64, count), block(256);
  hipLaunchKernelGGL(transpose_bf16_many_kernel, grid, block, 0, st, in_ptrs, out_ptrs, R, C);
}

bool gemm_nt_ok(long long M, long long N, long long K, long long lda, long long ldb, long long ldc, long long ld_aux,
                bool has_bias) {
  return gemm_nt_shape_ok(M, N, K, lda, ldb, ldc, ld_aux, has_bias);
}

bool gemm_nt_pair_ok(long long M, long long N, long long K, long long lda, long long ldb, long long ld_act,
                     long long ld_pair) {
  return gemm_nt_pair_shape_ok(M, N, K, lda, ldb, ld_act, ld_pair);
}

int g_gemm_nt_variant = 0;   // benchmark knob: bit 0 = one barrier per phase, bit 1 = drop the stores (timing only)
void gemm_nt_set_variant(int v) { g_gemm_nt_variant = v; }

long long gemm_nt_rows_per_launch(const GemmNtArgs& a, int epi) {
  long long ldm = a.ldc;       // the A operand is addressed per tile (any height); c and the auxiliary tensors by row offset
  if ((a.aux_in != nullptr || a.aux_out != nullptr) && a.ld_aux > ldm) ldm = a.ld_aux;
  long long align = 256;
  if (epi == kEpiRope && a.rope_T > 0) align = a.rope_T % 256 == 0 ? a.rope_T : 2ll * a.rope_T;   // rope_T % 128 == 0
  return gemm_nt_row_chunk(a.M, ldm, align);
}

static void gemm_nt_launch(GemmNtArgs a, int epi, int num_cus, hipStream_t st);

void gemm_nt(GemmNtArgs a, int epi, int num_cus, hipStream_t st) {
  if (a.M == 0 || a.N == 0) return;
  const long long chunk = gemm_nt_rows_per_launch(a, epi);
  if (chunk <= 0) {
    fprintf(stderr, "gemm_nt: a row range of the operands does not fit 2 GiB (callers check gemm_nt_rows_per_launch)\n");
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
    if (a.w_ptrs != nullptr) s.block_group = a.block_group + m0 / 256;   // grouped mode: one entry per 256-row block
    gemm_nt_launch(s, epi, num_cus, st);
  }
}

static void gemm_nt_launch(GemmNtArgs a, int epi, int num_cus, hipStream_t st) {
  a.tiles_m = (a.M + 255) / 256;
  const long long b_rows = epi == kEpiSwiglu ? 2ll * a.N : a.N;           // weight rows
  const long long c_cols = epi == kEpiDswiglu ? 2ll * a.N : a.N;          // columns of c
  const long long x_cols = (epi == kEpiSwiglu || epi == kEpiDswiglu) ? 2ll * a.N : a.N;   // columns of the auxiliary tensor
  a.tiles_n = epi == kEpiSwiglu ? (a.N + 127) / 128 : (a.N + 255) / 256;
  a.a_bytes = (unsigned)(((long long)a.M - 1) * a.lda * 2 + (long long)a.K * 2);
  a.b_bytes = (unsigned)((b_rows - 1) * a.ldb * 2 + (long long)a.K * 2);
  a.c_bytes = (unsigned)(((long long)a.M - 1) * a.ldc * 2 + c_cols * 2);
  a.aux_bytes = (unsigned)(((long long)a.M - 1) * a.ld_aux * 2 + x_cols * 2);
  if (g_gemm_nt_variant & 2) a.c_bytes = a.aux_bytes = 0;
  a.skew_ticks = (g_gemm_nt_variant >> 8);
  const int grid = gemm_nt_grid(a.tiles_m, a.tiles_n, num_cus);
  static bool attr = false;
#define DPA_GNT_ALL(F)                                                                  \
  F(kEpiNone, false) F(kEpiNone, true) F(kEpiGelu, false) F(kEpiGelu, true) F(kEpiMul, false) F(kEpiMul, true) \
  F(kEpiAdd, false) F(kEpiAdd, true) F(kEpiSwiglu, false) F(kEpiDswiglu, false) F(kEpiRope, false) F(kEpiRope, true)
  if (!attr) {
#define DPA_GNT_ATTR(E, B)                                                                                        \
  (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<E, B, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); \
  (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<E, B, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    DPA_GNT_ALL(DPA_GNT_ATTR)
#undef DPA_GNT_ATTR
    attr = true;
  }
#ifdef DPA_GEMM_ABLATE
  {
    const int abl = (g_gemm_nt_variant >> 4) & 15;
#define DPA_GNT_ABL(A)                                                                                       \
  if (abl == A) {                                                                                            \
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<kEpiNone, false, 0, A>,                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);                             \
    hipLaunchKernelGGL((gemm_nt_kernel<kEpiNone, false, 0, A>), dim3(grid), dim3(512), kLds, st, a);         \
    return;                                                                                                  \
  }
    DPA_GNT_ABL(1) DPA_GNT_ABL(2) DPA_GNT_ABL(3) DPA_GNT_ABL(4) DPA_GNT_ABL(8) DPA_GNT_ABL(9) DPA_GNT_ABL(10) DPA_GNT_ABL(12)
#undef DPA_GNT_ABL
  }
#endif
  if (a.w_ptrs != nullptr) {   // grouped mode: bias-free, epilogues none / GELU / multiply / SwiGLU / dSwiGLU
#define DPA_GNT_GROUPED(E)                                                                                          \
  if (epi == E) {                                                                                                   \
    static bool once = false;                                                                                       \
    if (!once) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<E, false, 0, 0, true>,                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLds);                                  \
      once = true;                                                                                                  \
    }                                                                                                               \
    hipLaunchKernelGGL((gemm_nt_kernel<E, false, 0, 0, true>), dim3(grid), dim3(512), kLds, st, a);                 \
    return;                                                                                                         \
  }
    DPA_GNT_GROUPED(kEpiNone) DPA_GNT_GROUPED(kEpiGelu) DPA_GNT_GROUPED(kEpiMul) DPA_GNT_GROUPED(kEpiSwiglu) DPA_GNT_GROUPED(kEpiDswiglu)
#undef DPA_GNT_GROUPED
    return;
  }
  const bool bias = a.bias != nullptr;
#define DPA_GNT_LAUNCH(E, B)                                                                              \
  if (epi == E && bias == B) {                                                                            \
    if (g_gemm_nt_variant & 1)                                                                            \
      hipLaunchKernelGGL((gemm_nt_kernel<E, B, 1>), dim3(grid), dim3(512), kLds, st, a);                  \
    else                                                                                                  \
      hipLaunchKernelGGL((gemm_nt_kernel<E, B, 0>), dim3(grid), dim3(512), kLds, st, a);                  \
    return;                                                                                               \
  }
  DPA_GNT_ALL(DPA_GNT_LAUNCH)
#undef DPA_GNT_LAUNCH
#undef DPA_GNT_ALL
}

}  // namespace dpa
