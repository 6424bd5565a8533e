// Causal flash attention for gfx950 (MI355X): forward, dK/dV backward and dQ backward.
//
// Design (wave64, v_mfma_f32_32x32x16_bf16, LDS-tiled, no score matrix in memory):
//  * Scores are computed TRANSPOSED, S^T = K * Q^T, so every lane owns one query column and
//    the keys run along its accumulator registers: row max / row sum are in-lane reductions
//    plus one exchange between the two 32-lane halves, and the probability tile is already in
//    the register layout the next MFMA wants as its B operand (O^T = V^T * P^T). Nothing but
//    K and V ever goes through LDS in the forward.
//  * K/V (forward, dQ) and Q/dO (dK/dV) tiles go global -> LDS by LDS-DMA (buffer_load ... lds issued
//    from inline assembly: no staging registers), the transfer of tile t+1 issued before the compute
//    on tile t (double-buffered LDS, one s_waitcnt vmcnt(0) + barrier per tile). The XOR swizzle of
//    the LDS image is applied on the global side of the transfer; one image serves both the row reads
//    (ds_read_b128, operand A of the score products) and the transposed reads (ds_read_b64_tr_b16,
//    operand A of the products that sum over keys/queries), conflict-free for both.
//  * Key width DK and value width DV are separate template parameters: (64, 64) and (128, 128) for
//    ordinary heads, (128, 64) and (256, 128) for latent attention with decoupled rotary keys (built in
//    csrc/attention_wide.hip). Wide keys read their operand fragments in batches of 8 K-steps.
//  * GQA is native: a query head reads the K/V head of its group directly (no repeat), and
//    the dK/dV kernel accumulates over the query heads of the group in registers.
//  * Backward is split in two kernels so no cross-workgroup reduction (atomics) is needed:
//    one workgroup owns 128 keys and sweeps the queries (dK, dV), the other owns 128 queries
//    and sweeps the keys (dQ). P is recomputed from the saved log-sum-exp.
//  * Attention dropout (template flag, off in the default instantiation) is applied to the
//    probability registers from a stateless per-pair hash, so no mask is stored or exchanged.
//  * The causal mask is bottom-right aligned (query i sees keys <= i + Tk - Tq), which is what
//    KV-cache decoding needs. Heaviest tiles are launched first.
//
// Replaces `F.scaled_dot_product_attention(is_causal=True)` and the `repeat_interleave` GQA
// expansion of the reference (reference: single-gpu/model.py:144-149).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "mfma_tiles.h"

namespace dpa {

#ifndef DPA_ATTN_NT_STORE
#define DPA_ATTN_NT_STORE 0   // output tiles (O, dQ, dK, dV) stored with the non-temporal policy: measured, no change in the step
#endif

// A wave's [32][HS] output tile, held transposed (rows = d on registers, column = lane), is
// turned into whole rows through LDS and stored with 16-byte accesses. The 16-byte chunks of a row are
// XOR-swizzled with the row index: without it the 32 lanes of a half-wave write one bank column (row
// stride = 128 or 256 bytes), which the bank-conflict counter showed as ~1 conflict cycle per LDS
// instruction of the whole kernel although the main loop's patterns have none.
// With rotary tables (rc, rs: fp32 [positions, HS / 2]; rc == nullptr: none) the stored rows are rotated BACK by
// the angles of position pos0 + row: the tile is a gradient with respect to rotated q or k rows, and what
// leaves is the gradient with respect to the rows before the rotation (the transpose of a rotation is its
// inverse). A lane holds 4 whole (even, odd) pairs of one row at that point, once per workgroup.
template <int HS>
__device__ __forceinline__ void store_tile_T(char* wave_lds, const f32x16 (&acc)[HS / 32], float mul, bf16_t* out,
                                             long long ts, int row0, int nrows, int lane, const float* rc = nullptr,
                                             const float* rs = nullptr, int pos0 = 0) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < HS / 32; ++dt) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = 32 * dt + 8 * g4 + 4 * hh;
      u32x2 w;
      w[0] = pack_bf16x2(acc[dt][4 * g4] * mul, acc[dt][4 * g4 + 1] * mul);
      w[1] = pack_bf16x2(acc[dt][4 * g4 + 2] * mul, acc[dt][4 * g4 + 3] * mul);
      const int chunk = (d0 >> 3) ^ (r & (HS / 8 - 1));
      *reinterpret_cast<u32x2*>(wave_lds + r * (HS * 2) + chunk * 16 + 8 * hh) = w;
    }
  }
  constexpr int CPR = HS / 8;
#pragma unroll
  for (int i = 0; i < 32 * CPR / 64; ++i) {
    const int id = lane + 64 * i, row = id / CPR, ch = id % CPR;
    u32x4 v = *reinterpret_cast<const u32x4*>(wave_lds + row * (HS * 2) + (ch ^ (row & (CPR - 1))) * 16);
    if (rc != nullptr && row0 + row < nrows) {
      const long long o = (long long)(pos0 + row0 + row) * (HS / 2) + ch * 4;
      const f32x4 c = *reinterpret_cast<const f32x4*>(rc + o), sn = *reinterpret_cast<const f32x4*>(rs + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float re = lo_bf16(v[j]), im = hi_bf16(v[j]);
        v[j] = pack_bf16x2(re * c[j] + im * sn[j], im * c[j] - re * sn[j]);
      }
    }
    if (row0 + row < nrows) {
      u32x4* dst = reinterpret_cast<u32x4*>(out + (long long)(row0 + row) * ts + ch * 8);
      if (DPA_ATTN_NT_STORE) __builtin_nontemporal_store(v, dst);
      else *dst = v;
    }
  }
}

constexpr float kLog2e = 1.4426950408889634f;

// Max / sum of x over the two lanes 32 apart, in both (v_permlane32_swap: a register exchange between the
// wave's halves, no trip through LDS as __shfl_xor takes). After swapping the upper half of one copy with the
// lower half of the other, one copy holds the lower half's x in both halves and the other the upper half's.
// Issued from assembly: with the same value as both operands of the builtin the compiler folds the two
// results into one.
__device__ __forceinline__ void swap_halves(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float max_halves(float x) {
  float a = x, b = x;
  swap_halves(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float sum_halves(float x) {
  float a = x, b = x;
  swap_halves(a, b);
  return a + b;
}
constexpr float kDefer = 6.0f;   // deferred-rescale threshold of the forward, in log2 units

// Dropout mask: keep(q, k) for one (batch, head) is a 32-bit mix of the pair's index with a per-head
// key derived from the 64-bit seed. Stateless, so the forward and both backward kernels (which visit
// the pairs in different orders and layouts) regenerate the same mask; about ten integer operations
// per score.
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned drop_key(unsigned long long s, int bh) {
  return mix32((unsigned)s ^ ((unsigned)bh * 0x27D4EB2Fu)) ^ (unsigned)(s >> 32);
}
__device__ __forceinline__ unsigned drop_key(const long long* seed, int bh) { return drop_key((unsigned long long)*seed, bh); }
__device__ __forceinline__ unsigned drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}
__device__ __forceinline__ bool drop_keep(unsigned key, int q, int k, unsigned thresh) {
  return mix32(((unsigned)q * 0x9E3779B1u + (unsigned)k) ^ key) >= thresh;
}

// Which (tile rank, batch*head) a workgroup works on. rank 0 is the heaviest tile of its head.
// group == 0: rank-major over all heads (all heaviest tiles first). group == G: workgroups are
// re-labelled so that each XCD (own L2) owns a contiguous run of heads, and inside it the tiles of G
// heads at a time run rank-major: all tiles that read one head's K/V (or Q/dO) are in flight
// together on one XCD and share the operand through its L2, while the heavy-first order inside a
// group keeps the tail short.
__device__ __forceinline__ void tile_order(int bid, int nwg, int ntile, int BH, int group, int& rank, int& bh) {
  if (group <= 0 || BH % group != 0) {
    rank = bid / BH;
    bh = bid % BH;
    return;
  }
  const int id = xcd_remap(bid, nwg);
  const int per = group * ntile;
  const int grp = id / per, rem = id % per;
  rank = rem / group;
  bh = grp * group + rem % group;
}

// ------------------------------------------------------------------------------------------
// [64][HS] operand tiles by LDS-DMA (buffer_load ... lds): no staging registers, no address or
// zero-fill arithmetic in the tile loop. A piece is the 1 KiB one wave instruction moves (64 lanes x 16
// bytes, contiguous in LDS); the XOR swizzle of the tile image is applied on the global side, the
// per-tile advance is the instruction's scalar offset. Issued from assembly: the compiler would drain
// an LDS-DMA builtin (vmcnt(0)) before the next LDS read, i.e. right after issuing it.
// ------------------------------------------------------------------------------------------
#ifndef DPA_ATTN_LEAN
#define DPA_ATTN_LEAN 1   // LDS-DMA pieces without the M0 save / restore pair (benchmark knob)
#endif
typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* ptr) {
  const unsigned long long a = (unsigned long long)ptr;
  rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = 0xffffffffu;   // no range check: rows past the end are redirected by the caller
  r[3] = 0x00020000u;
  return r;
}

// (M0 is written and not restored: nothing else in these kernels reads it. Precomputed per-lane fragment addresses, which
// took 15 % off the one-wave-per-SIMD weight-gradient kernel, were tried on the forward's tile loop too: no change with
// three waves per SIMD, profiles/r3_attention_lean_loop_experiment.txt)
__device__ __forceinline__ void dma_piece(unsigned lds_addr, unsigned voff, rsrc_t rs, unsigned soff) {
#if DPA_ATTN_LEAN
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, %3 offen lds"
      :
      : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
      : "memory", "m0");
#else
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
      : "memory");
#endif
}

__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_dma_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// K/V stages of the forward. Three (a tile requested two tiles ahead, 48 KiB per workgroup at head size 64) were
// measured 25 % SLOWER than two (186 -> 234 us at B64 T1024 nh12): the wait for a tile requested one tile ahead was
// short already, and the third stage costs the third workgroup per CU. Kept as a build knob.
#ifndef DPA_ATTN_FWD_STAGES
#define DPA_ATTN_FWD_STAGES 2
#endif
__host__ __device__ constexpr int fwd_stages(int dk, int dv) {
  return (DPA_ATTN_FWD_STAGES == 3 && 64 * dk * 2 + 64 * dv * 2 <= 16384) ? 3 : 2;
}

template <int HS>
struct TileDma {
  static constexpr int CPR = HS / 8;        // 16-byte chunks per row
  static constexpr int RPP = 64 / CPR;      // rows per piece
  static constexpr int NP = 64 / RPP / 4;   // pieces per wave
  unsigned voff[NP];                        // this lane's source byte offset inside a tile, by piece
  unsigned ts2;                             // row stride in bytes
  int wave, lane;
  __device__ __forceinline__ void init(long long ts, int wave_, int lane_) {
    ts2 = (unsigned)(ts * 2);
    wave = wave_;
    lane = lane_;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      const int row = (wave * NP + pc) * RPP + lane / CPR;
      voff[pc] = row * ts2 + 16 * swz<HS>(row, lane % CPR);
    }
  }
  // Rows [row0, row0 + 64) of the matrix behind `rs` -> tile at LDS address `lds`. Rows at or past
  // `nrows` re-read row `row0` (finite values; whatever is computed from them is masked).
  __device__ __forceinline__ void issue(rsrc_t rs, int row0, int nrows, unsigned lds) const {
    const unsigned soff = __builtin_amdgcn_readfirstlane(row0 * ts2);
    if (row0 + 64 <= nrows) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) dma_piece(lds + (wave * NP + pc) * 1024, voff[pc], rs, soff);
    } else {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
        const int row = (wave * NP + pc) * RPP + lane / CPR;
        const unsigned v = (row0 + row < nrows) ? voff[pc] : 16u * swz<HS>(row, lane % CPR);
        dma_piece(lds + (wave * NP + pc) * 1024, v, rs, soff);
      }
    }
  }
};

// 8 bf16 of a row, times c, as an MFMA operand.
__device__ __forceinline__ bfrag scaled_frag(const bf16_t* src, float c) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(src);
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(lo_bf16(v[j]) * c, hi_bf16(v[j]) * c);
  return __builtin_bit_cast(bfrag, o);
}

// Sums of probabilities a tile may reach before the running reference point of its rows is moved, and the
// least sum the first tile must reach for the initial reference (0) to be kept.
constexpr float kSumHi = 1.0e9f, kSumLo = 1.0e-9f;

// Timing-only ablations of the forward at head size 64 (benchmarks/ab_build.py ... -DDPA_ATTN_ABLATE=n, wrong
// results): 1 = no exponential and never a redo, 2 = additionally no P V products. Measured at B64 T1024: 173.1 us
// -> 170.3 (1) -> 138.9 (2): the exponentials are already hidden, a product is not. 3 = the V fragments come by row
// reads (ds_read_b128, the same bytes at the faster of the two LDS access patterns), 4 = no V fragment reads at all,
// 5 = no K fragment reads: what the LDS pipe itself costs. 6 = no K/V transfers after the first tile's (the loop
// computes on whatever the LDS holds): what the trip from L2 / HBM into the LDS costs.
#ifndef DPA_ATTN_ABLATE
#define DPA_ATTN_ABLATE 0
#endif
__device__ __forceinline__ float fwd_exp2(float x) {
  return (DPA_ATTN_ABLATE == 1 || DPA_ATTN_ABLATE == 2) ? x : __builtin_amdgcn_exp2f(x);
}

// Timing trace of the forward (benchmark builds with -DDPA_ATTN_TRACE only): the four waves of one workgroup stamp
// the shader clock at seven points of each of their first 16 key tiles into LDS; at the end the low words replace
// the first 448 log-sum-exp outputs of (batch 0, head 0), where the benchmark script reads them.
#ifdef DPA_ATTN_TRACE
#define DPA_ATR(slot)                                                                                       \
  do {                                                                                                      \
    if (tracing && t < 16) {                                                                                \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
      if (lane == 0) trace_s[wave][t][slot] = (unsigned)t_;                                                 \
    }                                                                                                       \
  } while (0)
#else
#define DPA_ATR(slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// Forward
//
// Softmax bookkeeping. Q is multiplied by scale * log2(e) once, when its fragments are loaded, and the
// score MFMA chain starts from an accumulator that holds -m (m = the row's reference point in log2
// units), so the chain's result is already the argument of exp2: a score costs one v_exp, one add for
// the row sum and half a convert. No row maximum is taken on the normal path. m starts at 0 and stays
// until a tile's probabilities sum past kSumHi for some row (or the first tile's stay below kSumLo);
// that tile is then redone the classic way: raw scores, row maximum, O and l rescaled to the new m.
// Probabilities up to ~1e9 instead of <= 1 cost nothing in accuracy: fp32 sums and the bf16 P operand
// have the same relative precision at any magnitude.
// ------------------------------------------------------------------------------------------
template <int DK, int DV, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  constexpr int NK = DK / 16, ND = DV / 32, KTILE = 64 * DK * 2, VTILE = 64 * DV * 2, STAGE = KTILE + VTILE;
  constexpr int NST = fwd_stages(DK, DV);   // K/V tiles in LDS: the transfer of tile t + NST - 1 is issued during tile t
  constexpr int KCH = NK > 8 ? 8 : NK;   // K fragments requested together (wide keys: in two or more batches)
  constexpr int KSUB = 2;   // head size 64: 32-key sub-tiles whose fragments are in registers together
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const int BH = p.B * p.nh;
  const int nqt = (p.Tq + 127) / 128;
  int rank, bh;
  tile_order((int)blockIdx.x, (int)gridDim.x, nqt, BH, p.group, rank, bh);
  const int qt = nqt - 1 - rank;  // heaviest (latest) query tiles first
  const int b = bh / p.nh, h = bh % p.nh, kvh = h / (p.nh / p.nkv);
  int Tk = p.Tk;
  if (p.kv_len != nullptr) {   // fill level of a preallocated cache (clamped to what is allocated)
    const int n = *p.kv_len;
    Tk = n < p.Tq ? p.Tq : (n > p.Tk ? p.Tk : n);
  }
  const int q0 = qt * 128, off = Tk - p.Tq;
  const bf16_t* qp = (const bf16_t*)p.q + b * p.q_bs + h * DK;
  const rsrc_t krs = make_rsrc((const bf16_t*)p.k + b * p.k_bs + kvh * DK);
  const rsrc_t vrs = make_rsrc((const bf16_t*)p.v + b * p.v_bs + kvh * DV);

  int kv_end = q0 + 128 + off;
  if (kv_end > Tk) kv_end = Tk;
  const int ntiles = (kv_end + 63) / 64;
  const int wave_qmax = q0 + wave * 32 + 31;

  TileDma<DK> kd;
  TileDma<DV> vd;
  kd.init(p.k_ts, wave, lane);
  vd.init(p.v_ts, wave, lane);
  kd.issue(krs, 0, Tk, lds0);
  vd.issue(vrs, 0, Tk, lds0 + KTILE);

  const float c = p.scale * kLog2e;
  const int qrow = q0 + wave * 32 + r;
  bfrag qf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    qf[ks] = zero_frag();
    if (qrow < p.Tq) qf[ks] = scaled_frag(qp + (long long)qrow * p.q_ts + ks * 16 + hh * 8, c);
  }

  if (NST == 3 && ntiles > 1) {   // (after the Q loads: the first tile's wait leaves exactly this tile's pieces in flight)
    kd.issue(krs, 64, Tk, lds0 + STAGE);
    vd.issue(vrs, 64, Tk, lds0 + STAGE + KTILE);
  }

  f32x16 oacc[ND], negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
  float m = 0.f, l = 0.f;   // m in log2 units
  int qlim = qrow + off;
  if (qlim > Tk - 1) qlim = Tk - 1;
  const unsigned dkey = DROP ? drop_key(p.seed, bh) : 0u, dthr = DROP ? drop_threshold(p.p_drop) : 0u;

#ifdef DPA_ATTN_TRACE
  __shared__ unsigned trace_s[4][16][7];
  const bool tracing = blockIdx.x == 300 && DK == 64;   // a workgroup of the first batch with all 16 tiles
  if (tracing) for (int i = tid; i < 4 * 16 * 7; i += 256) (&trace_s[0][0][0])[i] = 0u;
#endif
  int st_cur = 0;   // t % NST
  for (int t = 0; t < ntiles; ++t) {
    DPA_ATR(0);
    // three stages: tile t was requested two tiles ago and only tile t + 1's pieces may still be in flight
    if (NST == 3 && t + 1 < ntiles) wait_dma_but<TileDma<DK>::NP + TileDma<DV>::NP>();
    else wait_dma();
    __syncthreads();   // tile t has landed for every wave; every wave is done with tile t - 1's buffer
    DPA_ATR(1);
    if (t + NST - 1 < ntiles && DPA_ATTN_ABLATE != 6) {
      int sn = st_cur + NST - 1;
      if (sn >= NST) sn -= NST;
      const unsigned nb = lds0 + sn * STAGE;
      kd.issue(krs, (t + NST - 1) * 64, Tk, nb);
      vd.issue(vrs, (t + NST - 1) * 64, Tk, nb + KTILE);
    }
    DPA_ATR(2);
    const char* kb = smem + st_cur * STAGE;
    st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    const char* vb = kb + KTILE;
    const int kv0 = t * 64;
    if constexpr (DK == 64 && DV == 64) {
    // Head size 64: both 32-key sub-tiles of the tile together (3 waves per SIMD fit; measured 4 % faster than
    // one sub-tile at a time).
    if (kv0 <= wave_qmax + off) {  // wave-uniform: some key of this tile is visible to the wave
      const bool need_mask = (kv0 + 63 > q0 + wave * 32 + off) || (kv0 + 64 > Tk);
      // s = scores - m, from the accumulator `init`. LDS reads go out in batches ahead of their products
      // (all K fragments before the first MFMA): a read issued right before its MFMA, which is the
      // compiler's pressure-minimising order, exposes the LDS latency once per MFMA.
      f32x16 s[2];
      auto scores = [&](const f32x16& init) {
#pragma unroll
        for (int sb = 0; sb < 2; sb += KSUB) {
          bfrag kf[KSUB][NK];
#pragma unroll
          for (int u = 0; u < KSUB; ++u)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) kf[u][ks] = DPA_ATTN_ABLATE == 5 ? qf[ks] : row_frag<DK>(kb, (sb + u) * 32 + r, ks, hh);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < KSUB; ++u) {
            s[sb + u] = mfma32(kf[u][0], qf[0], init);
#pragma unroll
            for (int ks = 1; ks < NK; ++ks) s[sb + u] = mfma32(kf[u][ks], qf[ks], s[sb + u]);
          }
        }
        if (need_mask) {   // key > qlim, as one compare of a constant against a per-lane bound
          const int bound = qlim - kv0 - 4 * hh;
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (sub * 32 + acc_row(i, 0) > bound) s[sub][i] = -INFINITY;
        }
      };
      scores(negm);
      DPA_ATR(3);
      bfrag vf[KSUB][2][ND];
#pragma unroll
      for (int u = 0; u < KSUB; ++u)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < ND; ++dt) {
            if (DPA_ATTN_ABLATE == 3) vf[u][s2][dt] = row_frag<DV>(vb, u * 32 + r, 2 * dt + s2, hh);
            else if (DPA_ATTN_ABLATE == 4) vf[u][s2][dt] = qf[2 * dt + s2];
            else vf[u][s2][dt] = tr_frag<DV>(vb, u * 32, s2, dt, lane);
          }
      __builtin_amdgcn_sched_barrier(0);
      float rs = 0.f;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[sub][i] = fwd_exp2(s[sub][i]);
          rs += s[sub][i];
        }
      DPA_ATR(4);
      bool redo = !__all(rs <= kSumHi);
      if (t == 0) redo = redo || !__all(sum_halves(rs) >= kSumLo);
      if (DPA_ATTN_ABLATE) redo = false;
      if (redo) {
        f32x16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.f;
        scores(zero);
        float mx = s[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[sub][i]);
        mx = max_halves(mx);
        // The first tile replaces the initial guess outright (O and l are still zero); later the
        // reference only moves up, and what was accumulated at the old one is rescaled.
        const float m_new = t == 0 ? mx : fmaxf(m, mx);
        const float alpha = t == 0 ? 0.f : __builtin_amdgcn_exp2f(m - m_new);
        l *= alpha;
        m = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = -m_new;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
        rs = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            s[sub][i] = __builtin_amdgcn_exp2f(s[sub][i] - m_new);
            rs += s[sub][i];
          }
      }
      l += rs;   // the softmax normaliser is that of the undropped row
      if (DROP) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (!drop_keep(dkey, qrow, kv0 + sub * 32 + acc_row(i, hh), dthr)) s[sub][i] = 0.f;
      }
#pragma unroll
      for (int sb = 0; sb < 2; sb += KSUB) {
        if (sb > 0) {
#pragma unroll
          for (int u = 0; u < KSUB; ++u)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int dt = 0; dt < ND; ++dt) vf[u][s2][dt] = tr_frag<DV>(vb, (sb + u) * 32, s2, dt, lane);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < KSUB; ++u)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bfrag pb = acc_frag(s[sb + u], s2);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
              if (DPA_ATTN_ABLATE == 2) asm volatile("" : "+v"(oacc[dt]) : "v"(vf[u][s2][dt]), "v"(pb));
              else oacc[dt] = mfma32(vf[u][s2][dt], pb, oacc[dt]);
            }
          }
      }
      DPA_ATR(5);
    }
    } else {
    // Wider heads: one sub-tile at a time, which keeps the kernel at two waves per SIMD (+37 % at head size 128).
    const int wq0 = q0 + wave * 32 + off;
    // 32-key sub-tiles with a key visible to the wave, one at a time (a rolled loop: the accumulators stay
    // in place, the diagonal's second sub-tile is skipped by the waves that see none of it, and the mask
    // arithmetic runs only on the sub-tile the diagonal or the end of the keys crosses).
    const int nsub = kv0 > wave_qmax + off ? 0 : (kv0 + 32 > wave_qmax + off ? 1 : 2);
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      const int ks0 = kv0 + sub * 32;
      const char* ksub = kb + sub * (32 * DK * 2);
      const char* vsub = vb + sub * (32 * DV * 2);
      const bool mk = (ks0 + 31 > wq0) || (ks0 + 32 > Tk);
      // s = scores - m, from the accumulator `init`. The K fragments are all requested before the
      // first MFMA: a read issued right before its MFMA, which is the compiler's pressure-minimising
      // order, exposes the LDS latency once per MFMA.
      f32x16 s;
      auto scores = [&](const f32x16& init) {
        s = init;
#pragma unroll
        for (int k0 = 0; k0 < NK; k0 += KCH) {
          bfrag kf[KCH];
#pragma unroll
          for (int ks = 0; ks < KCH; ++ks) kf[ks] = row_frag<DK>(ksub, r, k0 + ks, hh);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < KCH; ++ks) s = mfma32(kf[ks], qf[k0 + ks], s);
        }
        if (mk) {
          const int bound = qlim - ks0 - 4 * hh;
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (acc_row(i, 0) > bound) s[i] = -INFINITY;
        }
      };
      scores(negm);
      bfrag vf[2][ND];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) vf[s2][dt] = tr_frag<DV>(vsub, 0, s2, dt, lane);
      __builtin_amdgcn_sched_barrier(0);
      float rs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[i] = __builtin_amdgcn_exp2f(s[i]);
        rs += s[i];
      }
      const bool first = t == 0 && sub == 0;   // holds key 0, which every query sees
      bool redo = !__all(rs <= kSumHi);
      if (first) redo = redo || !__all(sum_halves(rs) >= kSumLo);
      if (redo) {
        f32x16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.f;
        scores(zero);
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        mx = max_halves(mx);
        // The first sub-tile replaces the initial guess outright (O and l are still zero); later the
        // reference only moves up, and what was accumulated at the old one is rescaled.
        const float m_new = first ? mx : fmaxf(m, mx);
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(m - m_new);
        l *= alpha;
        m = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = -m_new;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
        rs = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[i] = __builtin_amdgcn_exp2f(s[i] - m_new);
          rs += s[i];
        }
      }
      l += rs;   // the softmax normaliser is that of the undropped row
      if (DROP) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (!drop_keep(dkey, qrow, ks0 + acc_row(i, hh), dthr)) s[i] = 0.f;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bfrag pb = acc_frag(s, s2);
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) oacc[dt] = mfma32(vf[s2][dt], pb, oacc[dt]);
      }
    }
    }
  }

  l = sum_halves(l);
  const float inv = (DROP ? 1.f / (1.f - p.p_drop) : 1.f) / l;
  if (hh == 0 && qrow < p.Tq)
    p.lse[((long long)b * p.nh + h) * p.Tq + qrow] = (m + __builtin_amdgcn_logf(l)) * (1.f / kLog2e);
  __syncthreads();  // all waves are done with the K/V tiles before the region is reused
  bf16_t* op = (bf16_t*)p.o + b * p.o_bs + h * DV;
  store_tile_T<DV>(smem + wave * 32 * DV * 2, oacc, inv, op, p.o_ts, q0 + wave * 32, p.Tq, lane);
#ifdef DPA_ATTN_TRACE
  if (tracing) {
    __syncthreads();
    for (int i = tid; i < 4 * 16 * 7; i += 256) reinterpret_cast<unsigned*>(p.lse)[i] = (&trace_s[0][0][0])[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------
// Backward. P is recomputed from the forward's log-sum-exp; the row constants ride in the MFMA
// chains' initial accumulators: S' = (c q) k - lse * log2(e) makes p = exp2(S') one instruction, and
// dP' = dO v - delta makes dS = p * dP' one multiply. The operand a wave keeps in registers for the
// whole kernel (Q in the dQ kernel, K in the dK/dV kernel) carries c = scale * log2(e).
//
// The dQ kernel runs first and publishes the row constants for the dK/dV kernel in `stats`
// ([2][B * nh][Tq rounded up to 128] floats): -delta (delta = rowsum(dO * O)) and -lse * log2(e), with
// 0 / -inf in the rows past Tq, so that the dK/dV kernel moves them into LDS by DMA as they are.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int stat_rows(int Tq) { return (Tq + 127) / 128 * 128; }

// 64 floats (one per lane) from rs[voff + soff] to LDS [lds_addr, lds_addr + 256).
__device__ __forceinline__ void dma_dwords(unsigned lds_addr, unsigned voff, rsrc_t rs, unsigned soff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dword %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
      : "memory");
}

// dQ: one workgroup = 128 queries, sweeps the keys.
// (Holding the head-size-64 backward kernels to 168 registers for a third wave per SIMD, amdgpu_waves_per_eu(3), was
// measured 34 % slower: profiles/r3_attention_bwd_three_waves_experiment.txt.)
template <int DK, int DV, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
  constexpr int NK = DK / 16, NV = DV / 16, ND = DK / 32, KTILE = 64 * DK * 2, VTILE = 64 * DV * 2, STAGE = KTILE + VTILE;
  constexpr int KCH = NK > 8 ? 8 : NK, DCH = ND > 4 ? 4 : ND;   // fragment batches (wide keys: several)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const int BH = p.B * p.nh;
  const int nqt = (p.Tq + 127) / 128;
  int rank, bh;
  tile_order((int)blockIdx.x, (int)gridDim.x, nqt, BH, p.group, rank, bh);
  const int qt = nqt - 1 - rank;
  const int b = bh / p.nh, h = bh % p.nh, kvh = h / (p.nh / p.nkv);
  const int q0 = qt * 128, off = p.Tk - p.Tq;
  const bf16_t* qp = (const bf16_t*)p.q + b * p.q_bs + h * DK;
  const bf16_t* dop = (const bf16_t*)p.dout + b * p.do_bs + h * DV;
  const rsrc_t krs = make_rsrc((const bf16_t*)p.k + b * p.k_bs + kvh * DK);
  const rsrc_t vrs = make_rsrc((const bf16_t*)p.v + b * p.v_bs + kvh * DV);

  int kv_end = q0 + 128 + off;
  if (kv_end > p.Tk) kv_end = p.Tk;
  const int ntiles = (kv_end + 63) / 64;
  const int wave_qmax = q0 + wave * 32 + 31;

  TileDma<DK> kd;
  TileDma<DV> vd;
  kd.init(p.k_ts, wave, lane);
  vd.init(p.v_ts, wave, lane);
  kd.issue(krs, 0, p.Tk, lds0);
  vd.issue(vrs, 0, p.Tk, lds0 + KTILE);

  const float c = p.scale * kLog2e;
  const int qrow = q0 + wave * 32 + r;
  bfrag qf[NK], dof[NV];
  float nlse = -INFINITY, dlt = 0.f;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) qf[ks] = zero_frag();
#pragma unroll
  for (int ks = 0; ks < NV; ++ks) dof[ks] = zero_frag();
  if (qrow < p.Tq) {
    // delta = rowsum(dO * O) from the dO fragments the kernel holds anyway (each lane half owns half
    // of the row).
    const bf16_t* orow = (const bf16_t*)p.o + b * p.o_bs + h * DV + (long long)qrow * p.o_ts;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qf[ks] = scaled_frag(qp + (long long)qrow * p.q_ts + ks * 16 + hh * 8, c);
#pragma unroll
    for (int ks = 0; ks < NV; ++ks) {
      dof[ks] = *reinterpret_cast<const bfrag*>(dop + (long long)qrow * p.do_ts + ks * 16 + hh * 8);
      const u32x4 ov = *reinterpret_cast<const u32x4*>(orow + ks * 16 + hh * 8);
      const u32x4 dv = __builtin_bit_cast(u32x4, dof[ks]);
#pragma unroll
      for (int j = 0; j < 4; ++j) dlt += lo_bf16(ov[j]) * lo_bf16(dv[j]) + hi_bf16(ov[j]) * hi_bf16(dv[j]);
    }
    nlse = -p.lse[((long long)b * p.nh + h) * p.Tq + qrow] * kLog2e;
  }
  dlt = sum_halves(dlt);
  if (hh == 0) {   // every row of the tile, the padding past Tq included
    const long long srow = (long long)bh * stat_rows(p.Tq) + qrow;
    p.delta[srow] = -dlt;
    p.delta[(long long)BH * stat_rows(p.Tq) + srow] = nlse;
  }
  f32x16 negl, negd;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    negl[i] = nlse;
    negd[i] = -dlt;
  }

  f32x16 dqacc[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dqacc[dt][i] = 0.f;
  int qlim = qrow + off;
  if (qlim > p.Tk - 1) qlim = p.Tk - 1;
  const unsigned dkey = DROP ? drop_key(p.seed, bh) : 0u, dthr = DROP ? drop_threshold(p.p_drop) : 0u;
  const float rkeep = DROP ? 1.f / (1.f - p.p_drop) : 1.f;

  for (int t = 0; t < ntiles; ++t) {
    wait_dma();
    __syncthreads();
    if (t + 1 < ntiles) {
      const unsigned nb = lds0 + ((t + 1) & 1) * STAGE;
      kd.issue(krs, (t + 1) * 64, p.Tk, nb);
      vd.issue(vrs, (t + 1) * 64, p.Tk, nb + KTILE);
    }
    const char* kb = smem + (t & 1) * STAGE;
    const char* vb = kb + KTILE;
    const int kv0 = t * 64;
    const int wq0 = q0 + wave * 32 + off;
    // 32-key sub-tiles with a key visible to the wave. A rolled loop: with one `if` per unrolled
    // sub-tile the compiler copies all the dQ accumulators at every join.
    const int nsub = kv0 > wave_qmax + off ? 0 : (kv0 + 32 > wave_qmax + off ? 1 : 2);
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      const int ks0 = kv0 + sub * 32;
      const char* ksub = kb + sub * (32 * DK * 2);
      const char* vsub = vb + sub * (32 * DV * 2);
      f32x16 s = negl, dp = negd;
      if constexpr (NK <= 8) {
        bfrag kf[NK], vf[NV];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) kf[ks] = row_frag<DK>(ksub, r, ks, hh);
#pragma unroll
        for (int ks = 0; ks < NV; ++ks) vf[ks] = row_frag<DV>(vsub, r, ks, hh);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) s = mfma32(kf[ks], qf[ks], s);
#pragma unroll
        for (int ks = 0; ks < NV; ++ks) dp = mfma32(vf[ks], dof[ks], dp);
      } else {
        // wide keys: the fragments of a sub-tile would not fit in registers together
#pragma unroll
        for (int k0 = 0; k0 < NK; k0 += KCH) {
          bfrag kf[KCH];
#pragma unroll
          for (int ks = 0; ks < KCH; ++ks) kf[ks] = row_frag<DK>(ksub, r, k0 + ks, hh);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < KCH; ++ks) s = mfma32(kf[ks], qf[k0 + ks], s);
        }
        bfrag vf[NV];
#pragma unroll
        for (int ks = 0; ks < NV; ++ks) vf[ks] = row_frag<DV>(vsub, r, ks, hh);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NV; ++ks) dp = mfma32(vf[ks], dof[ks], dp);
      }
      bfrag kt[2][DCH];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < DCH; ++dt) kt[s2][dt] = tr_frag<DK>(ksub, 0, s2, dt, lane);
      __builtin_amdgcn_sched_barrier(0);
      // compare + select only where the diagonal or the end of the keys crosses the sub-tile
      if ((ks0 + 31 > wq0) || (ks0 + 32 > p.Tk)) {
        const int bound = qlim - ks0 - 4 * hh;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (acc_row(i, 0) > bound) s[i] = -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(s[i]);
        float dpi = dp[i];
        // dropped pairs: dP = 0 before delta is taken off, i.e. dS = -p * delta
        if (DROP) dpi = drop_keep(dkey, qrow, ks0 + acc_row(i, hh), dthr) ? (dpi + dlt) * rkeep - dlt : -dlt;
        s[i] = pv * dpi;  // dS^T
      }
      const bfrag dsb0 = acc_frag(s, 0), dsb1 = acc_frag(s, 1);
#pragma unroll
      for (int d0 = 0; d0 < ND; d0 += DCH) {
        if (d0 > 0) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < DCH; ++dt) kt[s2][dt] = tr_frag<DK>(ksub, 0, s2, d0 + dt, lane);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < DCH; ++dt) {
          dqacc[d0 + dt] = mfma32(kt[0][dt], dsb0, dqacc[d0 + dt]);
          dqacc[d0 + dt] = mfma32(kt[1][dt], dsb1, dqacc[d0 + dt]);
        }
      }
    }
  }
  __syncthreads();
  bf16_t* dqp = (bf16_t*)p.dq + b * p.dq_bs + h * DK;
  store_tile_T<DK>(smem + wave * 32 * DK * 2, dqacc, p.scale, dqp, p.dq_ts, q0 + wave * 32, p.Tq, lane, p.rope_cos,
                   p.rope_sin, p.rope_pos0 + (p.Tk - p.Tq));
}

// Timing-only ablations of the dK/dV kernel (benchmarks/ab_build.py ... -DDPA_ATTN_DKV_ABLATE=n, wrong results): 1 = no
// exponentials / dS arithmetic, 2 = no transposing fragment reads (dO^T, Q^T), 3 = no row fragment reads (Q, dO, row
// constants), 4 = no dV / dK products, 5 = no per-tile wait + barrier.
#ifndef DPA_ATTN_DKV_ABLATE
#define DPA_ATTN_DKV_ABLATE 0
#endif
#ifndef DPA_ATTN_DKV_PIN_V
#define DPA_ATTN_DKV_PIN_V 1
#endif
// ------------------------------------------------------------------------------------------
// Backward, dK/dV: one workgroup = 128 keys of one (batch, kv head); each wave keeps dK^T and
// dV^T of its 32 keys in registers and sweeps the query heads of the group x the query tiles.
// ------------------------------------------------------------------------------------------
template <int DK, int DV, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
  constexpr int NK = DK / 16, NV = DV / 16, ND = DK / 32, NDV = DV / 32, QTILE = 64 * DK * 2, DOTILE = 64 * DV * 2;
  constexpr int BUF = QTILE + DOTILE + 512;
  constexpr int KCH = NK > 8 ? 8 : NK, DCH = ND > 4 ? 4 : ND;   // fragment batches (wide keys: several)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const int BHk = p.B * p.nkv;
  int kt, bh;
  tile_order((int)blockIdx.x, (int)gridDim.x, (p.Tk + 127) / 128, BHk, p.group, kt, bh);  // low key tiles (most queries) first
  const int b = bh / p.nkv, kvh = bh % p.nkv;
  const int G = p.nh / p.nkv;
  const int k0 = kt * 128, off = p.Tk - p.Tq;
  const int kw0 = k0 + wave * 32;
  const bf16_t* kp = (const bf16_t*)p.k + b * p.k_bs + kvh * DK;
  const bf16_t* vp = (const bf16_t*)p.v + b * p.v_bs + kvh * DV;
  const float c = p.scale * kLog2e;

  const int key = kw0 + r;
  int q_start = k0 - off;
  if (q_start < 0) q_start = 0;
  q_start = (q_start / 64) * 64;
  const int nqt = (p.Tq - q_start + 63) / 64;
  const int niter = DPA_ATTN_DKV_ABLATE == 6 ? 0 : (nqt > 0 ? nqt : 0) * G;

  TileDma<DK> qd;
  TileDma<DV> dd;
  qd.init(p.q_ts, wave, lane);
  dd.init(p.do_ts, wave, lane);
  const int srows = stat_rows(p.Tq);
  const rsrc_t srs = make_rsrc(p.delta);
  // Q and dO tiles plus the 64 + 64 row constants of iteration `it` -> buffer `buf`
  auto issue = [&](int it, int buf) {
    const int hq = kvh * G + it / nqt, qt0 = q_start + (it % nqt) * 64;
    const rsrc_t qrs = make_rsrc((const bf16_t*)p.q + b * p.q_bs + hq * DK);
    const rsrc_t drs = make_rsrc((const bf16_t*)p.dout + b * p.do_bs + hq * DV);
    const unsigned base = lds0 + buf * BUF;
    qd.issue(qrs, qt0, p.Tq, base);
    dd.issue(drs, qt0, p.Tq, base + QTILE);
    if (wave < 2) {   // wave 0: -lse * log2(e), wave 1: -delta
      const long long row = (long long)(wave == 0 ? p.B * p.nh : 0) * srows + ((long long)b * p.nh + hq) * srows + qt0;
      dma_dwords(base + QTILE + DOTILE + wave * 256, lane * 4, srs, __builtin_amdgcn_readfirstlane((unsigned)(row * 4)));
    }
  };

  // the first tile is on its way before the K / V fragment loads: the two latencies overlap
  if (niter > 0) issue(0, 0);

  bfrag kf[NK], vf[NV];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    kf[ks] = zero_frag();
    if (key < p.Tk) kf[ks] = scaled_frag(kp + (long long)key * p.k_ts + ks * 16 + hh * 8, c);
  }
#pragma unroll
  for (int ks = 0; ks < NV; ++ks) {
    vf[ks] = zero_frag();
    if (key < p.Tk) vf[ks] = *reinterpret_cast<const bfrag*>(vp + (long long)key * p.v_ts + ks * 16 + hh * 8);
  }
  // The V fragments are first read inside the tile loop, and a wait the compiler places there runs in every
  // iteration and also drains the tile prefetch issued just before it (the loads of the prefetch count on the same
  // counter). Reading them here puts that wait in front of the loop.
#if DPA_ATTN_DKV_PIN_V
#pragma unroll
  for (int ks = 0; ks < NV; ++ks) asm volatile("" : "+v"(vf[ks]));
#endif
  f32x16 dkacc[ND], dvacc[NDV];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dkacc[dt][i] = 0.f;
#pragma unroll
  for (int dt = 0; dt < NDV; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dvacc[dt][i] = 0.f;
  const unsigned dthr = DROP ? drop_threshold(p.p_drop) : 0u;
  const float rkeep = DROP ? 1.f / (1.f - p.p_drop) : 1.f;
  // read once here: a load inside the tile loop would have its wait drain the tile prefetch as well
  const long long sv = DROP ? *p.seed : 0ll;
  const unsigned long long seed = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sv >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)sv);

  for (int it = 0; it < niter; ++it) {
    if (DPA_ATTN_DKV_ABLATE != 5 || it == 0) {
      wait_dma();
      __syncthreads();
    }
    if (it + 1 < niter) issue(it + 1, (it + 1) & 1);
    const char* qb = smem + (it & 1) * BUF;
    const char* dob = qb + QTILE;
    const char* stb = qb + QTILE + DOTILE;
    const int qt0 = q_start + (it % nqt) * 64;
    const unsigned dkey = DROP ? drop_key(seed, b * p.nh + kvh * G + it / nqt) : 0u;
    // 32-query sub-tiles with a (query, key) pair visible to the wave: [sub_lo, sub_hi). A rolled loop:
    // with one `if` per unrolled sub-tile the compiler copies all the dK / dV accumulators at every join.
    const int sub_lo = qt0 + 31 + off >= kw0 ? 0 : (qt0 + 63 + off >= kw0 ? 1 : 2);
    const int sub_hi = qt0 + 32 < p.Tq ? 2 : 1;
#pragma unroll 1
    for (int sub = sub_lo; sub < sub_hi; ++sub) {
      const int qs0 = qt0 + sub * 32;
      const char* qsub = qb + sub * (32 * DK * 2);
      const char* dosub = dob + sub * (32 * DV * 2);
      const char* stsub = stb + sub * 128;
      bfrag qf[KCH], dof[NV];
#pragma unroll
      for (int ks = 0; ks < KCH; ++ks) qf[ks] = DPA_ATTN_DKV_ABLATE == 3 ? kf[ks] : row_frag<DK>(qsub, r, ks, hh);
#pragma unroll
      for (int ks = 0; ks < NV; ++ks) dof[ks] = DPA_ATTN_DKV_ABLATE == 3 ? vf[ks] : row_frag<DV>(dosub, r, ks, hh);
      f32x16 s, dp;   // start as the row constants of the 16 queries a lane's accumulators belong to
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int rr = 8 * g4 + 4 * hh;
        const f32x4 l4 = DPA_ATTN_DKV_ABLATE == 3 ? f32x4{-30.f, -30.f, -30.f, -30.f} : *reinterpret_cast<const f32x4*>(stsub + rr * 4);
        const f32x4 d4 = DPA_ATTN_DKV_ABLATE == 3 ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(stsub + 256 + rr * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[4 * g4 + j] = l4[j];
          dp[4 * g4 + j] = d4[j];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 nd;
      if (DROP) nd = dp;
#pragma unroll
      for (int ks = 0; ks < KCH; ++ks) s = mfma32(qf[ks], kf[ks], s);
#pragma unroll
      for (int k0 = KCH; k0 < NK; k0 += KCH) {   // wide keys: the rest of the Q fragments, batch by batch
#pragma unroll
        for (int ks = 0; ks < KCH; ++ks) qf[ks] = row_frag<DK>(qsub, r, k0 + ks, hh);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KCH; ++ks) s = mfma32(qf[ks], kf[k0 + ks], s);
      }
#pragma unroll
      for (int ks = 0; ks < NV; ++ks) dp = mfma32(dof[ks], vf[ks], dp);
      bfrag dot[2][NDV], qt[2][DCH];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int dt = 0; dt < NDV; ++dt) dot[s2][dt] = DPA_ATTN_DKV_ABLATE == 2 ? vf[(2 * s2 + dt) % NV] : tr_frag<DV>(dosub, 0, s2, dt, lane);
#pragma unroll
        for (int dt = 0; dt < DCH; ++dt) qt[s2][dt] = DPA_ATTN_DKV_ABLATE == 2 ? kf[(2 * s2 + dt) % NK] : tr_frag<DK>(qsub, 0, s2, dt, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      // compare + select only on blocks the diagonal crosses
      if (kw0 + 31 > qs0 + off) {
        const int bound = key - qs0 - off - 4 * hh;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (acc_row(i, 0) < bound) s[i] = -INFINITY;   // key > query
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = DPA_ATTN_DKV_ABLATE == 1 ? s[i] : __builtin_amdgcn_exp2f(s[i]);
        float pd = pv, dpi = dp[i];
        if (DROP) {   // dropped pairs: P = 0 for dV, dP = 0 before delta is taken off
          const bool keep = drop_keep(dkey, qs0 + acc_row(i, hh), key, dthr);
          pd = keep ? pv * rkeep : 0.f;
          dpi = keep ? (dpi - nd[i]) * rkeep + nd[i] : nd[i];
        }
        s[i] = pd;          // P (dropped and rescaled: the weights dV sees)
        dp[i] = DPA_ATTN_DKV_ABLATE == 1 ? dpi : pv * dpi;   // dS
      }
      if constexpr (ND == NDV && ND <= 4) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bfrag pb = acc_frag(s, s2);
          const bfrag dsb = acc_frag(dp, s2);
#pragma unroll
          for (int dt = 0; dt < ND; ++dt) {
            if (DPA_ATTN_DKV_ABLATE == 4) {
              asm volatile("" : "+v"(dvacc[dt]), "+v"(dkacc[dt]) : "v"(dot[s2][dt]), "v"(pb), "v"(qt[s2][dt]), "v"(dsb));
              continue;
            }
            dvacc[dt] = mfma32(dot[s2][dt], pb, dvacc[dt]);
            dkacc[dt] = mfma32(qt[s2][dt], dsb, dkacc[dt]);
          }
        }
      } else {
        const bfrag pb0 = acc_frag(s, 0), pb1 = acc_frag(s, 1), dsb0 = acc_frag(dp, 0), dsb1 = acc_frag(dp, 1);
#pragma unroll
        for (int dt = 0; dt < NDV; ++dt) {
          dvacc[dt] = mfma32(dot[0][dt], pb0, dvacc[dt]);
          dvacc[dt] = mfma32(dot[1][dt], pb1, dvacc[dt]);
        }
#pragma unroll
        for (int d0 = 0; d0 < ND; d0 += DCH) {
          if (d0 > 0) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int dt = 0; dt < DCH; ++dt) qt[s2][dt] = tr_frag<DK>(qsub, 0, s2, d0 + dt, lane);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int dt = 0; dt < DCH; ++dt) {
            dkacc[d0 + dt] = mfma32(qt[0][dt], dsb0, dkacc[d0 + dt]);
            dkacc[d0 + dt] = mfma32(qt[1][dt], dsb1, dkacc[d0 + dt]);
          }
        }
      }
    }
  }
  __syncthreads();
  bf16_t* dkp = (bf16_t*)p.dk + b * p.dk_bs + kvh * DK;
  bf16_t* dvp = (bf16_t*)p.dv + b * p.dv_bs + kvh * DV;
  char* wl = smem + wave * 32 * DK * 2;
  store_tile_T<DK>(wl, dkacc, p.scale, dkp, p.dk_ts, kw0, p.Tk, lane, p.rope_cos, p.rope_sin, p.rope_pos0);
  store_tile_T<DV>(wl, dvacc, 1.f, dvp, p.dv_ts, kw0, p.Tk, lane);
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------
template <int DK, int DV, bool DROP>
static void fwd_launch(const AttnParams& p, hipStream_t st) {
  const int nqt = (p.Tq + 127) / 128;
  size_t lds = (size_t)fwd_stages(DK, DV) * (64 * DK * 2 + 64 * DV * 2);
  // benchmark aid: unused LDS bytes added to the request, which bounds the workgroups per CU (160 KiB each)
  static const long pad = getenv("DPA_ATTN_FWD_PAD_LDS") ? atol(getenv("DPA_ATTN_FWD_PAD_LDS")) : 0;
  if (pad > 0 && lds + (size_t)pad <= 160 * 1024) lds += (size_t)pad;
  static bool attr_set = false;
  if (!attr_set && lds > 65536) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<DK, DV, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<DK, DV, DROP>), dim3(nqt * p.B * p.nh), dim3(256), lds, st, p);
}

template <int DK, int DV, bool DROP>
static void bwd_launch(const AttnParams& p, hipStream_t st) {
  // dQ first: it also produces delta, which the dK/dV kernel consumes.
  const int nqt = (p.Tq + 127) / 128;
  const size_t lds_q = 2 * (64 * DK * 2 + 64 * DV * 2);
  const int nkt = (p.Tk + 127) / 128;
  const size_t lds_kv = 2 * (64 * DK * 2 + 64 * DV * 2 + 512);
  static bool attr_set = false;
  if (!attr_set) {
    if (lds_q > 65536)
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DK, DV, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_q);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DK, DV, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_kv);
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DK, DV, DROP>), dim3(nqt * p.B * p.nh), dim3(256), lds_q, st, p);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DK, DV, DROP>), dim3(nkt * p.B * p.nkv), dim3(256), lds_kv, st, p);
}

// Heads per XCD-local group. Measured on MI355X (benchmarks/attn_bench.py): groups of 4 heads gain
// 3-5 % in the backward kernels and up to 9 % with grouped-query heads at T <= 2048; with more
// than 16 tiles per head the longer tail of a group outweighs the L2 reuse. DPA_ATTN_GROUP overrides.
static int tile_group(int tiles_per_head) {
  static const int g = [] { const char* e = getenv("DPA_ATTN_GROUP"); return e ? atoi(e) : -1; }();
  if (g >= 0) return g;
  return tiles_per_head <= 16 ? 4 : 0;
}

#ifdef DPA_ATTN_WIDE
// Key width != value width (multi-head latent attention with decoupled rotary keys: the key carries the rotary
// columns behind the head's own, the value does not): (128, 64) and (256, 128). This translation unit is built
// without the accumulators-in-VGPRs option of the square kernels: at key width 256 the dQ / dK accumulators alone
// take 128 registers, and with them in AGPRs the operand fragments still fit the 256 architectural ones.
bool attn_wide_ok(int dk, int dv) { return (dk == 128 && dv == 64) || (dk == 256 && dv == 128); }

void attn_fwd_wide(const AttnParams& p0, hipStream_t st) {
  AttnParams p = p0;
  p.group = tile_group((p.Tq + 127) / 128);
  const bool drop = p.p_drop > 0.f && p.seed != nullptr;
  if (p.hs == 128) drop ? fwd_launch<128, 64, true>(p, st) : fwd_launch<128, 64, false>(p, st);
  else drop ? fwd_launch<256, 128, true>(p, st) : fwd_launch<256, 128, false>(p, st);
}

void attn_bwd_wide(const AttnParams& p0, hipStream_t st) {
  AttnParams p = p0;
  p.group = tile_group((p.Tq + 127) / 128);
  const bool drop = p.p_drop > 0.f && p.seed != nullptr;
  if (p.hs == 128) drop ? bwd_launch<128, 64, true>(p, st) : bwd_launch<128, 64, false>(p, st);
  else drop ? bwd_launch<256, 128, true>(p, st) : bwd_launch<256, 128, false>(p, st);
}
#else
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(unsigned char* __restrict__ keep, int nh, int Tq,
                                                                int Tk, float p_drop,
                                                                const long long* __restrict__ seed, long long total) {
  const unsigned thr = drop_threshold(p_drop);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Tk);
    const long long rest = i / Tk;
    const int q = (int)(rest % Tq), bh = (int)(rest / Tq);
    keep[i] = drop_keep(drop_key(seed, bh), q, k, thr) ? 1 : 0;
  }
}

void attn_dropout_mask(unsigned char* keep, int B, int nh, int Tq, int Tk, float p_drop, const long long* seed,
                       hipStream_t st) {
  const long long total = (long long)B * nh * Tq * Tk;
  if (total == 0) return;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, keep, nh, Tq, Tk, p_drop, seed,
                     total);
}

void attn_fwd(const AttnParams& p0, hipStream_t st) {
  if (p0.B == 0 || p0.Tq == 0) return;
  if (p0.vw != p0.hs) return attn_fwd_wide(p0, st);
  AttnParams p = p0;
  p.group = tile_group((p.Tq + 127) / 128);
  const bool drop = p.p_drop > 0.f && p.seed != nullptr;
  if (p.hs == 64) drop ? fwd_launch<64, 64, true>(p, st) : fwd_launch<64, 64, false>(p, st);
  else drop ? fwd_launch<128, 128, true>(p, st) : fwd_launch<128, 128, false>(p, st);
}

void attn_bwd(const AttnParams& p0, hipStream_t st) {
  if (p0.B == 0 || p0.Tq == 0) return;
  if (p0.vw != p0.hs) return attn_bwd_wide(p0, st);
  AttnParams p = p0;
  p.group = tile_group((p.Tq + 127) / 128);
  const bool drop = p.p_drop > 0.f && p.seed != nullptr;
  if (p.hs == 64) drop ? bwd_launch<64, 64, true>(p, st) : bwd_launch<64, 64, false>(p, st);
  else drop ? bwd_launch<128, 128, true>(p, st) : bwd_launch<128, 128, false>(p, st);
}
#endif

}  // namespace dpa
