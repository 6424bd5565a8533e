// Host-callable launchers of the gfx950 kernels. Torch-free on purpose: the kernel sources
// compile in seconds and the single bindings translation unit is the only one that sees
// PyTorch headers. All launchers are asynchronous on `st` and perform no allocation.
#pragma once
#include <hip/hip_runtime.h>

namespace dpa {

// ---- LayerNorm -------------------------------------------------------------------------
int layernorm_max_cols();
int layernorm_bwd_parts(int rows, int C);
void layernorm_fwd(const void* x, const void* res, const void* w, const void* b, void* y, void* sum_out,
                   float* mean, float* rstd, int rows, int C, float eps, bool is_bf16, hipStream_t st);
void layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                   const void* dres, void* dx, void* dw, void* db, float* scratch, int nparts, int rows, int C,
                   bool accumulate, bool is_bf16, hipStream_t st);

// ---- Optimizer -------------------------------------------------------------------------
struct AdamSegs {
  int nseg;
  long long end[64];  // exclusive end offset (elements, multiple of 8) of each weight-decay segment
  float wd[64];
};
void adamw_step(float* master, float* m, float* v, const void* grad, bool grad_bf16, void* param_out,
                bool param_bf16, long long n, const AdamSegs& segs, float lr, float beta1, float beta2, float eps,
                float bc1, float bc2_sqrt, const float* norm2, float max_norm, float grad_mul, const float* dyn,
                hipStream_t st);  // dyn: optional device {lr, bc1, bc2_sqrt} overriding the arguments
int sumsq_parts(long long n);
void sumsq(const void* g, bool is_bf16, long long n, float* partials, float* out, bool accumulate, hipStream_t st);

// ---- Cross-entropy ---------------------------------------------------------------------
void xent(int mode, void* logits, bool is_bf16, const long long* targets, float* loss_rows, float* row_max,
          float* row_sum, float* row_tgt, const float* lse_in, const float* grad_scale, int rows, int V, int ld,
          int vocab_start, long long ignore_index, bool write_grad, hipStream_t st);

// ---- Embedding -------------------------------------------------------------------------
void embed_fwd(const long long* idx, const void* wte, const void* pos, void* out, long long N, int T_len, int pos0,
               int C, long long V, bool is_bf16, hipStream_t st);
// k_cache / v_cache [B, S, W] (bf16) row *pos <- kn / vn [B, W] (batch strides kn_bs / vn_bs, 16-byte aligned rows).
void kv_append(void* kc, void* vc, const void* kn, const void* vn, const long long* pos, int B, int S, int W,
               long long kn_bs, long long vn_bs, hipStream_t st);
int embed_bwd_chunks(long long N);  // scratch = 2 * chunks * C floats
void embed_bwd(const void* dout, const long long* ids_sorted, const long long* perm, void* dwte, float* scratch,
               long long N, int C, long long V, bool is_bf16, hipStream_t st);
void pos_bwd(const void* dout, void* dpos, int B, int T_len, int pos0, int C, bool is_bf16, hipStream_t st);

// ---- Activations -----------------------------------------------------------------------
void gelu_fwd(const void* x, void* y, long long n, bool is_bf16, hipStream_t st);
void gelu_bwd(const void* dy, const void* x, void* dx, long long n, bool is_bf16, hipStream_t st);
// Other activations: kind 0..act_kinds()-1 = relu, silu, mish, selu, elu, sigmoid, tanh, leaky relu (0.01).
// dy == nullptr: out = f(x); otherwise out = dy * f'(x).
int act_kinds();
void act_apply(int kind, const void* dy, const void* x, void* out, long long n, bool is_bf16, hipStream_t st);
void swiglu_fwd(const void* h, void* y, long long M, int U, bool is_bf16, hipStream_t st);
void swiglu_bwd(const void* dy, const void* h, void* dh, long long M, int U, bool is_bf16, hipStream_t st);

// ---- Attention (bf16, head size 64 or 128, causal, GQA-native) ---------------------------
// Element (b, t, h, d) of a tensor X lives at X + b*X_bs + t*X_ts + h*W + d (W = the tensor's head width), which lets the
// kernels read q/k/v straight out of a fused QKV projection and write dq/dk/dv straight into
// the fused gradient buffer.
struct AttnParams {
  const void *q, *k, *v;
  void* o;
  float* lse;          // [B, nh, Tq] natural-log log-sum-exp of the scaled scores
  const void* dout;    // backward only
  float* delta;        // backward scratch, [2][B * nh][Tq rounded up to 128]: -rowsum(dO * O), -lse * log2(e)
  void *dq, *dk, *dv;
  int B, nh, nkv, Tq, Tk, hs;   // hs: width of q and k
  int vw;                       // width of v and o (== hs, or half of it: see attn_wide_ok)
  long long q_bs, k_bs, v_bs, o_bs, do_bs, dq_bs, dk_bs, dv_bs;
  int q_ts, k_ts, v_ts, o_ts, do_ts, dq_ts, dk_ts, dv_ts;
  float scale;
  int group;           // tile order: 0 = tile-major over all heads; G > 0 = XCD-local groups of G heads
  // Attention dropout (p_drop > 0): the keep mask is a counter-based hash of (seed, batch*head, query,
  // key), evaluated in registers by all three kernels; the seed is read from device memory.
  float p_drop;
  const long long* seed;
  // Forward only: number of valid keys read from device memory (<= Tk, the allocated length). Lets a
  // captured decode step run over a preallocated KV cache whose fill level changes between replays.
  const int* kv_len;
  // Backward only, optional: q and k were rotated (interleaved-pair rotary embedding, fp32 tables
  // [positions, hs / 2]) after the projection that the gradients go back to, row t by the angles of position
  // rope_pos0 + t: dq and dk are rotated back as they are stored.
  const float *rope_cos, *rope_sin;
  int rope_pos0;
};
void attn_fwd(const AttnParams& p, hipStream_t st);
void attn_bwd(const AttnParams& p, hipStream_t st);
// key width != value width (csrc/attention_wide.hip): (128, 64) and (256, 128)
bool attn_wide_ok(int dk, int dv);
void attn_fwd_wide(const AttnParams& p, hipStream_t st);
void attn_bwd_wide(const AttnParams& p, hipStream_t st);
// keep[b, h, q, k] (1 byte each) of the dropout mask the attention kernels apply for this seed.
void attn_dropout_mask(unsigned char* keep, int B, int nh, int Tq, int Tk, float p_drop, const long long* seed,
                       hipStream_t st);

// Single-query attention over a KV cache (csrc/attn_decode.hip): q, o [B, 1, nh, hs] (batch strides q_bs / o_bs,
// heads packed), k, v [B, S, nkv, hs] (strides *_bs, *_ts), *kv_len valid keys (clamped to S), hs 64 or 128.
struct AttnDecodeArgs {
  const void *q, *k, *v;
  void* o;
  long long q_bs, o_bs, k_bs, v_bs;
  int k_ts, v_ts;
  int B, nh, nkv, hs, S;
  float scale;
  const int* kv_len;
};
void attn_decode(const AttnDecodeArgs& p, hipStream_t st);

// ---- Weight-gradient GEMM -----------------------------------------------------------------
// ws[N, K] (fp32) += dy[M, N]^T * x[M, K]; dbias[N] (fp32, optional) += column sums of dy.
// N and K must be multiples of 128; rows 16-byte aligned. With splits > 1 the token range is
// divided over workgroups that combine through float atomics.
int wgrad_splits(int M, int N, int K);
void wgrad_gemm(const void* dy, const void* x, float* ws, float* dbias, int M, int N, int K, long long ldy,
                long long ldx, int splits, hipStream_t st);
// LDS-DMA version (K % 256 == 0, M % 64 == 0). splits > 0: `splits` fp32 slabs are written;
// splits == 0: one pass that adds into ws[N*K].
bool wgrad2_ok(int M, int N, int K);
int wgrad2_splits(int M, int N, int K);
// x (n elements, n % 8 == 0, 16-byte aligned) *= g[0] / expected[0] unless the two are equal
void scale_unless(void* x, long long n, bool is_bf16, const float* g, const float* expected, hipStream_t st);
void wgrad2_set_stages(int s);
void wgrad2_force_splits(int s);
// bias_slabs (optional, only when wgrad2_fuses_bias(M, N, K) and splits > 0): fp32 [wgrad2_bias_slabs(M, N, K), N]
// receiving partial column sums of dy (their sum is the bias gradient), produced inside the same kernel.
bool wgrad2_fuses_bias(int M, int N, int K);
int wgrad2_bias_slabs(int M, int N, int K);   // rows of bias_slabs (all written; their sum is the bias gradient)
// Per-segment weight gradients on the 256 x 256 kernel (segment bounds in device memory, multiples of 64 rows),
// work divided in equal row chunks across the segments: out = fp32 [nseg, N, K], zero-initialised by the caller,
// receives the partial tiles through float atomics. N % 256 == 0, K % 256 == 0.
bool wgrad3_grouped_ok(long long rows, int N, int K);
void wgrad3_grouped(const void* dy, const void* x, float* out, const long long* seg_start, const long long* seg_rows,
                    int nseg, int rows, int N, int K, long long ldy, long long ldx, hipStream_t st);
void wgrad2_gemm(const void* dy, const void* x, float* ws, int M, int N, int K, long long ldy, long long ldx,
                 int splits, hipStream_t st, float* bias_slabs = nullptr, void* direct = nullptr,
                 bool direct_bf16 = false);
// wgrad2_direct_ok: the product runs as one split of the 256 x 256 kernel, which then takes `direct` (bf16 or fp32
// [N, K], added to in place) instead of writing a slab to ws
bool wgrad2_direct_ok(int M, int N, int K);
void wgrad_finish(const float* ws, void* out, long long n, int nslabs, bool accumulate, bool out_bf16,
                  hipStream_t st);
// out_g (+)= ws[g] for `count` destinations of n elements each, given by device addresses.
void wgrad_finish_many(const float* ws, const long long* dst_ptrs, int count, long long n, bool accumulate,
                       bool out_bf16, hipStream_t st);

// ---- NT GEMM with fused epilogues (csrc/gemm_nt.hip) ------------------------------------------------
// c[M, N] = epi(a[M, K] * b[N, K]^T [+ bias[N]]), bf16 operands and results, fp32 accumulation.
//   kEpiNone : c = v
//   kEpiGelu : c = GELU_erf(v), aux_out = GELU_erf'(v)      (aux_out [M, N], row stride ld_aux)
//   kEpiMul  : c = v * aux_in                                (aux_in  [M, N], row stride ld_aux)
//   kEpiAdd  : c = v + aux_in
//   kEpiSwiglu  : b is [2N, K] (gate rows 0..N-1, up rows N..2N-1): c[M, N] = silu(a gate^T) * (a up^T), and
//                 aux_out[M, 2N] (row stride ld_aux) receives the pre-activation [gate | up] for the backward
//                 pass. An output tile pairs 128 gate with the 128 matching up columns inside each lane.
//   kEpiDswiglu : c[M, 2N] (row stride ldc) = [v * up * silu'(gate) | v * silu(gate)] with [gate | up] =
//                 aux_in[M, 2N] (row stride ld_aux): the input gradient of the SwiGLU from v = dy W_proj.
//   kEpiRope : c = v with the interleaved pairs (2j, 2j + 1) of every rope_hs-wide head in columns
//                 [0, rope_cols) rotated by the angles of position rope_pos0 + row % rope_T (rotary embedding of
//                 the q and k heads of a fused qkv projection; tables rope_cos / rope_sin fp32 [positions, rope_hs / 2])
// K % 64 == 0, K >= 128, N % 16 == 0, rows 16-byte aligned, every tensor below 4 GB - 16 MB,
// bias only for N <= 8192 (gemm_nt_ok checks all of it).
enum { kEpiNone = 0, kEpiGelu = 1, kEpiMul = 2, kEpiAdd = 3, kEpiSwiglu = 4, kEpiDswiglu = 5, kEpiRope = 6 };
struct GemmNtArgs {
  const void *a, *b;
  void* c;
  const void* bias;     // [N] bf16 or nullptr
  void* aux_out;
  const void* aux_in;
  int M, N, K;
  long long lda, ldb, ldc, ld_aux;   // row strides in elements
  unsigned a_bytes, b_bytes, c_bytes, aux_bytes;   // filled by gemm_nt()
  int tiles_m, tiles_n;                            // filled by gemm_nt()
  int skew_ticks;                                  // filled by gemm_nt() (benchmark knob)
  // Grouped mode (MoE experts): b is ignored; every 256-row block of a uses the weight matrix
  // w_ptrs[block_group[row / 256]] ([N, K], row stride ldb, all of one shape); a block whose group is outside
  // [0, ngroups) multiplies by zeros. M must be a multiple of 256. Epilogues none / GELU / multiply / SwiGLU / dSwiGLU.
  const long long* w_ptrs;
  const int* block_group;
  int ngroups;
  // kEpiRope only: rope_hs in {64, 128}, rope_cols % rope_hs == 0, rope_T > 0, tables of at least rope_pos0 + rope_T rows
  const float *rope_cos, *rope_sin;
  int rope_cols, rope_hs, rope_T, rope_pos0;
};
bool gemm_nt_ok(long long M, long long N, long long K, long long lda, long long ldb, long long ldc, long long ld_aux,
                bool has_bias);
// the two SwiGLU forms: N = number of gate (= up) columns; ld_act the row stride of the [M, N] tensor, ld_pair that
// of the [M, 2N] one
bool gemm_nt_pair_ok(long long M, long long N, long long K, long long lda, long long ldb, long long ld_act,
                     long long ld_pair);
void gemm_nt(GemmNtArgs a, int epi, int num_cus, hipStream_t st);
// Rows one launch covers (the kernel's byte offsets stay below 2 GiB; gemm_nt() walks taller problems in row
// ranges of this many rows). 0: not even one aligned range fits; callers refuse such operands.
long long gemm_nt_rows_per_launch(const GemmNtArgs& a, int epi);
void gemm_nt_set_variant(int v);   // benchmark knob, see csrc/gemm_nt.hip
// out[C, R] = in[R, C]^T (bf16, R % 8 == 0 and C % 8 == 0, contiguous).
void transpose_bf16(const void* in, void* out, int R, int C, hipStream_t st);
// count equally shaped matrices (device arrays of addresses) in one launch; count <= 65535
void transpose_bf16_many(const long long* in_ptrs, const long long* out_ptrs, int count, int R, int C, hipStream_t st);

// ---- Decode-step linear layer (csrc/decode.hip) ------------------------------------------------------
// y[B, N] = act(LN(x)[B, K] * w[N, K]^T + bias) + residual for B <= 64 rows (bf16, fp32 accumulation), one launch.
//   ln_w == nullptr: no LayerNorm (ln_b may be nullptr with ln_w set: no shift)
//   act kDecodeSwiglu: w is [2N, K] (gate rows, then up rows), bias [2N]: y = silu(gate) * up
//   k_cache != nullptr: output columns [kv_col0, kv_col0 + kv_cols) go to row *pos of k_cache [B, S, kv_cols]
//   (batch stride cache_bs elements), the next kv_cols columns to v_cache; columns below kv_col0 to y.
// N % 16 == 0, K % 128 == 0 (K <= 2048 with the LayerNorm), kv_col0 % 16 == 0, kv_cols % 16 == 0 (decode_linear_ok
// checks the sizes).
enum { kDecodeNone = 0, kDecodeGelu = 1, kDecodeSwiglu = 2 };
constexpr int kDecodeMaxRows = 64;   // rows of one call: groups of 16 share a launch (and the weights, through L2); from ~128 rows on the library GEMMs are faster
struct DecodeLinearArgs {
  const void* x;
  long long ldx;
  const void *ln_w, *ln_b;
  float eps;
  const void* w;
  long long ldw;
  const void* bias;
  const void* residual;
  long long ldr;
  void* y;
  long long ldy;
  int B, N, K, act;
  void *k_cache, *v_cache;
  long long cache_bs;
  const long long* pos;
  int kv_col0, kv_cols;
  // rotary embedding of the output columns below rope_cols (heads of rope_hs columns, interleaved pairs), applied to
  // the fp32 result before rounding: rope_cos / rope_sin = the fp32 [rope_hs / 2] rows of the token's position
  // (nullptr: none). act must be kDecodeNone, rope_cols % rope_hs == 0.
  const float *rope_cos, *rope_sin;
  int rope_cols, rope_hs;
};
bool decode_linear_ok(long long B, long long N, long long K, long long ldx, long long ldw, bool pair, bool has_ln);
void decode_linear(const DecodeLinearArgs& p, hipStream_t st);

// ---- Grouped GEMM (MoE experts) ----------------------------------------------------------------
// c[rows, N] = a[rows, K] * W_g, one weight matrix per 128-row block (block_group[rb]; values outside
// [0, ngroups) skip the block). w_is_kn = false: W_g is [N, K]; true: W_g is [K, N]. K % 64 == 0,
// N % 8 == 0, all rows 16-byte aligned.
void gemm_grouped(const void* a, long long lda, const long long* w_ptrs, long long ldw, bool w_is_kn, void* c,
                  long long ldc, const int* block_group, int ngroups, int rows, int N, int K, hipStream_t st);
// out[g, N, K] (fp32, fully overwritten) = a[seg_g, N]^T * b[seg_g, K]; segment bounds (multiples of 64 rows)
// are read from device memory. K % 128 == 0, N % 8 == 0.
void gemm_grouped_wgrad(const void* a, long long lda, const void* b, long long ldb, float* out,
                        const long long* seg_start, const long long* seg_rows, int ngroups, int N, int K,
                        int max_rows, hipStream_t st);
// MoE dispatch / combine row movement (bf16, C % 8 == 0).
void moe_scatter_rows(const void* src, const long long* row_of_pair, void* dst, long long npairs, int k, int C,
                      hipStream_t st);
void moe_gather_combine(const void* y, const long long* row_of_pair, const void* gates, void* out, long long ntok,
                        int k, int C, hipStream_t st);
// zero the pair-less rows of a padded expert-sorted bf16 buffer [rows_total, C] (grid: (E + 1) x block workgroups)
void moe_zero_padding(void* dst, const long long* seg_start, const long long* counts, const long long* seg_rows, int E,
                      int C, long long rows_total, int block, hipStream_t st);
void moe_combine_bwd(const void* dout, const void* y, const long long* row_of_pair, const void* gates, void* dy,
                     float* dgate, long long npairs, int k, int C, hipStream_t st);

// ---- MoE routing ---------------------------------------------------------------------------------
// idx[t, j] / gates[t, j]: top-k of (logits + bias) per token (E <= 64, k <= 8) and the softmax of the
// selected raw logits.
void moe_topk_gate(const float* logits, const float* bias, long long* idx, float* gates, long long ntok, int E, int k,
                   hipStream_t st);
// Expert-sorted layout of the [npairs] expert ids (see csrc/moe_route.hip). pad = 128: every expert's
// segment is padded to whole 128-row blocks and block_group[nrb] is filled; pad = 1: dense (the send
// order of an all-to-all), block_group untouched. hist is scratch of moe_plan_blocks(npairs) * E ints.
int moe_plan_blocks(long long npairs);
void moe_plan(const long long* idx, long long npairs, int E, int nrb, int pad, int* hist, long long* row_of_pair,
              int* block_group, long long* seg_start, long long* seg_rows, long long* counts, hipStream_t st);

// ---- Column sums (bias gradients) ----------------------------------------------------------
// out[N] (fp32, caller-zeroed) += column sums of x[M, N] (row stride ld elements, N % 8 == 0).
void colsum(const void* x, float* out, long long M, int N, long long ld, bool is_bf16, hipStream_t st);
// out[row] = one token drawn from softmax(logits[row] / temperature) restricted to the top-k logits
// (topk <= 0: no restriction; temperature <= 0: 1). The uniform variate comes from a hash of *seed and the row.
// V % 8 == 0, V <= 57344 (sample_topk_ok), rows 16-byte aligned.
bool sample_topk_ok(int V);
// Optional bookkeeping of a captured decode step (all pointers to device memory, all null: plain sampling):
// history[row, *history_pos] = token; then, once every row is done, *pos += 1, *len_after += 1, *history_pos += 1.
// *pos is also mixed into the random stream, so one seed serves a whole generation. ticket: a zeroed counter.
struct SampleStep {
  long long* history = nullptr;
  long long history_ld = 0;
  long long* history_pos = nullptr;
  long long* pos = nullptr;
  int* len_after = nullptr;
  unsigned* ticket = nullptr;
};
void sample_topk(const void* logits, long long ld, int rows, int V, float temperature, int topk, const long long* seed,
                 long long* out, bool is_bf16, hipStream_t st, SampleStep step = SampleStep{});
// Benchmark aid: occupy `blocks` CUs for about `usec` microseconds.
void spin(int blocks, int usec, float* sink, hipStream_t st);

// ---- RoPE -------------------------------------------------------------------------------
// In-place interleaved-pair rotation of `nheads` heads of size hs at x + t*ts (+ b*bs), using
// fp32 cos/sin tables [T_total, hs/2] starting at row pos0. `inverse` applies the transpose
// rotation (the backward of the forward rotation).
void rope_inplace(void* x, long long bs, int ts, int B, int T_len, int nheads, int hs, const float* cos_t,
                  const float* sin_t, int pos0, bool inverse, bool is_bf16, hipStream_t st);

}  // namespace dpa
