// Decode-step linear layer for gfx950: one launch computes
//
//     y[B, N] = act( LN(x)[B, K] * W[N, K]^T + bias ) + residual          (B <= 64 rows, bf16, fp32 accumulate)
//
// A decode step of a transformer block is a chain of matrix-vector-sized products, each preceded or
// followed by a LayerNorm, a bias, an activation or a residual add. As separate launches (LayerNorm,
// library GEMM with 16 x 16 tiles, activation, add) a 124M-parameter model takes ~10 launches of 4-12 us
// per layer; the weights of a whole layer stream from HBM in 3 us. Here the LayerNorm is the prologue
// (every workgroup normalises the <= 16 rows itself: 12 KB from L2), bias / exact-erf GELU / SwiGLU /
// residual (and the rotary embedding of a qkv projection's query / key heads) are the epilogue, and the key / value columns of a fused qkv projection can be written
// straight into the preallocated KV cache at the device-side position, so a dense block is five launches.
//
// Structure: one 256-thread workgroup per 16 output features and group of 16 rows (blockIdx.y; the groups of a
// larger batch read the same weights, from L2 after the first). The activations sit in LDS as bf16 (rows
// padded by 16 bytes; only the B real rows, so 8 rows of width 8192 still fit). v_mfma_f32_16x16x32_bf16 with the WEIGHT rows as the row operand:
// a lane's 16-byte global load of W[n0 + lane % 16][k0 + 8 (lane / 16) ..] is exactly its fragment, no LDS
// and no cross-lane reduction for the weights; the activation fragment is one ds_read_b128. The four waves
// split K, their partial 16 x 16 tiles are added through LDS by wave 0, which applies the epilogue and
// stores 4 consecutive features per lane.
//
// Replaces, for cached decoding, the per-token nn.Linear / LayerNorm / activation calls of the reference's
// generate loop (reference: single-gpu/model.py:508-533,699-747).
#include "common.h"
#include "kernels.h"

namespace dpa {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 dfrag8;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// Sum over the wave: four rotations inside each row of 16 lanes (DPP, no LDS round trip), then the two
// exchanges between rows.
__device__ __forceinline__ float wave_sum_fast(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

constexpr int kLnVec = 4;   // 16-byte pieces of a row per lane in the LayerNorm prologue: K <= 2048

// The kernel is a chain of dependent memory round trips (activations -> statistics -> LDS -> products ->
// partial sums -> store), each about as long as the whole transfer of the workgroup's weights, so everything
// that does not depend on the activations is requested first: the first 8 weight fragments of the wave
// (all of them for K <= 1024), the bias and the residual of the lane's outputs.
// NW waves split K (4, or 16 for a long K without LayerNorm: K = 3072 is then one batch of 6 fragments per
// lane instead of three batches of 8, and the 48 workgroups of a 768-wide output each keep 98 KB in flight).
template <int ACT, int NW, bool LN>
__global__ __launch_bounds__(NW * 64) void decode_linear_kernel(DecodeLinearArgs p) {
  constexpr bool kPair = ACT == kDecodeSwiglu;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K;
  const int row0 = blockIdx.y * 16;                      // this workgroup's rows: row0 .. row0 + B - 1
  const int B = p.B - row0 < 16 ? p.B - row0 : 16;
  const int xstride = K * 2 + 16;                        // bytes per staged row
  float* part = reinterpret_cast<float*>(smem + B * xstride);    // [pair][wave][lane][4]
  const bf16_t* x = reinterpret_cast<const bf16_t*>(p.x) + (long long)row0 * p.ldx;

  // ---- weight fragments of this wave's quarter of K (pairs: 16 gate + the 16 matching up rows) ----
  const int n0 = blockIdx.x * 16;
  const int l15 = lane & 15, g = lane >> 4;
  const int kq = K / NW, kbeg = wave * kq, nf = kq >> 5;
  const bf16_t* w = reinterpret_cast<const bf16_t*>(p.w);
  const bf16_t* wrow = w + (long long)(n0 + l15) * p.ldw + kbeg + 8 * g;
  const bf16_t* wrow2 = wrow + (long long)p.N * p.ldw;   // up rows (pairs only)
  dfrag8 a[8] = {}, a2[8] = {};
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (u < nf) {
      a[u] = *reinterpret_cast<const dfrag8*>(wrow + 32 * u);
      if (kPair) a2[u] = *reinterpret_cast<const dfrag8*>(wrow2 + 32 * u);
    }
  // ---- bias and residual of the outputs wave 0 will write: features n0 + 4 g .. + 3 of row l15 ----
  const int n = n0 + 4 * g;
  u32x2 bw = {0u, 0u}, bw2 = {0u, 0u}, rr = {0u, 0u};
  if (wave == 0) {
    if (p.bias != nullptr) {
      bw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.bias) + n);
      if (kPair) bw2 = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.bias) + p.N + n);
    }
    if (p.residual != nullptr && l15 < B)
      rr = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.residual) + (long long)(row0 + l15) * p.ldr + n);
  }
  // rotary pairs (n, n + 1) and (n + 2, n + 3) of a head: angles (n % hs) / 2 and the next
  const bool rot = ACT == kDecodeNone && p.rope_cos != nullptr && n < p.rope_cols;
  f32x2 rc = {1.f, 1.f}, rs = {0.f, 0.f};
  if (wave == 0 && rot) {
    const int pi = (n % p.rope_hs) >> 1;
    rc = *reinterpret_cast<const f32x2*>(p.rope_cos + pi);
    rs = *reinterpret_cast<const f32x2*>(p.rope_sin + pi);
  }

  // ---- stage the activations: LayerNorm (fp32 statistics, two passes over registers) or a copy (rows past B are not staged) ----
  if constexpr (!LN) {
    for (int b = wave; b < B; b += NW) {
      char* row = smem + b * xstride;
      const bf16_t* xr = x + (long long)b * p.ldx;
      for (int k = lane * 8; k < K; k += 512) *reinterpret_cast<u32x4*>(row + k * 2) = *reinterpret_cast<const u32x4*>(xr + k);
    }
  } else {
    // the wave's rows b = wave + 4 i together: their loads and their reductions overlap
    u32x4 xv[4][kLnVec];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < kLnVec; ++j) {
        const int b = wave + 4 * i, k = lane * 8 + 512 * j;
        xv[i][j] = (b < B && k < K) ? *reinterpret_cast<const u32x4*>(x + (long long)b * p.ldx + k) : u32x4{0u, 0u, 0u, 0u};
      }
    u32x4 gv[kLnVec], bv[kLnVec];
    const bf16_t* gw = reinterpret_cast<const bf16_t*>(p.ln_w);
    const bf16_t* gb = reinterpret_cast<const bf16_t*>(p.ln_b);
#pragma unroll
    for (int j = 0; j < kLnVec; ++j) {
      const int k = lane * 8 + 512 * j;
      gv[j] = k < K ? *reinterpret_cast<const u32x4*>(gw + k) : u32x4{0u, 0u, 0u, 0u};
      bv[j] = (gb != nullptr && k < K) ? *reinterpret_cast<const u32x4*>(gb + k) : u32x4{0u, 0u, 0u, 0u};
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < kLnVec; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += lo_bf16(xv[i][j][e]) + hi_bf16(xv[i][j][e]);
      mean[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) mean[i] = wave_sum_fast(mean[i]) / (float)K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < kLnVec; ++j) {
        if (lane * 8 + 512 * j < K) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d0 = lo_bf16(xv[i][j][e]) - mean[i], d1 = hi_bf16(xv[i][j][e]) - mean[i];
            s2 += d0 * d0 + d1 * d1;
          }
        }
      }
      rstd[i] = s2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rstd[i] = rsqrtf(wave_sum_fast(rstd[i]) / (float)K + p.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = wave + 4 * i;
      char* row = smem + b * xstride;
#pragma unroll
      for (int j = 0; j < kLnVec; ++j) {
        const int k = lane * 8 + 512 * j;
        if (k < K && b < B) {
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = pack_bf16x2((lo_bf16(xv[i][j][e]) - mean[i]) * rstd[i] * lo_bf16(gv[j][e]) + lo_bf16(bv[j][e]),
                               (hi_bf16(xv[i][j][e]) - mean[i]) * rstd[i] * hi_bf16(gv[j][e]) + hi_bf16(bv[j][e]));
          *reinterpret_cast<u32x4*>(row + k * 2) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- products: fragments in batches of 8, the next batch requested ahead of the current one's MFMAs ----
  // only the B real rows are staged: the MFMA's columns past B read row B - 1 again and are never stored
  const char* xfrag = smem + (l15 < B ? l15 : B - 1) * xstride + (kbeg + 8 * g) * 2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < nf; base += 8) {
    dfrag8 an[8] = {}, an2[8] = {};
    if (base + 8 < nf) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + 8 + u < nf) {
          an[u] = *reinterpret_cast<const dfrag8*>(wrow + 32 * (base + 8 + u));
          if (kPair) an2[u] = *reinterpret_cast<const dfrag8*>(wrow2 + 32 * (base + 8 + u));
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + u < nf) {
        const dfrag8 bx = *reinterpret_cast<const dfrag8*>(xfrag + 64 * (base + u));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], bx, acc, 0, 0, 0);
        if (kPair) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[u], bx, acc2, 0, 0, 0);
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = an[u];
      if (kPair) a2[u] = an2[u];
    }
  }
  reinterpret_cast<f32x4*>(part)[wave * 64 + lane] = acc;
  if (kPair) reinterpret_cast<f32x4*>(part)[NW * 64 + wave * 64 + lane] = acc2;
  __syncthreads();
  if (wave != 0) return;

  // ---- wave 0: add the four partial tiles, epilogue, store ----
  f32x4 v = acc, v2 = acc2;
#pragma unroll
  for (int wv = 1; wv < NW; ++wv) {
    v += reinterpret_cast<const f32x4*>(part)[wv * 64 + lane];
    if (kPair) v2 += reinterpret_cast<const f32x4*>(part)[NW * 64 + wv * 64 + lane];
  }
  if (l15 >= B) return;
  v[0] += lo_bf16(bw[0]);
  v[1] += hi_bf16(bw[0]);
  v[2] += lo_bf16(bw[1]);
  v[3] += hi_bf16(bw[1]);
  if (kPair) {
    v2[0] += lo_bf16(bw2[0]);
    v2[1] += hi_bf16(bw2[0]);
    v2[2] += lo_bf16(bw2[1]);
    v2[3] += hi_bf16(bw2[1]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (ACT == kDecodeGelu) v[r] = gelu_erf(v[r]);
    if (kPair) v[r] = v[r] / (1.0f + __expf(-v[r])) * v2[r];
  }
  if (rot) {
    const float e0 = v[0], o0 = v[1], e1 = v[2], o1 = v[3];
    v[0] = e0 * rc[0] - o0 * rs[0];
    v[1] = e0 * rs[0] + o0 * rc[0];
    v[2] = e1 * rc[1] - o1 * rs[1];
    v[3] = e1 * rs[1] + o1 * rc[1];
  }
  v[0] += lo_bf16(rr[0]);
  v[1] += hi_bf16(rr[0]);
  v[2] += lo_bf16(rr[1]);
  v[3] += hi_bf16(rr[1]);
  u32x2 o;
  o[0] = pack_bf16x2(v[0], v[1]);
  o[1] = pack_bf16x2(v[2], v[3]);
  bf16_t* dst = reinterpret_cast<bf16_t*>(p.y) + (long long)(row0 + l15) * p.ldy + n;
  if (p.k_cache != nullptr && n0 >= p.kv_col0) {
    // key / value columns of a fused qkv projection: row `pos` of the preallocated caches [B, S, kv_cols]
    const long long pos = *p.pos;
    const bool is_k = n0 < p.kv_col0 + p.kv_cols;
    bf16_t* cache = reinterpret_cast<bf16_t*>(is_k ? p.k_cache : p.v_cache);
    dst = cache + (long long)(row0 + l15) * p.cache_bs + pos * p.kv_cols + (n - p.kv_col0 - (is_k ? 0 : p.kv_cols));
  }
  *reinterpret_cast<u32x2*>(dst) = o;
}

}  // namespace

static bool wide_split(const DecodeLinearArgs& p) {
  return p.ln_w == nullptr && p.act != kDecodeSwiglu && p.K >= 2048 && p.K % 512 == 0;
}

bool decode_linear_ok(long long B, long long N, long long K, long long ldx, long long ldw, bool pair, bool has_ln) {
  const long long lds = (B < 16 ? B : 16) * (K * 2 + 16) + (pair ? 2 : 1) * 16384;   // the staged rows + the waves' partial tiles
  if (has_ln && K > 512 * kLnVec) return false;   // the LayerNorm prologue keeps a row in registers
  return B >= 1 && B <= kDecodeMaxRows && N >= 16 && N % 16 == 0 && K >= 128 && K % 128 == 0 && ldx % 8 == 0 && ldw % 8 == 0 &&
         lds <= 160 * 1024;
}

template <int ACT, int NW, bool LN>
static void launch(const DecodeLinearArgs& p, hipStream_t st) {
  const size_t lds = (size_t)(p.B < 16 ? p.B : 16) * ((size_t)p.K * 2 + 16) + (ACT == kDecodeSwiglu ? 2 : 1) * NW * 1024;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)decode_linear_kernel<ACT, NW, LN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((decode_linear_kernel<ACT, NW, LN>), dim3(p.N / 16, (p.B + 15) / 16), dim3(NW * 64), lds, st, p);
}

void decode_linear(const DecodeLinearArgs& p, hipStream_t st) {
  const bool ln = p.ln_w != nullptr;
  if (wide_split(p)) {
    p.act == kDecodeGelu ? launch<kDecodeGelu, 16, false>(p, st) : launch<kDecodeNone, 16, false>(p, st);
  } else if (p.act == kDecodeGelu) {
    ln ? launch<kDecodeGelu, 4, true>(p, st) : launch<kDecodeGelu, 4, false>(p, st);
  } else if (p.act == kDecodeSwiglu) {
    ln ? launch<kDecodeSwiglu, 4, true>(p, st) : launch<kDecodeSwiglu, 4, false>(p, st);
  } else {
    ln ? launch<kDecodeNone, 4, true>(p, st) : launch<kDecodeNone, 4, false>(p, st);
  }
}

}  // namespace dpa
