// Single-query attention over a preallocated KV cache for gfx950 (one new token per sequence):
//
//     o[b, h, :] = softmax(scale * q[b, h, :] K[b, :n, kv(h), :]^T) V[b, :n, kv(h), :]        n = *kv_len
//
// The flash forward kernel gives a query block of 128 rows to a workgroup and walks the keys tile after tile,
// which for ONE query is a serial chain of small tiles on a quarter-empty workgroup (8 us at 400 keys). Here the
// keys are what is divided: a workgroup serves up to 4 query heads that share a KV head, its eight waves take the
// 64-key chunks of the cache in turn (lane = key for the scores: one 16-byte load per 8 features, the query
// broadcast from LDS; lane = (8 keys, 8 features) for the values), each wave keeps its own running maximum, sum
// and output, and the partial results are merged through LDS at the end. No MFMA: a 1 x 64 x HS product per
// chunk is vector work. Keys past n are masked, so a captured decode step can replay over a growing cache.
//
// Replaces, for cached decoding, `F.scaled_dot_product_attention` on a one-token query
// (reference: single-gpu/model.py:136-149, generate loop :699-747).
#include "common.h"
#include "kernels.h"

namespace dpa {

namespace {

__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int kDecWaves = 8;   // waves per workgroup = 64-key chunks in flight per (batch, head group)

// HS: head size (64 or 128); GT: query heads per workgroup (all on one KV head)
template <int HS, int GT>
__global__ __launch_bounds__(kDecWaves * 64) void attn_decode_kernel(AttnDecodeArgs p) {
  constexpr int DPL = HS / 8;          // features per lane in the value phase (8 feature groups)
  constexpr int NV = HS / 8;           // 16-byte pieces of a key row
  __shared__ __attribute__((aligned(16))) float q_s[GT][HS];
  __shared__ __attribute__((aligned(16))) float p_s[kDecWaves][GT][64];
  __shared__ float m_s[kDecWaves][GT], l_s[kDecWaves][GT];
  __shared__ __attribute__((aligned(16))) float o_s[kDecWaves][GT][HS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int groups = p.nh / p.nkv / GT;            // workgroups per KV head
  const int bid = blockIdx.x;
  const int b = bid / (p.nkv * groups), rem = bid % (p.nkv * groups);
  const int kvh = rem / groups, h0 = kvh * (p.nh / p.nkv) + (rem % groups) * GT;
  int n = *p.kv_len;
  n = n < 0 ? 0 : (n > p.S ? p.S : n);

  const bf16_t* q = reinterpret_cast<const bf16_t*>(p.q) + (long long)b * p.q_bs + (long long)h0 * HS;
  for (int i = tid; i < GT * HS; i += kDecWaves * 64) q_s[i / HS][i % HS] = bf16_to_f32(q[i]) * p.scale;
  __syncthreads();

  const bf16_t* kb = reinterpret_cast<const bf16_t*>(p.k) + (long long)b * p.k_bs + (long long)kvh * HS;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(p.v) + (long long)b * p.v_bs + (long long)kvh * HS;
  const int kg = lane >> 3, dg = lane & 7;

  float m[GT], l[GT], o[GT][DPL];
#pragma unroll
  for (int g = 0; g < GT; ++g) {
    m[g] = -INFINITY;
    l[g] = 0.f;
#pragma unroll
    for (int d = 0; d < DPL; ++d) o[g][d] = 0.f;
  }

  for (int c0 = wave * 64; c0 < n; c0 += kDecWaves * 64) {
    // ---- scores of key c0 + lane ----
    const int key = c0 + lane;
    const bool live = key < n;
    const bf16_t* krow = kb + (long long)(live ? key : 0) * p.k_ts;
    u32x4 kv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) kv[j] = *reinterpret_cast<const u32x4*>(krow + 8 * j);
    // the value rows of this lane's 8 keys (requested now, used after the softmax arithmetic)
    u32x4 vv[8][DPL / 8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int vk = c0 + 8 * kg + i;
      const bf16_t* vrow = vb + (long long)(vk < n ? vk : 0) * p.v_ts + DPL * dg;
#pragma unroll
      for (int j = 0; j < DPL / 8; ++j) vv[i][j] = *reinterpret_cast<const u32x4*>(vrow + 8 * j);
    }
    float s[GT];
#pragma unroll
    for (int g = 0; g < GT; ++g) s[g] = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float kf[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kf[2 * e] = lo_bf16(kv[j][e]);
        kf[2 * e + 1] = hi_bf16(kv[j][e]);
      }
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(&q_s[g][8 * j]);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(&q_s[g][8 * j + 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[g] += kf[e] * q0[e] + kf[4 + e] * q1[e];
      }
    }
    // ---- running maximum / sum of this wave, probabilities to LDS in key order ----
    float alpha[GT];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      const float sg = live ? s[g] : -INFINITY;
      const float mn = fmaxf(m[g], wave_max_all(sg));      // finite: the chunk's first key is below n
      const float pj = live ? __expf(sg - mn) : 0.f;
      alpha[g] = __expf(m[g] - mn);                          // exp(-inf) = 0 on the first chunk
      l[g] = l[g] * alpha[g] + wave_sum_all(pj);
      m[g] = mn;
      p_s[wave][g][lane] = pj;
    }
    // LDS operations of one wave execute in order, so the reads below see the writes above; the fence keeps
    // the compiler from moving them across each other
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- o += p v over this lane's 8 keys and DPL features ----
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(&p_s[wave][g][8 * kg]);
      const f32x4 pb = *reinterpret_cast<const f32x4*>(&p_s[wave][g][8 * kg + 4]);
#pragma unroll
      for (int d = 0; d < DPL; ++d) o[g][d] *= alpha[g];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pi = i < 4 ? pa[i] : pb[i - 4];
#pragma unroll
        for (int j = 0; j < DPL / 8; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[g][8 * j + 2 * e] += pi * lo_bf16(vv[i][j][e]);
            o[g][8 * j + 2 * e + 1] += pi * hi_bf16(vv[i][j][e]);
          }
      }
    }
  }

  // ---- sum the 8 key groups of the wave (lanes 8 apart hold the same features), publish the wave's result ----
#pragma unroll
  for (int g = 0; g < GT; ++g) {
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
      float v = o[g][d];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      o[g][d] = v;
    }
    if (kg == 0) {
#pragma unroll
      for (int d = 0; d < DPL; ++d) o_s[wave][g][DPL * dg + d] = o[g][d];
    }
    if (lane == 0) {
      m_s[wave][g] = m[g];
      l_s[wave][g] = l[g];
    }
  }
  __syncthreads();
  // ---- merge the waves: thread (g, d) ----
  for (int i = tid; i < GT * HS; i += kDecWaves * 64) {
    const int g = i / HS, d = i % HS;
    float M = m_s[0][g];
#pragma unroll
    for (int w = 1; w < kDecWaves; ++w) M = fmaxf(M, m_s[w][g]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < kDecWaves; ++w) {
      const float f = m_s[w][g] == -INFINITY ? 0.f : __expf(m_s[w][g] - M);   // a wave without keys
      L += l_s[w][g] * f;
      acc += o_s[w][g][d] * f;
    }
    bf16_t* out = reinterpret_cast<bf16_t*>(p.o) + (long long)b * p.o_bs + (long long)(h0 + g) * HS + d;
    *out = f32_to_bf16(L > 0.f ? acc / L : 0.f);
  }
}

template <int HS>
void launch_decode(const AttnDecodeArgs& p, hipStream_t st) {
  const int G = p.nh / p.nkv;
  const int gt = G % 4 == 0 ? 4 : (G % 2 == 0 ? 2 : 1);
  const dim3 grid(p.B * p.nkv * (G / gt)), block(kDecWaves * 64);
  if (gt == 4) hipLaunchKernelGGL((attn_decode_kernel<HS, 4>), grid, block, 0, st, p);
  else if (gt == 2) hipLaunchKernelGGL((attn_decode_kernel<HS, 2>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((attn_decode_kernel<HS, 1>), grid, block, 0, st, p);
}

}  // namespace

void attn_decode(const AttnDecodeArgs& p, hipStream_t st) {
  if (p.B == 0 || p.nh == 0) return;
  if (p.hs == 64) launch_decode<64>(p, st);
  else launch_decode<128>(p, st);
}

}  // namespace dpa
