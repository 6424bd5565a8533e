// Next-token sampling in one kernel (gfx950): temperature, top-k filter, softmax and one multinomial
// draw per row. One workgroup of 1024 threads per row; the row (V <= 57344 logits) is loaded once with
// coalesced 16-byte accesses and stays in registers (packed bf16) for all the passes:
//   1. row maximum;
//   2. the k-th largest logit by a bitwise search on the order-preserving integer image of the floats, which
//      gives the same keep set as `logits < topk(logits, k).values[:, -1]` masking, ties included. For
//      k <= 512 the search runs over the few logits above a lower bound (the k-th largest of the threads' own
//      maxima), two bits per round, by one wave without workgroup barriers; otherwise over the whole row, one
//      count-and-reduce per significant bit;
//   3. per-thread partial sums of exp((x - max) / T) over the kept logits, a scan over the threads, and a
//      walk of the one thread's registers in which u * total falls (u uniform in [0, 1) from a hash of the
//      seed and the row). The order of the walk is a fixed permutation of the vocabulary, which a
//      multinomial draw does not care about.
// Decode-step form (SampleStep): the kernel also records the token in the generation's history at the device-side
// column, and the last workgroup to finish advances the step's device counters (cache position, fill level,
// history column), so a captured decode step needs no bookkeeping launches and no per-step seed draw.
// Replaces the topk / masked_fill / softmax / multinomial chain of the reference's generate()
// (reference: single-gpu/model.py:733-745) with a single launch inside the captured decode step.
#include "common.h"
#include "kernels.h"

namespace dpa {

constexpr int kSampleThreads = 1024;
constexpr int kSampleVecs = 7;   // 8-element vectors per thread: V <= 1024 * 7 * 8

__device__ __forceinline__ unsigned smix32(unsigned x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}

// Monotone map float -> uint32 (larger float, larger key). NaN sorts above +inf; callers feed finite logits.
__device__ __forceinline__ unsigned order_key(float x) {
  const unsigned b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Sum of an int over the workgroup (every thread gets the result). `red` holds 16 ints of LDS.
__device__ __forceinline__ int block_count(int v, int* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  int r = 0;
#pragma unroll
  for (int i = 0; i < kSampleThreads / 64; ++i) r += red[i];
  return r;
}

// Sum of an unsigned over the wave: four rotations inside each row of 16 lanes (DPP), two exchanges between rows.
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);
  v += (unsigned)__shfl_xor((int)v, 16, 64);
  v += (unsigned)__shfl_xor((int)v, 32, 64);
  return v;
}

// By ONE wave: the largest key such that at least `topk` of keys[0 .. n) (LDS, n <= 2048, key 0 = none) are >= it.
// Two bits per round, the three candidate prefixes of a round counted together; the lane's keys stay in registers
// and the rounds need no workgroup barrier (a round over the whole workgroup costs about a microsecond of barriers).
__device__ __forceinline__ unsigned wave_search_kth_key(const unsigned* keys, int n, int topk, int low_bit, int lane) {
  unsigned k[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) k[i] = (lane + 64 * i) < n ? keys[lane + 64 * i] : 0u;
  const int used = (n + 63) >> 6;
  unsigned prefix = 0u;
  for (int bit = 30; bit >= low_bit; bit -= 2) {
    const unsigned c1 = prefix | (1u << bit), c2 = prefix | (2u << bit), c3 = prefix | (3u << bit);
    unsigned n12 = 0u, n3 = 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i)
      if (i < used) {
        n12 += (k[i] >= c1 ? 1u : 0u) + (k[i] >= c2 ? 0x10000u : 0u);
        n3 += k[i] >= c3 ? 1u : 0u;
      }
    n12 = wave_sum_u32(n12);
    n3 = wave_sum_u32(n3);
    const int n1 = (int)(n12 & 0xFFFFu), n2 = (int)(n12 >> 16);
    prefix = (int)n3 >= topk ? c3 : (n2 >= topk ? c2 : (n1 >= topk ? c1 : prefix));
  }
  return prefix;
}

constexpr int kSampleCand = 2048;   // capacity of the candidate list of the top-k search (two per thread)

template <typename T>
__global__ __launch_bounds__(kSampleThreads) void sample_topk_kernel(const T* __restrict__ logits, long long ld, int V,
                                                                    float inv_temp, int topk,
                                                                    const long long* __restrict__ seed,
                                                                    long long* __restrict__ out, SampleStep step) {
  __shared__ float red[16];
  __shared__ int ired[16];
  __shared__ float wave_tot[kSampleThreads / 64];
  __shared__ int chosen;
  __shared__ float chosen_rest;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const T* x = logits + (long long)row * ld;
  const int nvec = V >> 3;   // V % 8 == 0 (host-checked)

  // vector k of this thread is row vector tid + k * 1024 (coalesced); absent vectors hold -inf
  Vec8<T> v[kSampleVecs];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < kSampleVecs; ++k) {
    const int i = tid + k * kSampleThreads;
    if (i < nvec) {
      v[k].load(x + (long long)i * 8);
      float e[8];
      v[k].get(e);
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmaxf(mx, e[j]);
    }
  }
  const float tmax = mx;   // this thread's own maximum
  mx = block_max(mx, red);

  unsigned thr_key = 0u;   // keep everything
  bool found = false;
  constexpr int kLow = sizeof(T) == 2 ? 16 : 0;   // bf16 images have no low 16 bits
  if (topk > 0 && topk < V && topk <= kSampleThreads / 2) {
    // Counting all 56 logits of every thread once per bit is bound by vector instructions on the row's one
    // workgroup. The topk-th largest of the THREAD MAXIMA is a lower bound of the topk-th largest logit (those
    // maxima are topk different logits), and few logits lie above it: they are gathered in LDS and the search
    // runs over them alone. For a candidate prefix below the bound every count is >= topk either way, above it
    // the counts over the list are the counts over the row, so the result is the same key.
    __shared__ unsigned tkeys[kSampleThreads];
    __shared__ unsigned cand[kSampleCand];
    __shared__ int ncand;
    __shared__ unsigned found_key;
    tkeys[tid] = tid < nvec ? order_key(tmax) : 0u;
    if (tid == 0) ncand = 0;
    __syncthreads();
    if (wid == 0) {
      const unsigned bnd = wave_search_kth_key(tkeys, kSampleThreads, topk, kLow, lane);
      if (lane == 0) found_key = bnd;
    }
    __syncthreads();
    const unsigned bound = found_key;
#pragma unroll
    for (int k = 0; k < kSampleVecs; ++k) {
      if (tid + k * kSampleThreads < nvec) {
        float e[8];
        v[k].get(e);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned key = order_key(e[j]);
          if (key >= bound) {
            const int at = atomicAdd(&ncand, 1);
            if (at < kSampleCand) cand[at] = key;
          }
        }
      }
    }
    __syncthreads();
    const int n = ncand;
    if (n <= kSampleCand) {
      if (wid == 0) {
        const unsigned key = wave_search_kth_key(cand, n, topk, kLow, lane);
        if (lane == 0) found_key = key;
      }
      __syncthreads();
      thr_key = found_key;
      found = true;
    }
  }
  if (topk > 0 && topk < V && !found) {
    // largest key such that at least `topk` logits are >= it: the key of the k-th largest logit
    unsigned prefix = 0u;
    for (int bit = 31; bit >= kLow; --bit) {
      const unsigned cand = prefix | (1u << bit);
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < kSampleVecs; ++k) {
        if (tid + k * kSampleThreads < nvec) {
          float e[8];
          v[k].get(e);
#pragma unroll
          for (int j = 0; j < 8; ++j) cnt += order_key(e[j]) >= cand ? 1 : 0;
        }
      }
      if (block_count(cnt, ired) >= topk) prefix = cand;
    }
    thr_key = prefix;
  }

  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kSampleVecs; ++k) {
    if (tid + k * kSampleThreads < nvec) {
      float e[8];
      v[k].get(e);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (order_key(e[j]) >= thr_key) s += __expf((e[j] - mx) * inv_temp);
    }
  }
  // exclusive prefix of s over the threads (thread order = walk order) and the total
  float incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (tid == 0) chosen = -1;
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  float base = 0.f, total = 0.f;
#pragma unroll
  for (int w = 0; w < kSampleThreads / 64; ++w) {
    if (w < wid) base += wave_tot[w];
    total += wave_tot[w];
  }
  const float excl = base + incl - s;
  // decode-step form: the seed is fixed for the whole generation and the position counter makes the draws differ
  const unsigned long long sd = (unsigned long long)*seed + (step.pos != nullptr ? 0x9E3779B97F4A7C15ull * (unsigned long long)*step.pos : 0ull);
  const unsigned h = smix32((unsigned)sd ^ smix32((unsigned)(sd >> 32) + 0x9E3779B9u * (unsigned)(row + 1)));
  const float target = (float)(h >> 8) * (1.0f / 16777216.0f) * total;   // u in [0, 1) with 24 random bits
  if (s > 0.f && target >= excl && target < excl + s) {
    chosen = tid;                 // intervals are disjoint: at most one thread qualifies
    chosen_rest = target - excl;
  }
  __syncthreads();
  if (chosen < 0) {
    // rounding put the target at or past the last interval: the last thread that kept anything walks to its end
    const int cand = s > 0.f ? tid : -1;
    int best = cand;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
    if (lane == 0) ired[wid] = best;
    __syncthreads();
    if (tid == 0) {
      int b = -1;
      for (int w = 0; w < kSampleThreads / 64; ++w) b = max(b, ired[w]);
      chosen = b < 0 ? 0 : b;
      chosen_rest = INFINITY;
    }
    __syncthreads();
  }
  if (tid == chosen) {
    const float rest = chosen_rest;
    float cum = 0.f;
    int pick = -1, last = 0;
#pragma unroll
    for (int k = 0; k < kSampleVecs; ++k) {
      const int i = tid + k * kSampleThreads;
      if (i < nvec && pick < 0) {
        float e[8];
        v[k].get(e);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (pick < 0 && order_key(e[j]) >= thr_key) {
            last = i * 8 + j;
            cum += __expf((e[j] - mx) * inv_temp);
            if (cum > rest) pick = last;
          }
        }
      }
    }
    const long long token = pick >= 0 ? pick : last;
    out[row] = token;
    if (step.history != nullptr) step.history[(long long)row * step.history_ld + *step.history_pos] = token;
  }
  if (step.ticket != nullptr) {
    // The step's counters move once every row has read them: the last workgroup to get here does it.
    __shared__ unsigned arrived;
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      arrived = atomicAdd(step.ticket, 1u);
    }
    __syncthreads();
    if (arrived == gridDim.x - 1 && tid == 0) {
      *step.pos += 1;
      *step.len_after += 1;
      *step.history_pos += 1;
      *step.ticket = 0u;
    }
  }
}

bool sample_topk_ok(int V) { return V >= 8 && V % 8 == 0 && V <= kSampleThreads * kSampleVecs * 8; }

void sample_topk(const void* logits, long long ld, int rows, int V, float temperature, int topk, const long long* seed,
                 long long* out, bool is_bf16, hipStream_t st, SampleStep step) {
  if (rows == 0) return;
  const float inv_temp = temperature > 0.f ? 1.f / temperature : 1.f;
  if (is_bf16)
    hipLaunchKernelGGL((sample_topk_kernel<bf16_t>), dim3(rows), dim3(kSampleThreads), 0, st, (const bf16_t*)logits, ld, V,
                       inv_temp, topk, seed, out, step);
  else
    hipLaunchKernelGGL((sample_topk_kernel<float>), dim3(rows), dim3(kSampleThreads), 0, st, (const float*)logits, ld, V,
                       inv_temp, topk, seed, out, step);
}

}  // namespace dpa
