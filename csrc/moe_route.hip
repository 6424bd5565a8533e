// MoE routing for gfx950: fused top-k gate and the expert-sorted padded layout, all on the device.
//
//   moe_topk_gate   : per token, top-k of (logits + bias) over E <= 64 experts and the softmax of the
//                     selected raw logits (the gate weights). One thread per token.
//   moe_plan        : from the [N, k] expert ids, the row of every (token, slot) pair in a buffer
//                     that holds the pairs sorted by expert with every expert's segment padded to
//                     a multiple of 128 rows, plus the per-expert counts / segment bounds and the
//                     expert of each 128-row block. Stable (pairs of one expert keep their order),
//                     deterministic, atomic-free: per-block histograms -> scan -> placement.
//
// Wave-level ranking: a wave walks 64 consecutive pairs; for every distinct expert present it
// forms the ballot of the lanes that chose it, a lane's rank among them is the population count
// of the ballot below it, and lane e of the wave keeps the running count of expert e.
//
// Replaces torch.topk + gather + softmax + bincount and the argsort-based permutation
// (reference behaviour: single-gpu/model.py:447-502).
#include "common.h"
#include "kernels.h"

namespace dpa {

constexpr int kPlanChunk = 1024;   // pairs per workgroup (one wave, 16 rounds of 64)

__global__ __launch_bounds__(256) void moe_topk_gate_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ bias,
                                                            long long* __restrict__ idx, float* __restrict__ gates,
                                                            long long ntok, int E, int k) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntok) return;
  const float* row = logits + t * E;
  unsigned long long taken = 0ull;
  float sel[8];
  int who[8];
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY;
    int bi = -1;
    for (int e = 0; e < E; ++e) {
      if ((taken >> e) & 1ull) continue;
      const float v = row[e] + (bias ? bias[e] : 0.f);
      if (bi < 0 || v > best) {   // first maximum wins ties; NaN never replaces a number
        best = v;
        bi = e;
      }
    }
    taken |= 1ull << bi;
    who[j] = bi;
    sel[j] = row[bi];
  }
  float mx = sel[0];
  for (int j = 1; j < k; ++j) mx = fmaxf(mx, sel[j]);
  float sum = 0.f;
  for (int j = 0; j < k; ++j) {
    sel[j] = __expf(sel[j] - mx);
    sum += sel[j];
  }
  const float inv = 1.f / sum;
  for (int j = 0; j < k; ++j) {
    idx[t * k + j] = who[j];
    gates[t * k + j] = sel[j] * inv;
  }
}

// Expert ids are trusted to lie in [0, E); a corrupt id is folded into range so that it can never
// become an out-of-bounds row downstream.
__device__ __forceinline__ int clamp_expert(long long v, int E) {
  return v < 0 ? 0 : (v >= E ? E - 1 : (int)v);
}

// One wave per workgroup; lane e accumulates the number of pairs of expert e in this chunk.
__global__ __launch_bounds__(64) void moe_plan_hist_kernel(const long long* __restrict__ idx, long long npairs, int E,
                                                           int* __restrict__ hist) {
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * kPlanChunk;
  int cnt = 0;
  for (int c = 0; c < kPlanChunk / 64; ++c) {
    const long long j = base + c * 64 + lane;
    const int e = j < npairs ? clamp_expert(idx[j], E) : -1;
    unsigned long long active = __ballot(e >= 0);
    while (active) {
      const int leader = __ffsll((long long)active) - 1;
      const int e0 = __shfl(e, leader, 64);
      const unsigned long long mask = __ballot(e == e0);
      if (lane == e0) cnt += __popcll(mask);
      active &= ~mask;
    }
  }
  if (lane < E) hist[(long long)blockIdx.x * E + lane] = cnt;
}

// One workgroup. Thread e < E scans its expert's column of the histogram; the table is staged in
// LDS with coalesced accesses when it fits (nblocks * E <= 8192), else walked in global memory.
__global__ __launch_bounds__(64) void moe_plan_scan_kernel(int* __restrict__ hist, int nblocks, int E, int nrb, int pad,
                                                           long long* __restrict__ seg_start,
                                                           long long* __restrict__ seg_rows,
                                                           long long* __restrict__ counts,
                                                           int* __restrict__ block_group) {
  __shared__ int sh[8192];
  const int lane = threadIdx.x;
  const int total_cells = nblocks * E;
  const bool in_lds = total_cells <= 8192;
  int* tab = in_lds ? sh : hist;
  if (in_lds) {
    for (int i = lane; i < total_cells; i += 64) sh[i] = hist[i];
    __syncthreads();
  }
  int run = 0;
  if (lane < E) {
    for (int b = 0; b < nblocks; ++b) {
      const int v = tab[b * E + lane];
      tab[b * E + lane] = run;   // exclusive prefix over the chunks, per expert
      run += v;
    }
  }
  const int padded = lane < E ? (run + pad - 1) / pad * pad : 0;   // pad = 128 / 256 (GEMM row blocks) or 1 (dense)
  int incl = padded;                          // inclusive scan over the experts (lanes)
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  const int start = incl - padded;
  if (lane < E) {
    seg_start[lane] = start;
    seg_rows[lane] = padded;
    counts[lane] = run;
    for (int b = 0; b < nblocks; ++b) tab[b * E + lane] += start;   // absolute row base
    if (pad > 1)
      for (int rb = start / pad; rb < (start + padded) / pad; ++rb) block_group[rb] = lane;
  }
  if (in_lds) {
    __syncthreads();
    for (int i = lane; i < total_cells; i += 64) hist[i] = sh[i];
  }
  const int total = __shfl(incl, 63, 64);
  if (pad > 1)
    for (int rb = total / pad + lane; rb < nrb; rb += 64) block_group[rb] = E;   // blocks past the last segment
}

__global__ __launch_bounds__(64) void moe_plan_place_kernel(const long long* __restrict__ idx, long long npairs, int E,
                                                            const int* __restrict__ base_rows,
                                                            long long* __restrict__ row_of_pair) {
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * kPlanChunk;
  int cnt = lane < E ? base_rows[(long long)blockIdx.x * E + lane] : 0;   // next free row of expert `lane`
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int c = 0; c < kPlanChunk / 64; ++c) {
    const long long j = base + c * 64 + lane;
    const int e = j < npairs ? clamp_expert(idx[j], E) : -1;
    int row = 0;
    unsigned long long active = __ballot(e >= 0);
    while (active) {
      const int leader = __ffsll((long long)active) - 1;
      const int e0 = __shfl(e, leader, 64);
      const unsigned long long mask = __ballot(e == e0);
      const int first = __shfl(cnt, e0, 64);
      if (e == e0) row = first + __popcll(mask & below);
      if (lane == e0) cnt += __popcll(mask);
      active &= ~mask;
    }
    if (j < npairs) row_of_pair[j] = row;
  }
}

void moe_topk_gate(const float* logits, const float* bias, long long* idx, float* gates, long long ntok, int E, int k,
                   hipStream_t st) {
  if (ntok == 0) return;
  hipLaunchKernelGGL(moe_topk_gate_kernel, dim3((unsigned)((ntok + 255) / 256)), dim3(256), 0, st, logits, bias, idx,
                     gates, ntok, E, k);
}

int moe_plan_blocks(long long npairs) { return (int)((npairs + kPlanChunk - 1) / kPlanChunk); }

void moe_plan(const long long* idx, long long npairs, int E, int nrb, int pad, int* hist, long long* row_of_pair,
              int* block_group, long long* seg_start, long long* seg_rows, long long* counts, hipStream_t st) {
  const int nb = moe_plan_blocks(npairs);
  if (nb > 0) hipLaunchKernelGGL(moe_plan_hist_kernel, dim3(nb), dim3(64), 0, st, idx, npairs, E, hist);
  hipLaunchKernelGGL(moe_plan_scan_kernel, dim3(1), dim3(64), 0, st, hist, nb, E, nrb, pad, seg_start, seg_rows, counts,
                     block_group);
  if (nb > 0)
    hipLaunchKernelGGL(moe_plan_place_kernel, dim3(nb), dim3(64), 0, st, idx, npairs, E, hist, row_of_pair);
}

}  // namespace dpa
