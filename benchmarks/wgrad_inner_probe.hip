// Inner-loop ceiling of the weight-gradient kernel on gfx950: 4 waves, each owning a 128x128 fp32
// accumulator tile, consume [64 tokens][128] bf16 LDS tiles (no global traffic, optional barrier per
// 64-token stage). Two ways to build the MFMA operands of a product that contracts over the tile ROWS:
//   mode 0: ds_read_b64_tr_b16 transposed reads (16 per 16-token step and wave)
//   mode 1: plain ds_read_b64 of 4 columns x 8 tokens per lane (16 per step) + 32 v_perm_b32 that
//           transpose the 16-bit pairs in registers; MFMA i of a lane then owns column 4*r + i
// Prints the MFMA pipe utilisation each variant reaches.
//
//   hipcc --offload-arch=gfx950 -O3 -I csrc -o build/wgrad_inner_probe benchmarks/wgrad_inner_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#include "mfma_tiles.h"

using namespace dpa;

__device__ __forceinline__ void perm_frags(const unsigned (&r)[8][2], bfrag (&q)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x4 v;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      v[d] = __builtin_amdgcn_perm(r[2 * d + 1][j >> 1], r[2 * d][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
    q[j] = __builtin_bit_cast(bfrag, v);
  }
}

template <int MODE, bool BARRIER>
__global__ __launch_bounds__(256, 1) void inner(int iters, float* sink) {
  constexpr int SUB = 64 * 128 * 2, STAGE = 4 * SUB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, r = lane & 31;
  for (int i = tid; i < 2 * STAGE / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  const int wn = wave >> 1, wk = wave & 1;
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (BARRIER) __builtin_amdgcn_s_barrier();
    const char* ya = smem + (it & 1) * STAGE + wn * SUB;
    const char* xb = smem + (it & 1) * STAGE + (2 + wk) * SUB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bfrag a[4], b[4];
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = tr_frag<128>(ya, 0, s, i, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = tr_frag<128>(xb, 0, s, j, lane);
      } else {
        unsigned ra[8][2], rb[8][2];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int off = (16 * s + 8 * hh + t) * 256 + 8 * r;
          const uint2 va = *reinterpret_cast<const uint2*>(ya + off);
          const uint2 vb = *reinterpret_cast<const uint2*>(xb + off);
          ra[t][0] = va.x; ra[t][1] = va.y;
          rb[t][0] = vb.x; rb[t][1] = vb.y;
        }
        perm_frags(ra, a);
        perm_frags(rb, b);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  }
  float tot = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) tot += acc[a][b][i];
  if (tot == 12345.f) sink[blockIdx.x * 256 + tid] = tot;
}

template <int MODE, bool BARRIER>
static void run(const char* name, float* sink) {
  const int iters = 4000, lds = 2 * 4 * 64 * 128 * 2;
  hipFuncSetAttribute(reinterpret_cast<const void*>(inner<MODE, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((inner<MODE, BARRIER>), dim3(256), dim3(256), lds, 0, 200, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((inner<MODE, BARRIER>), dim3(256), dim3(256), lds, 0, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 4 * iters * 64 * (2.0 * 32 * 32 * 16);
  printf("%-34s %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)\n", name, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500 * 100);
}

int main() {
  float* sink;
  hipMalloc(&sink, 256 * 256 * sizeof(float));
  run<0, false>("tr_b64 reads", sink);
  run<0, true>("tr_b64 reads + barrier/stage", sink);
  run<1, false>("b64 reads + v_perm", sink);
  run<1, true>("b64 reads + v_perm + barrier/stage", sink);
  return 0;
}
