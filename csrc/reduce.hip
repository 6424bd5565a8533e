// Column sums of a row-major [M, N] activation (bias gradients): out[n] += sum_m x[m, n].
// Every thread owns 8 adjacent columns (one 16-byte load per row), a workgroup of 4 waves
// covers 512 columns x a strip of rows, the waves are combined through LDS and one float
// atomic per column and workgroup lands in the fp32 output (zeroed by the caller).
//
// Replaces the `dy.sum(0)` reductions that `nn.Linear`'s backward issues for its bias
// (reference: single-gpu/model.py:112-114, biases of c_attn / c_proj).
#include "common.h"
#include "kernels.h"

namespace dpa {

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long long M,
                                                     int N, long long ld, int rows_per_block) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 512 + lane * 8;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < N) {
    for (long long r = r0 + wave; r < r1; r += 4) {
      float v[8];
      load8(x + r * ld + c0, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int col = blockIdx.x * 512 + c;
    if (col < N) atomicAdd(out + col, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

void colsum(const void* x, float* out, long long M, int N, long long ld, bool is_bf16, hipStream_t st) {
  if (M == 0 || N == 0) return;
  const int cblocks = (N + 511) / 512;
  long long strips = 2048 / cblocks;
  if (strips < 1) strips = 1;
  long long rpb = (M + strips - 1) / strips;
  if (rpb < 16) rpb = 16;
  strips = (M + rpb - 1) / rpb;
  const dim3 grid(cblocks, (unsigned)strips), block(256);
  if (is_bf16)
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, out, M, N, ld, (int)rpb);
  else
    hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, st, (const float*)x, out, M, N, ld, (int)rpb);
}

// x *= g[0] / expected[0] (both read from device memory), unless the two are equal (then every thread leaves after reading g: no pass over x). The fused
// head loss writes its gradients already multiplied by the factor the caller said the loss would be scaled by; its
// backward only has to correct them when the incoming gradient turns out to be something else.
template <typename T>
__global__ __launch_bounds__(256) void scale_unless_kernel(T* __restrict__ x, long long nvec, const float* __restrict__ g,
                                                           const float* __restrict__ expected) {
  const float r = g[0] / expected[0];
  if (r == 1.0f) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= r;
    store8(x + i * 8, v);
  }
}

void scale_unless(void* x, long long n, bool is_bf16, const float* g, const float* expected, hipStream_t st) {
  const long long nvec = n >> 3;
  if (nvec == 0) return;
  long long blocks = (nvec + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (is_bf16)
    hipLaunchKernelGGL((scale_unless_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (bf16_t*)x, nvec, g, expected);
  else
    hipLaunchKernelGGL((scale_unless_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (float*)x, nvec, g, expected);
}

// Benchmark aid: `blocks` workgroups that each hold a CU busy for about `usec` microseconds (a stand-in
// for a collective's kernels when the effect of lost CUs on the compute kernels is measured on one GPU).
// Every wave leaves after the deadline or after a bounded number of polls, whichever comes first.
__global__ __launch_bounds__(256) void spin_kernel(long long ticks, float* sink) {
  const long long t0 = wall_clock64();
  float acc = 0.f;
  for (int i = 0; i < (1 << 24); ++i) {
    acc += 1e-9f * (float)i;
    if ((i & 63) == 0 && wall_clock64() - t0 >= ticks) break;
  }
  if (acc == -1.f) sink[0] = acc;
}

void spin(int blocks, int usec, float* sink, hipStream_t st) {
  if (blocks <= 0 || usec <= 0) return;
  // wall_clock64 counts at 100 MHz on this part
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, st, (long long)usec * 100, sink);
}

}  // namespace dpa
