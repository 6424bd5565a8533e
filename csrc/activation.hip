// Elementwise activation kernels (exact-erf GELU, SwiGLU, and the other names of the reference's
// activation map) for gfx950, forward and backward.
// 16-byte vector accesses, grid-stride, fp32 math. These are the stand-alone forms used where
// the activation is not folded into a GEMM epilogue.
//
// Replaces `nn.GELU()` / `F.silu(x1) * x2` of the reference MLP
// (reference: single-gpu/model.py:376-394).
#include "common.h"
#include "kernels.h"

namespace dpa {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float a[8];
    load8(x + i * 8, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = gelu_f(a[j]);
    store8(y + i * 8, a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       T* __restrict__ dx, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float a[8], d[8];
    load8(x + i * 8, a);
    load8(dy + i * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] *= gelu_grad_f(a[j]);
    store8(dx + i * 8, d);
  }
}

// h is [M, 2U] (gate half first, value half second); y is [M, U].
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, long long M,
                                                         int U) {
  const int uvec = U >> 3;
  const long long total = M * uvec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / uvec;
    const int c = (int)(i - r * uvec) * 8;
    float g[8], v[8];
    load8(h + r * 2 * U + c, g);
    load8(h + r * 2 * U + U + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = g[j] * sigmoid_f(g[j]) * v[j];
    store8(y + r * U + c, g);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ h,
                                                         T* __restrict__ dh, long long M, int U) {
  const int uvec = U >> 3;
  const long long total = M * uvec;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / uvec;
    const int c = (int)(i - r * uvec) * 8;
    float g[8], v[8], d[8], dg[8], dv[8];
    load8(h + r * 2 * U + c, g);
    load8(h + r * 2 * U + U + c, v);
    load8(dy + r * U + c, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = sigmoid_f(g[j]);
      const float silu = g[j] * s;
      dv[j] = d[j] * silu;
      dg[j] = d[j] * v[j] * (s + silu * (1.f - s));
    }
    store8(dh + r * 2 * U + c, dg);
    store8(dh + r * 2 * U + U + c, dv);
  }
}

// The remaining activations of the reference map, one kernel templated on the function. The backward
// recomputes from the saved input x (no saved output).
enum ActKind { ACT_RELU = 0, ACT_SILU, ACT_MISH, ACT_SELU, ACT_ELU, ACT_SIGMOID, ACT_TANH, ACT_LRELU, ACT_COUNT };

template <int KIND>
__device__ __forceinline__ float act_f(float x) {
  constexpr float kSeluA = 1.6732632423543772f, kSeluS = 1.0507009873554805f;
  switch (KIND) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_SILU: return x * sigmoid_f(x);
    case ACT_MISH: {
      const float sp = x > 20.f ? x : log1pf(__expf(x));   // softplus with torch's threshold
      return x * tanhf(sp);
    }
    case ACT_SELU: return kSeluS * (x > 0.f ? x : kSeluA * (__expf(x) - 1.f));
    case ACT_ELU: return x > 0.f ? x : __expf(x) - 1.f;     // also CELU with alpha = 1
    case ACT_SIGMOID: return sigmoid_f(x);
    case ACT_TANH: return tanhf(x);
    default: return x > 0.f ? x : 0.01f * x;                 // leaky ReLU, slope 0.01
  }
}

template <int KIND>
__device__ __forceinline__ float act_grad_f(float x) {
  constexpr float kSeluA = 1.6732632423543772f, kSeluS = 1.0507009873554805f;
  switch (KIND) {
    case ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ACT_SILU: {
      const float s = sigmoid_f(x);
      return s * (1.f + x * (1.f - s));
    }
    case ACT_MISH: {
      const float sp = x > 20.f ? x : log1pf(__expf(x));
      const float t = tanhf(sp);
      return t + x * (1.f - t * t) * sigmoid_f(x);
    }
    case ACT_SELU: return kSeluS * (x > 0.f ? 1.f : kSeluA * __expf(x));
    case ACT_ELU: return x > 0.f ? 1.f : __expf(x);
    case ACT_SIGMOID: {
      const float s = sigmoid_f(x);
      return s * (1.f - s);
    }
    case ACT_TANH: {
      const float t = tanhf(x);
      return 1.f - t * t;
    }
    default: return x > 0.f ? 1.f : 0.01f;
  }
}

template <typename T, int KIND>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float a[8];
    load8(x + i * 8, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = act_f<KIND>(a[j]);
    store8(y + i * 8, a);
  }
}

template <typename T, int KIND>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                      T* __restrict__ dx, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float a[8], d[8];
    load8(x + i * 8, a);
    load8(dy + i * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] *= act_grad_f<KIND>(a[j]);
    store8(dx + i * 8, d);
  }
}

static int ew_grid(long long nthreads) {
  long long b = (nthreads + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

void gelu_fwd(const void* x, void* y, long long n, bool is_bf16, hipStream_t st) {
  const long long nvec = n >> 3;
  if (nvec == 0) return;
  if (is_bf16)
    hipLaunchKernelGGL((gelu_fwd_kernel<bf16_t>), dim3(ew_grid(nvec)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, nvec);
  else
    hipLaunchKernelGGL((gelu_fwd_kernel<float>), dim3(ew_grid(nvec)), dim3(256), 0, st, (const float*)x, (float*)y, nvec);
}

void gelu_bwd(const void* dy, const void* x, void* dx, long long n, bool is_bf16, hipStream_t st) {
  const long long nvec = n >> 3;
  if (nvec == 0) return;
  if (is_bf16)
    hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t>), dim3(ew_grid(nvec)), dim3(256), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)x, (bf16_t*)dx, nvec);
  else
    hipLaunchKernelGGL((gelu_bwd_kernel<float>), dim3(ew_grid(nvec)), dim3(256), 0, st, (const float*)dy,
                       (const float*)x, (float*)dx, nvec);
}

template <typename T, int KIND>
static void act_launch(const void* dy, const void* x, void* out, long long nvec, hipStream_t st) {
  const dim3 grid(ew_grid(nvec)), block(256);
  if (dy == nullptr)
    hipLaunchKernelGGL((act_fwd_kernel<T, KIND>), grid, block, 0, st, (const T*)x, (T*)out, nvec);
  else
    hipLaunchKernelGGL((act_bwd_kernel<T, KIND>), grid, block, 0, st, (const T*)dy, (const T*)x, (T*)out, nvec);
}

template <typename T>
static void act_dispatch(int kind, const void* dy, const void* x, void* out, long long nvec, hipStream_t st) {
  switch (kind) {
    case ACT_RELU: return act_launch<T, ACT_RELU>(dy, x, out, nvec, st);
    case ACT_SILU: return act_launch<T, ACT_SILU>(dy, x, out, nvec, st);
    case ACT_MISH: return act_launch<T, ACT_MISH>(dy, x, out, nvec, st);
    case ACT_SELU: return act_launch<T, ACT_SELU>(dy, x, out, nvec, st);
    case ACT_ELU: return act_launch<T, ACT_ELU>(dy, x, out, nvec, st);
    case ACT_SIGMOID: return act_launch<T, ACT_SIGMOID>(dy, x, out, nvec, st);
    case ACT_TANH: return act_launch<T, ACT_TANH>(dy, x, out, nvec, st);
    default: return act_launch<T, ACT_LRELU>(dy, x, out, nvec, st);
  }
}

int act_kinds() { return ACT_COUNT; }

// dy == nullptr: out = f(x); else out = dy * f'(x). kind indexes ActKind.
void act_apply(int kind, const void* dy, const void* x, void* out, long long n, bool is_bf16, hipStream_t st) {
  const long long nvec = n >> 3;
  if (nvec == 0) return;
  if (is_bf16) act_dispatch<bf16_t>(kind, dy, x, out, nvec, st);
  else act_dispatch<float>(kind, dy, x, out, nvec, st);
}

void swiglu_fwd(const void* h, void* y, long long M, int U, bool is_bf16, hipStream_t st) {
  if (M == 0) return;
  const dim3 grid(ew_grid(M * (U >> 3))), block(256);
  if (is_bf16)
    hipLaunchKernelGGL((swiglu_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)h, (bf16_t*)y, M, U);
  else
    hipLaunchKernelGGL((swiglu_fwd_kernel<float>), grid, block, 0, st, (const float*)h, (float*)y, M, U);
}

void swiglu_bwd(const void* dy, const void* h, void* dh, long long M, int U, bool is_bf16, hipStream_t st) {
  if (M == 0) return;
  const dim3 grid(ew_grid(M * (U >> 3))), block(256);
  if (is_bf16)
    hipLaunchKernelGGL((swiglu_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)h,
                       (bf16_t*)dh, M, U);
  else
    hipLaunchKernelGGL((swiglu_bwd_kernel<float>), grid, block, 0, st, (const float*)dy, (const float*)h,
                       (float*)dh, M, U);
}

}  // namespace dpa
