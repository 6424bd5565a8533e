// Rotary position embedding for gfx950: in-place interleaved-pair rotation of a block of heads
// inside a (possibly fused QKV) activation buffer, using host-precomputed fp32 cos/sin tables
// (on-device trig would turn this memory-bound pass into a VALU-bound one). The inverse
// rotation is the backward pass.
//
// Replaces `LLMconfig.apply_rotary_emb` and its complex64 table
// (reference: single-gpu/model.py:77-96,566-577).
#include "common.h"
#include "kernels.h"

namespace dpa {

template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ x, long long bs, int ts, int B, int T_len,
                                                   int nheads, int hs, const float* __restrict__ cos_t,
                                                   const float* __restrict__ sin_t, int pos0, float sgn) {
  const int vph = hs >> 3;  // 8-element vectors per head
  const long long total = (long long)B * T_len * nheads * vph;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vph);
    long long rest = i / vph;
    const int h = (int)(rest % nheads);
    rest /= nheads;
    const int t = (int)(rest % T_len);
    const int b = (int)(rest / T_len);
    T* p = x + b * bs + (long long)t * ts + h * hs + v * 8;
    float a[8];
    load8(p, a);
    const f32x4 cs = *reinterpret_cast<const f32x4*>(cos_t + (long long)(pos0 + t) * (hs >> 1) + v * 4);
    const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_t + (long long)(pos0 + t) * (hs >> 1) + v * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float re = a[2 * j], im = a[2 * j + 1], s = sn[j] * sgn;
      a[2 * j] = re * cs[j] - im * s;
      a[2 * j + 1] = re * s + im * cs[j];
    }
    store8(p, a);
  }
}

void rope_inplace(void* x, long long bs, int ts, int B, int T_len, int nheads, int hs, const float* cos_t,
                  const float* sin_t, int pos0, bool inverse, bool is_bf16, hipStream_t st) {
  const long long total = (long long)B * T_len * nheads * (hs >> 3);
  if (total == 0) return;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const float sgn = inverse ? -1.f : 1.f;
  if (is_bf16)
    hipLaunchKernelGGL((rope_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (bf16_t*)x, bs, ts, B, T_len,
                       nheads, hs, cos_t, sin_t, pos0, sgn);
  else
    hipLaunchKernelGGL((rope_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (float*)x, bs, ts, B, T_len,
                       nheads, hs, cos_t, sin_t, pos0, sgn);
}

}  // namespace dpa
