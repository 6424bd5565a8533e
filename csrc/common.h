// Shared device-side helpers for the gfx950 (CDNA4 / MI355X) kernels.
// Everything here assumes wave64 and the gfx950 MFMA / LDS instruction set.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpa {

constexpr int kWave = 64;  // CDNA wavefront width

typedef uint16_t bf16_t;   // raw bf16 bits; arithmetic is always done in fp32

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}

// Round-to-nearest-even fp32 -> bf16 through the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// Two fp32 -> packed bf16x2 in one dword (lo in bits 0..15).
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  bf2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ float lo_bf16(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf16(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Load 8 consecutive elements as fp32 (16-byte load for bf16, 2x16-byte for fp32).
__device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
  u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = lo_bf16(v[i]);
    o[2 * i + 1] = hi_bf16(v[i]);
  }
}
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = a[i];
    o[4 + i] = b[i];
  }
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&o)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = v;
}
__device__ __forceinline__ void store8(float* p, const float (&o)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = o[i];
    b[i] = o[4 + i];
  }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024). `red` holds >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

// Eight consecutive elements held in their storage format: bf16 stays packed (4 registers per
// vector), so values that must live across a reduction cost half the registers of fp32.
// Streaming variants (NT = true: non-temporal cache policy) for tensors that pass through a kernel once and are too
// large to be found in a cache by their next reader: measured 5.35 -> 5.9 TB/s on the cross-entropy rows.
template <bool NT>
__device__ __forceinline__ void load8s(const bf16_t* p, float (&o)[8]) {
  const u32x4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = lo_bf16(v[i]);
    o[2 * i + 1] = hi_bf16(v[i]);
  }
}
template <bool NT>
__device__ __forceinline__ void load8s(const float* p, float (&o)[8]) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  const f32x4 a = NT ? __builtin_nontemporal_load(q) : q[0], b = NT ? __builtin_nontemporal_load(q + 1) : q[1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = a[i];
    o[4 + i] = b[i];
  }
}
template <bool NT>
__device__ __forceinline__ void store8s(bf16_t* p, const float (&o)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  else *reinterpret_cast<u32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void store8s(float* p, const float (&o)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = o[i];
    b[i] = o[4 + i];
  }
  f32x4* q = reinterpret_cast<f32x4*>(p);
  if (NT) {
    __builtin_nontemporal_store(a, q);
    __builtin_nontemporal_store(b, q + 1);
  } else {
    q[0] = a;
    q[1] = b;
  }
}
// Tensors of at least this many bytes are streamed (the 256 MiB memory-side cache holds nothing of them by the time
// their reader comes).
inline long long stream_threshold_bytes() { return 256ll << 20; }

template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  u32x4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const u32x4*>(p); }
  template <bool NT>
  __device__ __forceinline__ void loads(const bf16_t* p) {
    v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
  }
  // Makes the packed value opaque to the optimizer at this point, so an unpack that follows inside a
  // loop is not hoisted out of it (which would bring the fp32 copy back into live registers).
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v)); }
  __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = lo_bf16(v[i]);
      o[2 * i + 1] = hi_bf16(v[i]);
    }
  }
};
template <> struct Vec8<float> {
  float f[8];
  __device__ __forceinline__ void load(const float* p) { load8(p, f); }
  template <bool NT>
  __device__ __forceinline__ void loads(const float* p) { load8s<NT>(p, f); }
  __device__ __forceinline__ void pin() {}
  __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f[i];
  }
};


// Bijective XCD-aware remap of a linear workgroup id: workgroups dealt round-robin to the
// 8 XCDs are re-labelled so that each XCD (its own L2) works on a contiguous chunk of tiles.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int nx = 8;
  if (nwg < nx) return orig;
  const int xcd = orig % nx, q = nwg / nx, r = nwg % nx;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / nx;
}

}  // namespace dpa
