// LDS access-pattern probe for gfx950: cycles per ds_read_b128 / ds_read_b64_tr_b16 wave
// instruction for the address patterns the kernels use, measured with one workgroup so that the
// only limiter is the LDS pipeline itself (bank conflicts show up as extra cycles).
//
//   hipcc --offload-arch=gfx950 -O3 -o build/lds_probe benchmarks/lds_probe.hip && build/lds_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)); }
__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)); }

// Byte address of lane `lane` for pattern `pat`, variant `k` (k walks the k-steps / tiles so that
// consecutive instructions touch different data, as in the kernels).
__device__ int pattern_addr(int pat, int lane, int k) {
  const int r = lane & 31, hh = lane >> 5;
  switch (pat) {
    case 0: return lane * 16 + k * 1024;                                                  // linear
    case 1: return r * 128 + 16 * swz64(r, 2 * (k & 3) + hh) + (k >> 2) * 4096;            // row_frag<64>
    case 2: return r * 128 + 16 * ((2 * (k & 3) + hh) ^ (r & 7)) + (k >> 2) * 4096;        // row read, xor (row & 7)
    case 3: return r * 256 + 16 * swz128(r, 2 * (k & 7) + hh);                             // row_frag<128>
    case 4: return r * 256 + 16 * ((2 * (k & 7) + hh) ^ (r & 15));                         // row read 128, xor (row & 15)
    case 5: return r * 128 + 16 * (2 * (k & 3) + hh) + (k >> 2) * 4096;                    // unswizzled rows of 128 B
    case 6: return r * 128 + 16 * ((2 * (k & 3) + hh) ^ ((r >> 1) & 7)) + (k >> 2) * 4096; // xor (row >> 1)
    case 7: {   // Stage64<64>::commit: 8 lanes per 128-byte row, 8 rows per wave instruction
      const int row = 8 * k + (lane >> 3);
      return row * 128 + 16 * swz64(row, lane & 7);
    }
    case 8: {   // Stage64<128>::commit: 16 lanes per 256-byte row, 4 rows per wave instruction
      const int row = 4 * k + (lane >> 4);
      return row * 256 + 16 * swz128(row, lane & 15);
    }
    case 9: {   // the same 8-lanes-per-row layout written without the swizzle
      const int row = 8 * k + (lane >> 3);
      return row * 128 + 16 * (lane & 7);
    }
    default: return 0;
  }
}

// tr16_b64 read address (8 bytes per lane) of the tr_frag helpers: HS = 64 or 128, part jj.
__device__ int tr_addr(int hs, int lane, int s, int dt, int jj, bool natural) {
  const int hh = lane >> 5, g = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  const int chunk = 4 * dt + 2 * g + (pp >> 1);
  const int row = natural ? 16 * s + 8 * hh + 4 * jj + qq : 16 * s + 8 * jj + 4 * hh + qq;
  const int sw = hs == 64 ? swz64(row, chunk) : swz128(row, chunk);
  return row * (hs * 2) + 16 * sw + 8 * (pp & 1);
}

template <int MODE>  // 0: ds_read_b128 patterns, 1: tr reads, 2: ds_write_b128 patterns
__global__ void probe(int pat, int iters, long long* cycles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  int addr[8];
  for (int k = 0; k < 8; ++k) {
    if (MODE == 1) addr[k] = tr_addr(pat & 1 ? 128 : 64, lane, k & 3, (k >> 2) & 1, 0, (pat >> 1) & 1);
    else addr[k] = pattern_addr(pat, lane, k);
  }
  unsigned acc = 0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const u32x4*>(smem + addr[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc ^= v[k][0] ^ v[k][3];
    } else if (MODE == 1) {
      s16x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + addr[k]));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc ^= (unsigned)v[k][0] ^ (unsigned)v[k][3];
    } else {
      u32x4 w = {acc, acc, acc, (unsigned)it};
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<u32x4*>(smem + addr[k]) = w;
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = acc + reinterpret_cast<unsigned*>(smem)[threadIdx.x];
}

int main() {
  long long* dc;
  unsigned* ds;
  hipMalloc(&dc, 8);
  hipMalloc(&ds, 4096);
  const int iters = 2000;
  const char* names0[] = {"linear", "row_frag<64> (swz64)", "rows128B xor(row&7)", "row_frag<128> (swz128)",
                          "rows256B xor(row&15)", "rows128B unswizzled", "rows128B xor(row>>1)",
                          "Stage64<64> commit (8 lanes/row)", "Stage64<128> commit (16 lanes/row)",
                          "8 lanes/row, no swizzle"};
  for (int waves = 1; waves <= 4; waves *= 4) {
    for (int mode = 0; mode < 3; ++mode) {
      const int npat = mode == 1 ? 4 : 10;
      for (int pat = 0; pat < npat; ++pat) {
        long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {
          if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64 * waves), 65536, 0, pat, iters, dc, ds);
          if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64 * waves), 65536, 0, pat, iters, dc, ds);
          if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64 * waves), 65536, 0, pat, iters, dc, ds);
          hipDeviceSynchronize();
          hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        }
        const double per = (double)c / (iters * 8.0);
        if (mode == 1)
          printf("waves %d  tr16_b64 hs%d %s: %.2f clk per wave-instruction (per CU: %.2f)\n", waves,
                 pat & 1 ? 128 : 64, (pat >> 1) & 1 ? "natural" : "acc-order", per, per / waves);
        else
          printf("waves %d  %s %-26s: %.2f clk per wave-instruction (per CU: %.2f)\n", waves,
                 mode == 0 ? "ds_read_b128 " : "ds_write_b128", names0[pat], per, per / waves);
      }
    }
  }
  return 0;
}
