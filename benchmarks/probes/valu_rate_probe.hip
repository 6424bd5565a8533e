// Issue cost of VALU instruction kinds on gfx950: one workgroup of 1 - 16 waves (up to four per SIMD), every wave runs
// N dependent-free instructions of one kind in a loop, clocks by s_memtime. Prints ticks per instruction for wave 0 (the
// oldest wave, which the issue arbiter prefers) and per instruction and SIMD over all waves (the pipe's throughput).
// Build: hipcc -O3 --offload-arch=gfx950 benchmarks/probes/valu_rate_probe.hip -o benchmarks/probes/valu_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void probe(float* out, unsigned long long* clk, int iters) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {        // v_exp_f32, 8 independent chains
      REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 1) { // v_fma_f32
      REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 2) { // v_pk_fma_f32
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (KIND == 3) { // v_add_f32
      REP16(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 4) { // v_cvt_pk_bf16_f32
      REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 5) { // v_rcp_f32
      REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 6) { // v_exp_f32 alternating with v_add_f32 (does an add hide under a transcendental?)
      REP16(asm volatile("v_exp_f32 %0, %0\n v_add_f32 %4, %4, %4\n v_exp_f32 %1, %1\n v_add_f32 %5, %5, %5\n v_exp_f32 %2, %2\n v_add_f32 %6, %6, %6\n v_exp_f32 %3, %3\n v_add_f32 %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 7) { // v_pk_add_f32
      REP16(asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (KIND == 8) { // v_exp_f16
      REP16(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  asm volatile("s_nop 7\n s_nop 7");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
  if ((threadIdx.x & 63) == 0) {   // every wave: start and end
    clk[2 * (threadIdx.x >> 6)] = t0;
    clk[2 * (threadIdx.x >> 6) + 1] = t1;
  }
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long* clk, int waves) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64 * waves), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
  }
  unsigned long long h[32];
  hipMemcpy(h, clk, 16 * waves, hipMemcpyDeviceToHost);
  unsigned long long lo = h[0], hi = h[1];
  for (int w = 0; w < waves; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
  const int per_simd = (waves + 3) / 4;
  printf("%-34s waves/workgroup %2d: wave 0 %6.2f ticks per instruction; all waves %6.2f ticks per instruction and SIMD\n", name, waves,
         (double)(h[1] - h[0]) / (iters * 128.0), (double)(hi - lo) / (iters * 128.0 * per_simd));
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4096 * 4); hipMalloc(&clk, 8 * 32);
  for (int waves : {1, 4, 8, 12, 16}) {
    run<1>("v_fma_f32", out, clk, waves);
    run<3>("v_add_f32", out, clk, waves);
    run<2>("v_pk_fma_f32", out, clk, waves);
    run<7>("v_pk_add_f32", out, clk, waves);
    run<4>("v_cvt_pk_bf16_f32", out, clk, waves);
    run<0>("v_exp_f32", out, clk, waves);
    run<8>("v_exp_f16", out, clk, waves);
    run<5>("v_rcp_f32", out, clk, waves);
    run<6>("v_exp_f32 + v_add_f32 alternating", out, clk, waves);
  }
  return 0;
}
