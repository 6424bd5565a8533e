// Stand-alone check + timing of csrc/gemm_nt.hip (no PyTorch): every epilogue against a plain fp32
// reference kernel on random operands, then TFLOP/s on the GPT-2 layer shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Icsrc benchmarks/gemm_nt_driver.hip csrc/gemm_nt.hip -o benchmarks/gemm_nt_driver.bin
// With -DDPA_WITH_FORM2 and benchmarks/experiments/gemm_nt2.hip on the command line, the modes check2 / ab / skew2 and
// DPA_FORM=2 also run the two-workgroups-per-CU experiment (measured slower, kept for the record).
//   ./benchmarks/gemm_nt_driver.bin [check|bench|all]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "kernels.h"

#ifdef DPA_WITH_FORM2
namespace dpa {
bool gemm_nt2_ok(int epi, bool grouped);
void gemm_nt2(GemmNtArgs a, int epi, int num_cus, hipStream_t st);
void gemm_nt2_set_skew(int ticks);
}
#endif

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

typedef unsigned short bf16;

static inline bf16 f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u = (u + 0x7fff + ((u >> 16) & 1)) >> 16;
  return (bf16)u;
}

__device__ inline float bf2f(bf16 v) { return __uint_as_float(((unsigned)v) << 16); }

__global__ void fill_kernel(bf16* p, long long n, unsigned seed, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float f = ((h & 0xffffff) / 8388608.0f - 1.0f) * scale;
    __bf16 b = (__bf16)f;
    p[i] = __builtin_bit_cast(bf16, b);
  }
}

// fp32 reference of one epilogue: out[m][n] (fp32), aux[m][n] (fp32, GELU' for epi 1)
__global__ void ref_kernel(const bf16* a, const bf16* b, const bf16* bias, const bf16* aux_in, float* out, float* aux,
                           int M, int N, int K, long long lda, long long ldb, long long ld_aux, int epi, int row_step) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y * row_step;
  if (n >= N || m >= M) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += bf2f(a[(long long)m * lda + k]) * bf2f(b[(long long)n * ldb + k]);
  if (bias) s += bf2f(bias[n]);
  const long long o = (long long)blockIdx.y * N + n;
  if (epi == 1) {
    const float phi = 0.5f * (1.f + erff(s * 0.70710678f));
    out[o] = s * phi;
    aux[o] = phi + s * 0.3989422804f * expf(-0.5f * s * s);
  } else if (epi == 2) {
    out[o] = s * bf2f(aux_in[(long long)m * ld_aux + n]);
  } else if (epi == 3) {
    out[o] = s + bf2f(aux_in[(long long)m * ld_aux + n]);
  } else {
    out[o] = s;
  }
}

static int g_cus = 256;
static int g_form = 1;   // 1: gemm_nt (one workgroup per CU), 2: gemm_nt2 (two per CU)
static void run_gemm(const dpa::GemmNtArgs& g, int epi) {
#ifdef DPA_WITH_FORM2
  if (g_form == 2) {
    dpa::gemm_nt2(g, epi, g_cus, 0);
    return;
  }
#endif
  dpa::gemm_nt(g, epi, g_cus, 0);
}

#ifdef DPA_GEMM_TRACE
namespace dpa { void gemm_nt_read_trace(unsigned long long* out); }
// Per-phase clock stamps of two consecutive output tiles of one workgroup (waves 0 and 4 = the two wave groups).
static void print_trace(int nk) {
  std::vector<unsigned long long> t(8 * 128);
  dpa::gemm_nt_read_trace(t.data());
  for (int w : {0, 4}) {
    const unsigned long long* s = t.data() + w * 128;
    printf("wave %d (ticks of s_memtime; one row per K-tile: the four phase durations)\n", w);
    for (int tile = 0; tile < 2; ++tile) {
      const unsigned long long* b = s + 60 * tile;
      unsigned long long prev = b[58];
      printf(" tile %d: start-of-tile stamp %llu\n", tile, b[58] - s[58]);
      for (int kt = 0; kt < nk; ++kt) {
        printf("   kt %2d:", kt);
        for (int ph = 0; ph < 4; ++ph) {
          printf(" %5llu", b[kt * 4 + ph] - prev);
          prev = b[kt * 4 + ph];
        }
        printf("\n");
      }
      printf("   K loop %llu, issue-ahead %llu, epilogue %llu, tile total %llu\n", b[nk * 4 - 1] - b[58], b[48] - b[nk * 4 - 1],
             b[49] - b[48], b[49] - b[58]);
    }
  }
}
#endif

struct Bufs {
  bf16 *a, *b, *c, *bias, *aux;
  long long na, nb, nc;
};

static double run_case(int M, int N, int K, int epi, bool bias, bool check, int iters) {
  bf16 *a, *b, *c, *bv, *aux;
  const long long na = (long long)M * K, nb = (long long)N * K, nc = (long long)M * N;
  CK(hipMalloc(&a, na * 2));
  CK(hipMalloc(&b, nb * 2));
  CK(hipMalloc(&c, nc * 2));
  CK(hipMalloc(&aux, nc * 2));
  CK(hipMalloc(&bv, (long long)N * 2));
  fill_kernel<<<2048, 256>>>(a, na, 1u, 1.0f);
  fill_kernel<<<2048, 256>>>(b, nb, 2u, 0.05f);
  fill_kernel<<<64, 256>>>(bv, N, 3u, 0.5f);
  fill_kernel<<<2048, 256>>>(aux, nc, 4u, 1.0f);
  CK(hipMemset(c, 0xff, nc * 2));
  dpa::GemmNtArgs g{};
  g.a = a; g.b = b; g.c = c; g.bias = bias ? bv : nullptr;
  g.aux_out = aux; g.aux_in = aux;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ld_aux = N;
  if (!dpa::gemm_nt_ok(M, N, K, K, K, N, N, bias)) {
    printf("shape M=%d N=%d K=%d not supported\n", M, N, K);
    return -1;
  }
  double worst = 0;
  if (check) {
    // reference rows: every row_step-th row
    const int row_step = M > 4096 ? 37 : 1;
    const int nrows = (M + row_step - 1) / row_step;
    float *ro, *ra;
    CK(hipMalloc(&ro, (long long)nrows * N * 4));
    CK(hipMalloc(&ra, (long long)nrows * N * 4));
    ref_kernel<<<dim3((N + 127) / 128, nrows), 128>>>(a, b, bias ? bv : nullptr, aux, ro, ra, M, N, K, K, K, N, epi, row_step);
    CK(hipDeviceSynchronize());
    run_gemm(g, epi);
    CK(hipDeviceSynchronize());
    std::vector<float> hro((size_t)nrows * N), hra((size_t)nrows * N);
    std::vector<bf16> hc((size_t)nc), hx;
    CK(hipMemcpy(hro.data(), ro, hro.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hra.data(), ra, hra.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), c, nc * 2, hipMemcpyDeviceToHost));
    if (epi == 1) {
      hx.resize(nc);
      CK(hipMemcpy(hx.data(), aux, nc * 2, hipMemcpyDeviceToHost));
    }
    long long bad = 0;
    for (int r = 0; r < nrows; ++r) {
      const long long m = (long long)r * row_step;
      for (int n = 0; n < N; ++n) {
        unsigned u = ((unsigned)hc[m * N + n]) << 16;
        float got;
        memcpy(&got, &u, 4);
        const float want = hro[(size_t)r * N + n];
        double err = fabs(got - want) / (fabs(want) + 0.05);
        if (!(err < 2e-2)) {
          if (bad < 5) printf("   mismatch c[%lld][%d] got %g want %g\n", m, n, got, want);
          ++bad;
        }
        if (err > worst) worst = err;
        if (epi == 1) {
          unsigned u2 = ((unsigned)hx[m * N + n]) << 16;
          float g2;
          memcpy(&g2, &u2, 4);
          const float w2 = hra[(size_t)r * N + n];
          double e2 = fabs(g2 - w2) / (fabs(w2) + 0.05);
          if (!(e2 < 2e-2)) {
            if (bad < 5) printf("   mismatch aux[%lld][%d] got %g want %g\n", m, n, g2, w2);
            ++bad;
          }
          if (e2 > worst) worst = e2;
        }
      }
    }
    printf("check form %d M=%d N=%d K=%d epi=%d bias=%d: worst rel err %.3e, %lld bad of %lld  %s\n", g_form, M, N, K, epi, (int)bias,
           worst, bad, (long long)nrows * N, bad ? "FAIL" : "ok");
    fflush(stdout);
    CK(hipFree(ro));
    CK(hipFree(ra));
    if (bad) exit(1);
  }
  double tf = 0;
  if (iters > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) run_gemm(g, epi);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
      CK(hipEventRecord(e0, 0));
      run_gemm(g, epi);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const float med = ts[ts.size() / 2];
    tf = 2.0 * M * N * K / (med * 1e-3) / 1e12;
    printf("bench form %d M=%d N=%d K=%d epi=%d bias=%d: median %.1f us (min %.1f)  %.1f TF\n", g_form, M, N, K, epi, (int)bias,
           med * 1e3, ts[0] * 1e3, tf);
    fflush(stdout);
  }
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c)); CK(hipFree(bv)); CK(hipFree(aux));
  return tf;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, g_cus);
  if (getenv("DPA_FORM")) g_form = atoi(getenv("DPA_FORM"));   // 2: the two-workgroups-per-CU kernel in every mode
  if (!strcmp(mode, "check") || !strcmp(mode, "all")) {
    run_case(256, 256, 128, 0, false, true, 0);
    run_case(512, 512, 192, 0, false, true, 0);
    run_case(1000, 752, 192, 0, true, true, 0);      // row and column tails
    run_case(3000, 1040, 320, 0, false, true, 0);    // several tiles per workgroup? (no: fewer tiles than CUs)
    run_case(40000, 2304, 768, 0, true, true, 0);    // persistent path: many tiles per workgroup, row tail
    run_case(2048, 768, 256, 1, false, true, 0);
    run_case(33000, 3072, 768, 1, false, true, 0);
    run_case(2048, 768, 256, 2, true, true, 0);
    run_case(33000, 3072, 768, 2, false, true, 0);
    run_case(2100, 768, 256, 3, true, true, 0);
    run_case(8192, 50304, 768, 0, false, true, 0);   // lm_head chunk: column tail, tiles_n = 197
  }
  if (!strcmp(mode, "check2")) {   // the two-workgroups-per-CU form against the reference
    g_form = 2;
    run_case(128, 256, 128, 0, false, true, 0);
    run_case(256, 256, 192, 0, false, true, 0);
    run_case(512, 512, 192, 0, false, true, 0);
    run_case(1000, 752, 192, 0, true, true, 0);      // row and column tails
    run_case(3000, 1040, 320, 0, false, true, 0);
    run_case(40000, 2304, 768, 0, true, true, 0);    // persistent path: many tiles per workgroup, row tail
    run_case(2048, 768, 256, 1, false, true, 0);
    run_case(33000, 3072, 768, 1, false, true, 0);
    run_case(2048, 768, 256, 2, true, true, 0);
    run_case(33000, 3072, 768, 2, false, true, 0);
    run_case(2100, 768, 256, 3, true, true, 0);
    run_case(33000, 768, 3072, 3, false, true, 0);
    run_case(8192, 50304, 768, 0, false, true, 0);   // lm_head chunk: column tail, tiles_n = 197
    run_case(4096, 768, 50304, 0, false, true, 0);   // lm_head dx: long K
    run_case(6000, 2048, 1024, 0, false, true, 0);   // nk = 16: not a multiple of the ring's period
  }
  if (!strcmp(mode, "ab")) {   // both forms, interleaved, on the layer shapes (M = argv[2], default 65536)
    const int M = argc > 2 ? atoi(argv[2]) : 65536;
    struct S { int N, K, epi; bool bias; };
    const S shapes[] = {{2304, 768, 0, true}, {768, 768, 3, true}, {3072, 768, 0, false}, {3072, 768, 1, false}, {3072, 768, 2, false},
                        {768, 3072, 3, false}, {768, 3072, 0, false}, {768, 2304, 0, false}, {768, 768, 0, false}};
    for (int rep = 0; rep < 2; ++rep)
      for (const S& s : shapes)
        for (int f : {1, 2}) {
          g_form = f;
          run_case(M, s.N, s.K, s.epi, s.bias, false, 20);
        }
    for (int f : {1, 2}) {
      g_form = f;
      run_case(16384, 50304, 768, 0, false, false, 10);
      run_case(16384, 768, 50304, 0, false, false, 10);
      run_case(4096, 4096, 4096, 0, false, false, 20);
      run_case(8192, 8192, 8192, 0, false, false, 10);
    }
    g_form = 1;
  }
#ifdef DPA_WITH_FORM2
  if (!strcmp(mode, "skew2")) {   // start skew of the second workgroup of each CU (10 ns ticks)
    g_form = 2;
    for (int ticks : {0, 300, 600, 1000, 1500, -1}) {
      printf("---- skew %d ticks\n", ticks);
      dpa::gemm_nt2_set_skew(ticks);
      run_case(65536, 3072, 768, 1, false, false, 20);
      run_case(65536, 3072, 768, 2, false, false, 20);
      run_case(65536, 3072, 768, 0, false, false, 20);
      run_case(65536, 768, 3072, 3, false, false, 20);
    }
    dpa::gemm_nt2_set_skew(-1);
    g_form = 1;
  }
#endif
  if (!strcmp(mode, "variants")) {
    const int M = 65536;
    for (int v : {0, 1, 2, 3}) {
      printf("---- variant %d (bit0: one barrier per phase, bit1: stores dropped)\n", v);
      dpa::gemm_nt_set_variant(v);
      if (v < 2) run_case(33000, 3072, 768, 1, false, true, 0);
      run_case(M, 2304, 768, 0, true, false, 20);
      run_case(M, 3072, 768, 0, false, false, 20);
      run_case(M, 768, 3072, 0, false, false, 20);
      run_case(M, 3072, 768, 1, false, false, 20);
      run_case(M, 3072, 768, 2, false, false, 20);
      run_case(4096, 4096, 4096, 0, false, false, 20);
    }
  }
#ifdef DPA_GEMM_TRACE
  if (!strcmp(mode, "trace")) {   // trace <variant> <N> <K> <epi>
    dpa::gemm_nt_set_variant(argc > 2 ? atoi(argv[2]) : 0);
    const int K = argc > 4 ? atoi(argv[4]) : 768;
    run_case(65536, argc > 3 ? atoi(argv[3]) : 3072, K, argc > 5 ? atoi(argv[5]) : 0, false, false, 5);
    print_trace(K / 64);
  }
#endif
  if (!strcmp(mode, "prof")) {   // prof <variant> <N> <K> <epi>: a few launches of one shape for rocprofv3
    dpa::gemm_nt_set_variant(argc > 2 ? atoi(argv[2]) : 0);
    run_case(65536, argc > 3 ? atoi(argv[3]) : 3072, argc > 4 ? atoi(argv[4]) : 768, argc > 5 ? atoi(argv[5]) : 0, false, false, 5);
  }
  if (!strcmp(mode, "ablate")) {   // timing-only ablations (build with -DDPA_GEMM_ABLATE), stores dropped
    for (int abl : {0, 1, 2, 3, 4, 8, 9, 10, 12}) {
      printf("---- ablation %d (1 no DMA, 2 no reads, 4 no barriers, 8 no MFMA)\n", abl);
      dpa::gemm_nt_set_variant(2 | (abl << 4));
      run_case(65536, 3072, 768, 0, false, false, 10);
      run_case(4096, 4096, 4096, 0, false, false, 10);
    }
  }
  if (!strcmp(mode, "grid")) {   // store cost against the number of workgroups (per-CU or chip-wide limit?)
    for (int cus : {256, 128, 64, 32}) {
      g_cus = cus;
      for (int v : {0, 2}) {
        printf("---- %d workgroups, stores %s\n", cus, v ? "dropped" : "on");
        dpa::gemm_nt_set_variant(v);
        run_case(65536, 3072, 768, 0, false, false, 8);
      }
    }
  }
  if (!strcmp(mode, "quick")) {
    const int M = 65536;
    run_case(4000, 2304, 768, 1, true, true, 0);
    run_case(M, 2304, 768, 0, true, false, 20);
    run_case(M, 3072, 768, 0, false, false, 20);
    run_case(M, 768, 3072, 0, false, false, 20);
    run_case(M, 3072, 768, 1, false, false, 20);
    run_case(M, 3072, 768, 2, false, false, 20);
  }
  if (!strcmp(mode, "gelu")) {   // the GELU epilogue against the plain one, stores on / dropped, 2 us start skew (build with -DDPA_GELU_VARIANT=n)
    const int M = 65536;
    if (argc > 2 && atoi(argv[2])) run_case(33000, 3072, 768, 1, false, true, 0);
    for (int v : {0, 2}) {
      printf("---- stores %s\n", v ? "dropped" : "on");
      dpa::gemm_nt_set_variant(v | (200 << 8));
      for (int rep = 0; rep < 2; ++rep) {
        run_case(M, 3072, 768, 0, false, false, 30);
        run_case(M, 3072, 768, 1, false, false, 30);
      }
    }
    dpa::gemm_nt_set_variant(0);
  }
  if (!strcmp(mode, "aux")) {   // the multiply / add epilogues (auxiliary operand read in the epilogue), 2 us start skew
    const int M = argc > 2 ? atoi(argv[2]) : 65536;
    dpa::gemm_nt_set_variant(200 << 8);
    for (int rep = 0; rep < 2; ++rep) {
      run_case(M, 3072, 768, 2, false, false, 30);
      run_case(M, 768, 768, 3, true, false, 30);
      run_case(M, 768, 3072, 3, false, false, 30);
    }
    dpa::gemm_nt_set_variant(0);
  }
  if (!strcmp(mode, "skew")) {
    const int M = 65536;
    for (int ticks : {0, 200, 400, 600, 800}) {   // 100 MHz ticks per group step: 0 .. 8 us
      printf("---- skew step %d ticks (%.1f us)\n", ticks, ticks / 100.0);
      dpa::gemm_nt_set_variant(ticks << 8);
      run_case(M, 2304, 768, 0, true, false, 20);
      run_case(M, 3072, 768, 0, false, false, 20);
      run_case(M, 3072, 768, 1, false, false, 20);
      run_case(M, 3072, 768, 2, false, false, 20);
    }
  }
  if (!strcmp(mode, "bench") || !strcmp(mode, "all")) {
    const int M = 65536;
    run_case(M, 2304, 768, 0, true, false, 20);
    run_case(M, 768, 768, 0, true, false, 20);
    run_case(M, 3072, 768, 0, false, false, 20);
    run_case(M, 768, 3072, 0, false, false, 20);
    run_case(M, 3072, 768, 1, false, false, 20);
    run_case(M, 3072, 768, 2, false, false, 20);
    run_case(8192, 50304, 768, 0, false, false, 10);
    run_case(8192, 768, 50304, 0, false, false, 10);
    run_case(8192, 8192, 8192, 0, false, false, 10);
    run_case(4096, 4096, 4096, 0, false, false, 20);
  }
  return 0;
}
