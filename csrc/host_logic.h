// Host-checkable logic shared by the kernels and their launchers: shape admission rules and the
// persistent tile walk. Plain C++ (no HIP headers) so that the same code also builds with g++ under
// AddressSanitizer / UBSan in tests/native/.
#pragma once

#if defined(__HIPCC__)
#define DPA_HD __host__ __device__ __forceinline__
#else
#define DPA_HD inline
#endif

namespace dpa {

constexpr int kGemmNtMaxBiasN = 8192;   // bf16 bias row kept in LDS: 16 KiB

struct TileCoord {
  int m0, n0;
};

// Tile walk of the persistent NT GEMM. Workgroups are dealt round-robin to the 8 XCDs (each with its
// own L2): the m-tiles are split into 8 contiguous ranges, one per XCD label (blockIdx % 8), and inside
// a range the tiles that run at the same time form blocks of 4 m-tiles x all n-tiles, so that
// concurrent workgroups of an XCD share A and B panels through that L2. Any (blockIdx, iteration) ->
// tile bijection is correct; this one is for locality.
struct GemmWalk {
  int tiles_m, tiles_n, nwg;
  int tile_h = 256;   // rows of an output tile (columns are always 256): 128 for the two-workgroups-per-CU kernel
  int group_m = 4;    // m-tiles of a block of tiles that run at the same time on one XCD
  DPA_HD bool xcd_mode() const { return tiles_m >= 8 && (nwg & 7) == 0; }
  DPA_HD int count(int wg) const {
    if (!xcd_mode()) {
      const int total = tiles_m * tiles_n;
      return wg < total ? (total - wg + nwg - 1) / nwg : 0;
    }
    const int x = wg & 7, slot = wg >> 3, per = nwg >> 3;
    const int lo = x * tiles_m / 8, hi = (x + 1) * tiles_m / 8;
    const int total = (hi - lo) * tiles_n;
    return slot < total ? (total - slot + per - 1) / per : 0;
  }
  DPA_HD TileCoord at(int wg, int it) const {
    TileCoord t;
    if (!xcd_mode()) {
      const int s = wg + it * nwg;
      t.m0 = (s % tiles_m) * tile_h;
      t.n0 = (s / tiles_m) * 256;
      return t;
    }
    const int x = wg & 7, slot = wg >> 3, per = nwg >> 3;
    const int lo = x * tiles_m / 8, hi = (x + 1) * tiles_m / 8;
    const int s = slot + it * per;
    const int gsz = group_m * tiles_n;
    const int nfull = (hi - lo) / group_m;
    int g = s / gsz, r = s % gsz, gm = group_m;
    if (g >= nfull) {
      g = nfull;
      r = s - nfull * gsz;
      gm = (hi - lo) - group_m * nfull;
    }
    t.n0 = (r / gm) * 256;
    t.m0 = (lo + group_m * g + r % gm) * tile_h;
    return t;
  }
};

// Number of workgroups the NT GEMM launches for a tile grid on a part with `num_cus` compute units.
inline int gemm_nt_grid(int tiles_m, int tiles_n, int num_cus, int wg_per_cu = 1) {
  const int total = tiles_m * tiles_n;
  int grid = (num_cus > 0 ? num_cus : 256) * wg_per_cu;
  grid &= ~7;
  if (grid < 8) grid = 8;
  if (tiles_m < 8 && grid > total) grid = total;
  return grid;
}

// Rows of one launch of the NT GEMM. Every byte offset the kernel forms must stay below 2 GiB: offsets at or
// above that are its "outside every buffer" range, used to turn loads into zeros and stores into no-ops. The
// rows of an NT product are independent, so a taller problem runs as several launches over row ranges of at
// most this many rows (a multiple of `align`; 0 if not even `align` rows fit). `ld_max` is the widest row
// stride (in elements) among the row-indexed operands.
inline long long gemm_nt_row_chunk(long long M, long long ld_max, long long align) {
  const long long lim = 0x80000000ll;
  if (M * ld_max * 2 < lim) return M;
  return (lim - 1) / (ld_max * 2) / align * align;
}

// Shapes the NT GEMM accepts (any number of rows: see gemm_nt_row_chunk).
inline bool gemm_nt_shape_ok(long long M, long long N, long long K, long long lda, long long ldb, long long ldc,
                             long long ld_aux, bool has_bias) {
  const long long lim = 0x80000000ll;
  if (M < 1 || N < 16 || K < 128 || K % 64 != 0 || N % 16 != 0) return false;
  if (lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || ld_aux % 8 != 0) return false;
  if (lda < K || ldb < K || ldc < N || (ld_aux != 0 && ld_aux < N)) return false;
  if (M >= lim || N >= lim || lda >= lim || ldb >= lim || ldc >= lim || ld_aux >= lim) return false;
  if (N * ldb * 2 >= lim) return false;
  {
    long long ldm = lda > ldc ? lda : ldc;
    if (ld_aux > ldm) ldm = ld_aux;
    if (gemm_nt_row_chunk(M, ldm, 256) < 256 && M * ldm * 2 >= lim) return false;
  }
  if (256 * lda * 2 >= (1ll << 30) || 256 * ldb * 2 >= (1ll << 30)) return false;
  if (has_bias && N > kGemmNtMaxBiasN) return false;
  return true;
}

// The SwiGLU forms of the NT GEMM (paired gate / up columns): N gate columns, an [M, N] tensor with row stride
// ld_act and an [M, 2N] tensor with row stride ld_pair; the weight has 2N rows in the forward form and N in the
// backward form (the larger is checked).
inline bool gemm_nt_pair_shape_ok(long long M, long long N, long long K, long long lda, long long ldb, long long ld_act,
                                  long long ld_pair) {
  const long long lim = 0x80000000ll;
  if (N < 16 || N >= lim / 2 || N % 16 != 0) return false;
  if (!gemm_nt_shape_ok(M, 2 * N, K, lda, ldb, ld_pair, ld_pair, false)) return false;
  if (ld_act % 8 != 0 || ld_act < N || ld_act >= lim || ld_act > ld_pair) return false;   // ld_pair bounds the row chunk
  return true;
}

}  // namespace dpa
