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

// Tile numbering of the 256 x 256 weight-gradient kernel: number t of a [tiles_n x tiles_k] grid -> (row tile, column
// tile). Consecutive numbers run through blocks of about 32 tiles (up to 8 column tiles wide), and consecutive
// workgroups land on one XCD, so the 32 tiles an XCD works on at the same time need about 12 operand panels instead
// of the 33 of a row-major run through a wide grid (dW[2048, 8192]: 8 x 32 tiles).
DPA_HD void wgrad_tile_coord(int t, int tiles_n, int tiles_k, int& tn, int& tk) {
  const int bw = tiles_k < 8 ? tiles_k : 8;
  int bh = 32 / bw;
  if (bh < 1) bh = 1;
  const int rowblk = bh * tiles_k;               // tiles of a full row of blocks
  const int br = t / rowblk;
  const int r = t - br * rowblk;
  int h = tiles_n - br * bh;                     // tile rows of this row of blocks
  if (h > bh) h = bh;
  const int blk = h * bw;
  const int nfull = tiles_k / bw;
  int bc = r / blk, q = r - bc * blk, w = bw;
  if (bc >= nfull) {
    bc = nfull;
    q = r - nfull * blk;
    w = tiles_k - nfull * bw;
  }
  tn = br * bh + q / w;
  tk = bc * bw + q % w;
}

// Work items of the accumulating weight-gradient kernel (one 256 x 256 output tile of dW += dY^T X per item, one
// workgroup per CU). Workgroup b runs on XCD b % 8, which owns the contiguous tile range [x T/8, (x+1) T/8): tiles
// that are neighbours in the (row-tile major) numbering share their dY columns, so the workgroups of an XCD that
// run at the same time pull shared operand panels through that XCD's L2. Whole rounds of `cus` tiles sum over all
// rows; what is left of the range (fewer tiles than CUs) is cut into `parts` row ranges per tile, so that the last,
// shorter round fills the XCD too instead of leaving most of it idle for a whole pass over the rows (591 tiles of
// the 50304 x 768 head gradient: 2.33 rounds instead of 3). A tile summed in parts is added with float atomics.
struct WgradItem {
  int tile, part, parts;   // parts == 0: no work for this workgroup
};
struct WgradWalk {
  int tiles = 0;
  int cus = 32;            // CUs of an XCD
  int max_parts = 1;       // bound from the row count (a part keeps at least 2048 rows)
  DPA_HD void range(int x, int& lo, int& full, int& rest, int& parts) const {
    lo = (int)((long long)x * tiles / 8);
    const int hi = (int)((long long)(x + 1) * tiles / 8);
    full = (hi - lo) / cus * cus;
    rest = hi - lo - full;
    parts = rest > 0 ? cus / rest : 1;
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
  }
  DPA_HD int items(int x) const {
    int lo, full, rest, parts;
    range(x, lo, full, rest, parts);
    return full + rest * parts;
  }
  DPA_HD int grid() const {
    int m = 0;
    for (int x = 0; x < 8; ++x) {
      const int n = items(x);
      if (n > m) m = n;
    }
    return 8 * m;
  }
  DPA_HD WgradItem at(int block) const {
    int lo, full, rest, parts;
    range(block & 7, lo, full, rest, parts);
    const int j = block >> 3;
    WgradItem it;
    if (j < full) {
      it.tile = lo + j; it.part = 0; it.parts = 1;
    } else if (j - full < rest * parts) {
      it.tile = lo + full + (j - full) % rest; it.part = (j - full) / rest; it.parts = parts;
    } else {
      it.tile = 0; it.part = 0; it.parts = 0;
    }
    return it;
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
// stride (in elements) among the operands addressed by row offset: c and the auxiliary tensors (the kernel
// addresses a from the first row of each output tile, so a's height and stride do not enter).
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
    long long ldm = ldc;     // a is addressed per output tile, whatever its height
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
