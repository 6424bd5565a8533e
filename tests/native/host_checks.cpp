// Host-side checks meant to run under AddressSanitizer + UBSan (built by tests/test_native_sanitized.py
// with g++; never on a GPU box). Covers the token-file reader and the launch-side logic of the NT GEMM.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "host_logic.h"
#include "token_file.h"

static int g_failed = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) {                                                      \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
      ++g_failed;                                                       \
    }                                                                   \
  } while (0)

template <class F>
static bool throws(F&& f) {
  try {
    f();
  } catch (const std::exception&) {
    return true;
  }
  return false;
}

static std::string write_tokens(const std::string& dir, const char* name, const std::vector<uint16_t>& v, size_t extra_bytes = 0) {
  const std::string path = dir + "/" + name;
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) std::abort();
  if (!v.empty()) std::fwrite(v.data(), 2, v.size(), f);
  for (size_t i = 0; i < extra_bytes; ++i) std::fputc(7, f);
  std::fclose(f);
  return path;
}

static void test_token_stream(const std::string& dir) {
  std::vector<uint16_t> tok(10007);
  for (size_t i = 0; i < tok.size(); ++i) tok[i] = (uint16_t)((i * 2654435761u) >> 7);
  const std::string path = write_tokens(dir, "tokens.bin", tok);
  dpa::TokenStream ts(path);
  CHECK(ts.size() == (int64_t)tok.size());

  const int64_t T = 33;
  std::mt19937_64 rng(5);
  for (int64_t B : {int64_t(1), int64_t(7), int64_t(64)}) {
    for (int64_t threads : {int64_t(1), int64_t(3), int64_t(16), int64_t(200)}) {
      std::vector<int64_t> starts(B);
      for (auto& s : starts) s = (int64_t)(rng() % (uint64_t)(ts.size() - T));
      starts[0] = ts.size() - (T + 1);   // the last window that fits, to the byte
      if (B > 1) starts[1] = 0;
      // exact-size heap buffers: one element past either end is an ASan report
      std::vector<int64_t> x(B * T), y(B * T);
      ts.gather(starts.data(), B, T, x.data(), y.data(), threads);
      bool same = true;
      for (int64_t i = 0; i < B && same; ++i)
        for (int64_t t = 0; t < T; ++t)
          same = same && x[i * T + t] == tok[starts[i] + t] && y[i * T + t] == tok[starts[i] + t + 1];
      CHECK(same);
    }
  }
  // windows that leave the file are refused before anything is written
  std::vector<int64_t> x(2 * T, -1), y(2 * T, -1);
  for (int64_t bad : {ts.size() - T, ts.size(), int64_t(-1), INT64_MAX, INT64_MIN}) {
    int64_t starts[2] = {0, bad};
    CHECK(throws([&] { ts.gather(starts, 2, T, x.data(), y.data(), 2); }));
    CHECK(x[0] == -1 && y[2 * T - 1] == -1);
  }
  int64_t zero = 0;
  CHECK(throws([&] { ts.gather(&zero, 1, 0, x.data(), y.data(), 1); }));
  CHECK(throws([&] { ts.gather(&zero, 1, INT64_MAX, x.data(), y.data(), 1); }));
  ts.gather(&zero, 0, T, x.data(), y.data(), 4);   // empty batch: nothing to do
  CHECK(x[0] == -1);

  // malformed files: missing, empty, odd byte count (each must also release its descriptor)
  CHECK(throws([&] { dpa::TokenStream t(dir + "/nope.bin"); }));
  const std::string empty = write_tokens(dir, "empty.bin", {});
  const std::string odd = write_tokens(dir, "odd.bin", {1, 2, 3}, 1);
  for (int i = 0; i < 2000; ++i) {   // would run out of descriptors if a failed open leaked one
    CHECK(throws([&] { dpa::TokenStream t(empty); }));
    CHECK(throws([&] { dpa::TokenStream t(odd); }));
  }
}

static void test_gemm_walk() {
  // every (workgroup, iteration) pair maps to a distinct tile inside the grid and all tiles are covered, for the
  // 256-row tiles of the one-workgroup-per-CU kernel and the 128-row tiles of the two-per-CU kernel
  struct Form { int tile_h, group_m, wg_per_cu; };
  for (const Form f : {Form{256, 4, 1}, Form{128, 8, 2}})
  for (int tm : {1, 2, 3, 7, 8, 9, 12, 31, 32, 33, 64, 100, 256, 513})
    for (int tn : {1, 2, 3, 9, 12, 197})
      for (int cus : {0, 8, 64, 104, 250, 256, 304}) {
        const int nwg = dpa::gemm_nt_grid(tm, tn, cus, f.wg_per_cu);
        CHECK(nwg >= 1 && (nwg % 8 == 0 || nwg == tm * tn));
        dpa::GemmWalk w{tm, tn, nwg, f.tile_h, f.group_m};
        std::vector<char> seen((size_t)tm * tn, 0);
        long long visited = 0;
        for (int wg = 0; wg < nwg; ++wg) {
          const int c = w.count(wg);
          CHECK(c >= 0);
          for (int it = 0; it < c; ++it) {
            const dpa::TileCoord t = w.at(wg, it);
            CHECK(t.m0 % f.tile_h == 0 && t.n0 % 256 == 0);
            const int mi = t.m0 / f.tile_h, ni = t.n0 / 256;
            CHECK(mi >= 0 && mi < tm && ni >= 0 && ni < tn);
            if (mi >= 0 && mi < tm && ni >= 0 && ni < tn) {
              CHECK(!seen[(size_t)mi * tn + ni]);
              seen[(size_t)mi * tn + ni] = 1;
            }
            ++visited;
          }
        }
        CHECK(visited == (long long)tm * tn);
      }
}

static void test_wgrad_tile_coord() {
  // a bijection onto the tile grid, whatever the grid's shape
  for (int tn : {1, 2, 3, 4, 5, 8, 9, 12, 24, 32, 33, 197})
    for (int tk : {1, 2, 3, 7, 8, 9, 12, 16, 31, 32, 33}) {
      std::vector<char> seen((size_t)tn * tk, 0);
      for (int t = 0; t < tn * tk; ++t) {
        int a = -1, b = -1;
        dpa::wgrad_tile_coord(t, tn, tk, a, b);
        CHECK(a >= 0 && a < tn && b >= 0 && b < tk);
        if (a >= 0 && a < tn && b >= 0 && b < tk) {
          CHECK(!seen[(size_t)a * tk + b]);
          seen[(size_t)a * tk + b] = 1;
        }
      }
    }
  // narrow grids keep the row-major order; an 8 x 32 grid is cut into 4 x 8 blocks
  int a, b;
  dpa::wgrad_tile_coord(7, 12, 3, a, b);
  CHECK(a == 2 && b == 1);
  dpa::wgrad_tile_coord(31, 8, 32, a, b);
  CHECK(a == 3 && b == 7);
  dpa::wgrad_tile_coord(32, 8, 32, a, b);
  CHECK(a == 0 && b == 8);
}

static void test_wgrad_walk() {
  // every (tile, part) of the accumulating weight-gradient walk is visited exactly once, parts of a tile agree on
  // their count, and a tile summed in parts never also has a whole-rows item
  for (int tiles : {1, 3, 7, 8, 9, 36, 192, 255, 256, 257, 591, 1024, 1031})
    for (int max_parts : {1, 2, 8})
      for (int cus : {4, 32}) {
        dpa::WgradWalk w;
        w.tiles = tiles;
        w.cus = cus;
        w.max_parts = max_parts;
        const int grid = w.grid();
        CHECK(grid % 8 == 0 && grid >= tiles);
        std::vector<int> parts(tiles, 0), seen(tiles, 0);
        std::vector<std::vector<char>> hit(tiles);
        for (int b = 0; b < grid; ++b) {
          const dpa::WgradItem it = w.at(b);
          if (it.parts == 0) continue;
          CHECK(it.tile >= 0 && it.tile < tiles && it.parts >= 1 && it.parts <= max_parts && it.part >= 0 && it.part < it.parts);
          if (it.tile < 0 || it.tile >= tiles) continue;
          if (parts[it.tile] == 0) {
            parts[it.tile] = it.parts;
            hit[it.tile].assign(it.parts, 0);
          }
          CHECK(parts[it.tile] == it.parts);
          if (parts[it.tile] == it.parts && it.part < it.parts) {
            CHECK(!hit[it.tile][it.part]);
            hit[it.tile][it.part] = 1;
            ++seen[it.tile];
          }
        }
        for (int t = 0; t < tiles; ++t) CHECK(parts[t] >= 1 && seen[t] == parts[t]);
      }
  // the head gradient of GPT-2 (197 x 3 tiles): two whole rounds and a third of thirds on every XCD
  dpa::WgradWalk w;
  w.tiles = 591;
  w.max_parts = 8;
  CHECK(w.grid() == 8 * 94);
  for (int x = 0; x < 8; ++x) {
    int lo, full, rest, parts;
    w.range(x, lo, full, rest, parts);
    CHECK(full == 64 && (rest == 9 || rest == 10) && parts == 3);
  }
}

static void test_gemm_shapes() {
  using dpa::gemm_nt_shape_ok;
  CHECK(gemm_nt_shape_ok(65536, 2304, 768, 768, 768, 2304, 0, true));
  CHECK(gemm_nt_shape_ok(65536, 768, 3072, 3072, 3072, 768, 768, false));
  CHECK(gemm_nt_shape_ok(8192, 50304, 768, 768, 768, 50304, 0, false));
  CHECK(!gemm_nt_shape_ok(8192, 50304, 768, 768, 768, 50304, 0, true));    // bias row would not fit in LDS
  CHECK(gemm_nt_shape_ok(65536, 50304, 768, 768, 768, 50304, 0, false));   // C passes 2 GiB: walked in row ranges
  CHECK(dpa::gemm_nt_row_chunk(65536, 50304, 256) == 21248);               // (2^31 - 1) / (50304 * 2), down to 256 rows
  CHECK(dpa::gemm_nt_row_chunk(16384, 50304, 256) == 16384);               // fits: one launch
  CHECK(dpa::gemm_nt_row_chunk(262144, 4096, 1024) == 261120);             // rotary forms cut at whole sequences
  CHECK(21248ll * 50304 * 2 < 0x80000000ll && 261120ll * 4096 * 2 < 0x80000000ll);
  CHECK(dpa::gemm_nt_pair_shape_ok(262144, 2048, 768, 768, 768, 2048, 4096));   // [gate | up] rows of 8 KiB, 2 GiB in all
  CHECK(!gemm_nt_shape_ok(64, 768, 96, 96, 96, 768, 0, false));            // K not a multiple of 64
  CHECK(!gemm_nt_shape_ok(64, 768, 64, 64, 64, 768, 0, false));            // K below two K-tiles
  CHECK(!gemm_nt_shape_ok(64, 760, 768, 768, 768, 760, 0, false));         // N not a multiple of 16
  CHECK(!gemm_nt_shape_ok(64, 768, 768, 772, 768, 768, 0, false));         // row pitch not 16-byte aligned
  CHECK(!gemm_nt_shape_ok(64, 768, 768, 512, 768, 768, 0, false));         // pitch shorter than a row
  CHECK(!gemm_nt_shape_ok(64, 768, 768, 768, 768, 768, 512, false));       // aux pitch shorter than a row
  CHECK(!gemm_nt_shape_ok(0, 768, 768, 768, 768, 768, 0, false));
  // extreme values must be refused without signed overflow (UBSan watches the products)
  const long long big = 0x7fffffffffffffffll;
  CHECK(!gemm_nt_shape_ok(big, 768, 768, 768, 768, 768, 0, false));
  CHECK(!gemm_nt_shape_ok(64, 768, 768, big - 7, 768, 768, 0, false));
  CHECK(!gemm_nt_shape_ok(64, big - 15, 768, 768, 768, big - 15, 0, false));
  CHECK(!gemm_nt_shape_ok(64, 768, 768, 768, 768, 768, big - 7, false));
}

static void test_gemm_pair_shapes() {
  using dpa::gemm_nt_pair_shape_ok;
  // N gate columns; [M, N] activation with row stride ld_act, [M, 2N] pre-activation with row stride ld_pair
  CHECK(gemm_nt_pair_shape_ok(65536, 3072, 768, 768, 768, 3072, 6144));
  CHECK(gemm_nt_pair_shape_ok(2100, 1088, 128, 128, 128, 1088, 2176));
  CHECK(!gemm_nt_pair_shape_ok(2100, 1080, 128, 128, 128, 1080, 2160));       // N not a multiple of 16
  CHECK(!gemm_nt_pair_shape_ok(2100, 1088, 128, 128, 128, 1088, 2000));       // pair pitch shorter than 2N
  CHECK(!gemm_nt_pair_shape_ok(2100, 1088, 128, 128, 128, 1000, 2176));       // activation pitch shorter than N
  CHECK(gemm_nt_pair_shape_ok(262144, 4096, 768, 768, 768, 4096, 8192));      // [M, 2N] passes 2 GiB: row ranges
  CHECK(!gemm_nt_pair_shape_ok(2100, 1088, 128, 128, 128, 4096, 2176));       // activation pitch wider than the pair's
  const long long big = 0x7fffffffffffffffll;
  CHECK(!gemm_nt_pair_shape_ok(64, big - 15, 768, 768, 768, big - 15, big - 15));
  CHECK(!gemm_nt_pair_shape_ok(64, (big / 2) & ~15ll, 768, 768, 768, 768, 768));
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: host_checks <scratch dir>\n");
    return 2;
  }
  test_token_stream(argv[1]);
  test_gemm_walk();
  test_wgrad_tile_coord();
  test_wgrad_walk();
  test_gemm_shapes();
  test_gemm_pair_shapes();
  if (g_failed) {
    std::fprintf(stderr, "%d check(s) failed\n", g_failed);
    return 1;
  }
  std::puts("host checks passed");
  return 0;
}
