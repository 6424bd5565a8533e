"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only, never on a GPU box).

The token-file reader and the launch-side logic of the kernels (shape admission, the persistent tile
walk) live in headers without GPU or tensor-library dependencies, so they build with plain g++. The
sanitizer runtime is linked statically: it then does not depend on its position in the initial
library list of the process.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
         "-fno-sanitize-recover=all", "-static-libasan", "-static-libubsan", "-pthread"]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_logic_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_checks"
    build = subprocess.run(["g++", *FLAGS, "-I", os.path.join(ROOT, "csrc"),
                            os.path.join(ROOT, "tests", "native", "host_checks.cpp"), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), str(scratch)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "host checks passed" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_sanitizer_catches_a_planted_overrun(tmp_path):
    """The harness itself works: a one-element overrun of an exact-size heap buffer is reported."""
    src = tmp_path / "planted.cpp"
    src.write_text('#include "token_file.h"\n#include <cstdio>\n#include <vector>\n'
                   "int main(int, char** argv) {\n"
                   "  std::vector<uint16_t> tok(64, 3);\n"
                   '  FILE* f = std::fopen(argv[1], "wb"); std::fwrite(tok.data(), 2, tok.size(), f); std::fclose(f);\n'
                   "  dpa::TokenStream ts(argv[1]);\n"
                   "  std::vector<int64_t> x(2 * 8 - 1), y(2 * 8);\n"   # x is one element short
                   "  int64_t starts[2] = {0, 10};\n"
                   "  ts.gather(starts, 2, 8, x.data(), y.data(), 1);\n"
                   "  return 0;\n}\n")
    exe = tmp_path / "planted"
    build = subprocess.run(["g++", *FLAGS, "-I", os.path.join(ROOT, "csrc"), str(src), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    run = subprocess.run([str(exe), str(tmp_path / "t.bin")], capture_output=True, text=True, timeout=120)
    assert run.returncode != 0 and "heap-buffer-overflow" in run.stderr
