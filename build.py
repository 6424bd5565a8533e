#!/usr/bin/env python3
"""Build the gfx950 extension in-tree with hipcc.

    python build.py            # incremental
    python build.py --force    # rebuild everything
    python build.py --asm      # also keep the per-kernel ISA + resource usage under build/asm

Kernel sources (csrc/*.hip) are PyTorch-free and compile in seconds; csrc/bindings.cpp is the
only translation unit that includes PyTorch headers. The result is
distributed_pytorch_amd/_C.so, loaded with torch.ops.load_library (no Python C-API involved).
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build" / "obj"
OUT = ROOT / "distributed_pytorch_amd" / "_C.so"
ARCH = "gfx950"
# Per-file code generation options. Keeping MFMA accumulators in architectural VGPRs avoids the
# v_accvgpr_read/write traffic that VALU work on accumulators (softmax rescale, dS) otherwise pays.
EXTRA_FLAGS = {
    "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}
# Sources that textually include another source (rebuilt when that one changes).
EXTRA_DEPS = {
    "attention_wide.hip": ["attention.hip"],
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _torch_paths():
    import torch
    from torch.utils import cpp_extension as ce

    return ce.include_paths(), os.path.join(os.path.dirname(torch.__file__), "lib")


def _stale(src: Path, obj: Path, deps) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, *deps])


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[-1]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build(force: bool = False, asm: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    headers = sorted(CSRC.glob("*.h")) + [Path(__file__).resolve()]
    incs, torch_lib = _torch_paths()
    common = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
              f"-I{CSRC}", "-Wno-unused-result",
              "-Wno-inline-asm"]    # (the LDS-DMA pieces write M0 and say so in their clobber lists; clang remarks on every one)
    jobs = []
    objs = []
    for src in sorted(CSRC.glob("*.hip")):
        obj = OBJ / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(src, obj, headers + [CSRC / d for d in EXTRA_DEPS.get(src.name, [])]):
            cmd = [hipcc, *common, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
            if asm:
                (ROOT / "build" / "asm").mkdir(parents=True, exist_ok=True)
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append(cmd)
    # translation units that include PyTorch headers (csrc/comm: the RCCL layer)
    for bsrc in sorted(CSRC.glob("*.cpp")) + sorted((CSRC / "comm").glob("*.cpp")):
        bobj = OBJ / (bsrc.stem + ".o")
        objs.append(bobj)
        if force or _stale(bsrc, bobj, headers):
            cmd = [hipcc, *common, "-x", "hip", "-DTORCH_EXTENSION_NAME=_C", "-D_GLIBCXX_USE_CXX11_ABI=1"]
            for i in incs:
                cmd += ["-isystem", i]
            cmd += ["-c", str(bsrc), "-o", str(bobj)]
            jobs.append(cmd)
    if jobs:
        if verbose:
            print(f"[build] compiling {len(jobs)} translation unit(s) for {ARCH}", flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if jobs or not OUT.exists() or force:
        link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(OUT), *map(str, objs),
                f"-L{torch_lib}", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
                "-lrccl",   # PyTorch's own copy (same soname as the system one): one RCCL per process
                f"-Wl,-rpath,{torch_lib}"]
        _run(link)
        if verbose:
            print(f"[build] linked {OUT.relative_to(ROOT)}", flush=True)
    elif verbose:
        print("[build] up to date", flush=True)
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--asm", action="store_true")
    a = ap.parse_args()
    build(force=a.force, asm=a.asm)
