#!/usr/bin/env python3
"""Second build of the extension with one kernel source compiled under other flags, for A/B timing
of two builds on one machine:

    python benchmarks/ab_build.py old gemm_nt.hip -DDPA_GEMM_XPOSE=0 -DDPA_GEMM_LATE_BARRIER=0
    DPA_EXT_LIB=benchmarks/ab/_C_old.so python bench.py        # against the default build

Every other object is taken from the regular in-tree build (build/obj), which must be up to date.
"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import build as B  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build(verbose=False)
    out_dir = ROOT / "benchmarks" / "ab"
    out_dir.mkdir(exist_ok=True)
    obj = out_dir / f"{Path(src).stem}_{name}.o"
    hipcc = B._hipcc()
    common = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={B.ARCH}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
              f"-I{B.CSRC}", "-Wno-unused-result", "-Wno-inline-asm"]
    subprocess.run([hipcc, *common, *B.EXTRA_FLAGS.get(src, []), *flags, "-c", str(B.CSRC / src), "-o", str(obj)], check=True)
    stems = [f.stem for f in [*sorted(B.CSRC.glob("*.hip")), *sorted(B.CSRC.glob("*.cpp")), *sorted((B.CSRC / "comm").glob("*.cpp"))]]
    objs = [obj if st == Path(src).stem else B.OBJ / (st + ".o") for st in stems]
    _, torch_lib = B._torch_paths()
    out = out_dir / f"_C_{name}.so"
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", str(out), *map(str, objs), f"-L{torch_lib}",
                    "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip", "-lrccl", f"-Wl,-rpath,{torch_lib}"], check=True)
    print(out.relative_to(ROOT))


if __name__ == "__main__":
    main()
