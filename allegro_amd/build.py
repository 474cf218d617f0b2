"""Builds the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).

    python -m allegro_amd.build          # -> allegro_amd/liballegro_amd.so
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["aa_gemm.hip", "aa_tp.hip", "aa_tp_spec.hip", "aa_tp_op.hip", "aa_edge.hip", "aa_model.hip", "aa_nl.hip"]
LIB_PATH = os.path.join(HERE, "liballegro_amd.so")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "allegro_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    t0 = time.time()
    if verbose:
        print("[allegro_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    if verbose:
        print(f"[allegro_amd.build] built {LIB_PATH} in {time.time() - t0:.1f}s", flush=True)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
