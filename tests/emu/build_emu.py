"""TEST INFRASTRUCTURE ONLY: builds the product's HIP sources, unmodified, against the CPU emulation
header (tests/emu/include/hip/hip_runtime.h) into tests/emu/_build/liballegro_amd_emu.so.
Used by `pytest -m "not gpu"` to exercise kernel logic without a GPU; never loaded by `allegro_amd`.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "allegro_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liballegro_amd_emu.so")
SOURCES = ["aa_gemm.hip", "aa_tp.hip", "aa_tp_spec.hip", "aa_tp_op.hip", "aa_tp_dense.hip", "aa_train.hip", "aa_edge.hip", "aa_fused.hip", "aa_fused8.hip", "aa_chain_res.hip", "aa_model.hip", "aa_nl.hip", "aa_hostfile.hip"]


EXPERIMENTAL = os.environ.get("AA_BUILD_EXPERIMENTAL", "0")[:1] == "1"  # (see allegro_amd/build.py)
if EXPERIMENTAL:
    SOURCES = SOURCES[:SOURCES.index("aa_model.hip")] + ["aa_fused_bwd.hip"] + SOURCES[SOURCES.index("aa_model.hip"):]
    LIB = os.path.join(OUT_DIR, "liballegro_amd_emu_experimental.so")


EXTRA = [d for d in os.environ.get("AA_EMU_DEFINES", "").split() if d]  # e.g. "-DAA_NO_PROJ_MFMA" (A/B debugging of kernel variants)
if EXTRA:
    import hashlib

    LIB = LIB[:-3] + "_" + hashlib.sha1(" ".join(EXTRA).encode()).hexdigest()[:8] + ".so"


def build_emu(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "include", "hip", "hip_runtime.h"),
        os.path.join(ROOT, "include", "allegro_amd.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    cxx = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-Wno-unused-function", "-Wno-psabi"] + EXTRA + (["-DAA_EXPERIMENTAL_TAIL"] if EXPERIMENTAL else []) + [
           "-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "emu_runtime.cpp"), "-o", LIB]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_emu(force=True))
