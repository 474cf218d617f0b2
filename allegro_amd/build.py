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
SOURCES = ["aa_gemm.hip", "aa_tp.hip", "aa_tp_spec.hip", "aa_tp_op.hip", "aa_edge.hip", "aa_fused.hip", "aa_fused16.hip", "aa_tp_mfma.hip", "aa_model.hip", "aa_nl.hip"]
LIB_PATH = os.path.join(HERE, "liballegro_amd.so")
TORCH_LIB_PATH = os.path.join(HERE, "liballegro_amd_torch.so")  # dispatcher op for torch.export / AOTI / C++ hosts


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _source_deps())


def _source_deps():
    """Files the device library is compiled from (not __pycache__, generators or the host-only torch_ops.cpp)."""
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    return deps + [os.path.join(ROOT, "include", "allegro_amd.h")]


def source_hash() -> str:
    """sha256 over the device sources: identifies the kernels a profile / PMC measurement was taken on
    (profiles/pmc_traffic_*.json carry it; bench.py attaches measured traffic only when it matches)."""
    import hashlib

    h = hashlib.sha256()
    for d in _source_deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


class _BuildLock:
    """Serialises concurrent builders (the ranks of a multi-GPU launch all import the package at once)."""

    def __enter__(self):
        import fcntl

        self._f = open(os.path.join(HERE, ".build.lock"), "w")
        fcntl.flock(self._f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl

        fcntl.flock(self._f, fcntl.LOCK_UN)
        self._f.close()


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB_PATH
    with _BuildLock():
        if not force and not _stale():  # another process built it while we waited
            return LIB_PATH
        return _build_library_locked(verbose)


def _build_library_locked(verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    t0 = time.time()
    if verbose:
        print("[allegro_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB_PATH)  # atomic: a concurrently starting process never maps a half-written library
    if verbose:
        print(f"[allegro_amd.build] built {LIB_PATH} in {time.time() - t0:.1f}s", flush=True)
    return LIB_PATH


def build_torch_ops(force: bool = False, verbose: bool = True) -> str:
    """csrc/torch_ops.cpp: plain host C++ (no device code) registering `allegro_amd_native::energy_forces` with the
    PyTorch dispatcher; links against liballegro_amd.so next to it."""
    src = os.path.join(CSRC, "torch_ops.cpp")
    build_library(verbose=verbose)

    def fresh():
        return (os.path.exists(TORCH_LIB_PATH) and
                os.path.getmtime(TORCH_LIB_PATH) > max(os.path.getmtime(src), os.path.getmtime(LIB_PATH)))

    if not force and fresh():
        return TORCH_LIB_PATH
    with _BuildLock():  # ranks importing at once: one builds, the others wait and find it fresh
        if not force and fresh():
            return TORCH_LIB_PATH
        return _build_torch_ops_locked(src, verbose)


def _build_torch_ops_locked(src: str, verbose: bool) -> str:
    import torch

    tdir = os.path.dirname(torch.__file__)
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-I", os.path.join(tdir, "include"),
           "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-I", "/opt/rocm/include",
           "-I", os.path.join(ROOT, "include"), src, "-o", TORCH_LIB_PATH + f".tmp{os.getpid()}", "-L", HERE, "-lallegro_amd",
           "-Wl,-rpath,$ORIGIN", "-L", os.path.join(tdir, "lib"), "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip",
           "-ltorch_hip", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    t0 = time.time()
    if verbose:
        print("[allegro_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(TORCH_LIB_PATH + f".tmp{os.getpid()}", TORCH_LIB_PATH)  # atomic, like the device library
    if verbose:
        print(f"[allegro_amd.build] built {TORCH_LIB_PATH} in {time.time() - t0:.1f}s", flush=True)
    return TORCH_LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    build_torch_ops(force="--force" in sys.argv)
