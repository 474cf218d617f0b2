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
SOURCES = ["aa_gemm.hip", "aa_tp.hip", "aa_tp_spec.hip", "aa_tp_op.hip", "aa_tp_dense.hip", "aa_train.hip", "aa_edge.hip", "aa_fused.hip", "aa_fused8.hip", "aa_chain_res.hip", "aa_model.hip", "aa_nl.hip", "aa_hostfile.hip"]
# public C ABI header: the package carries a copy (package data, so that an installed package can rebuild itself); in the source
# tree that copy is GENERATED from <repo>/include/allegro_amd.h at build time and not tracked (tests/test_lib_symbols.py checks identity)
INCLUDE_DIR = os.path.join(HERE, "include")
LIB_PATH = os.path.join(HERE, "liballegro_amd.so")
# Measured-and-rejected kernels stay out of the product library; AA_BUILD_EXPERIMENTAL=1 adds them (their own opt-in
# switches, their own tests) so that the measurements recorded in DESIGN.md section 9 stay reproducible:
#   aa_fused_bwd.hip  fused per-atom-tile reverse tail (aa_plan_options.fused_tail / AA_FUSED_TAIL=1), section 9.4
EXPERIMENTAL = os.environ.get("AA_BUILD_EXPERIMENTAL", "0")[:1] == "1"
if EXPERIMENTAL:
    SOURCES = SOURCES[:SOURCES.index("aa_model.hip")] + ["aa_fused_bwd.hip"] + SOURCES[SOURCES.index("aa_model.hip"):]
    LIB_PATH = os.path.join(HERE, "liballegro_amd_experimental.so")  # (never overwrites the product library)
TORCH_LIB_PATH = os.path.join(HERE, "liballegro_amd_torch.so")  # dispatcher op for torch.export / AOTI / C++ hosts


def _sync_public_header() -> None:
    """Source tree only: <repo>/include/allegro_amd.h is the canonical C ABI header; the package's copy follows it."""
    src = os.path.join(ROOT, "include", "allegro_amd.h")
    dst = os.path.join(INCLUDE_DIR, "allegro_amd.h")
    if os.path.exists(src):
        with open(src, "rb") as f:
            want = f.read()
        have = open(dst, "rb").read() if os.path.exists(dst) else None
        if have != want:
            os.makedirs(INCLUDE_DIR, exist_ok=True)
            with open(dst, "wb") as f:
                f.write(want)


def _stale() -> bool:
    _sync_public_header()
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _source_deps())


def _source_deps():
    """Files the device library is compiled from (not __pycache__, generators or the host-only torch_ops.cpp)."""
    _sync_public_header()  # (the package's copy of the C ABI header is generated from <repo>/include: not tracked, see .gitignore)
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h") or f in SOURCES]
    return deps + [os.path.join(INCLUDE_DIR, "allegro_amd.h")]


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


OBJ_DIR = os.path.join(HERE, ".objcache")  # per-source objects keyed by content hash (git-ignored, rebuilt on demand)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _object_for(src: str, header_hash: str) -> str:
    import hashlib

    h = hashlib.sha256(header_hash.encode())
    h.update(" ".join(CFLAGS + EXTRA_DEFINES).encode())
    with open(os.path.join(CSRC, src), "rb") as f:
        h.update(f.read())
    return os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}-{h.hexdigest()[:16]}.o")


EXTRA_DEFINES = [d for d in os.environ.get("AA_BUILD_DEFINES", "").split() if d] + (["-DAA_EXPERIMENTAL_TAIL"] if EXPERIMENTAL else [])  # e.g. "-DAA_FUSED_TIMING" (experiments)


def _build_library_locked(verbose: bool) -> str:
    """One `hipcc -c` per translation unit, in parallel, cached by content hash (a source-only edit recompiles one
    file; a header edit all of them), then one link."""
    _sync_public_header()
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hh = hashlib.sha256()
    for d in _source_deps():
        if d.endswith(".h"):
            with open(d, "rb") as f:
                hh.update(os.path.basename(d).encode() + f.read())
    objs = {s: _object_for(s, hh.hexdigest()) for s in SOURCES}
    t0 = time.time()

    def compile_one(s):
        if os.path.exists(objs[s]):
            return
        tmp = objs[s] + f".tmp{os.getpid()}"
        cmd = [hipcc] + CFLAGS + EXTRA_DEFINES + ["-I", INCLUDE_DIR, "-I", CSRC, "-c", os.path.join(CSRC, s), "-o", tmp]
        if verbose:
            print("[allegro_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(tmp, objs[s])

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        list(ex.map(compile_one, SOURCES))
    keep = set(objs.values())
    for f in os.listdir(OBJ_DIR):  # objects of earlier source states
        if os.path.join(OBJ_DIR, f) not in keep and f.endswith(".o"):
            os.remove(os.path.join(OBJ_DIR, f))
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
                   + [objs[s] for s in SOURCES] + ["-o", tmp], check=True)
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
           "-I", INCLUDE_DIR, src, "-o", TORCH_LIB_PATH + f".tmp{os.getpid()}", "-L", HERE, "-lallegro_amd",
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
