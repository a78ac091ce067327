"""Builds poly_amd/libpolyhip.so (hand-written HIP for gfx950) in-tree.

    python -m poly_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached next to the sources
(csrc/*.o) and rebuilt when a source or header is newer.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpolyhip.so")
ARCH = "gfx950"

CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
    # Tm parity with Go/amd64, which never fuses multiply-add
    "-ffp-contract=off",
    "-Wall", "-Wno-unused-function",
    # the 256-row register-tiled kernels unroll 64 row groups; past the default 16K the pragma is silently dropped and
    # the H arrays land in scratch (profiles/r01_kernel_resources.md)
    "-mllvm", "-pragma-unroll-threshold=100000",
    "-I", os.path.join(ROOT, "include"), "-I", CSRC,
]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        if force or _newer(o, [s] + hdrs):
            jobs.append([hipcc] + CXXFLAGS + ["-c", s, "-o", o])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose or res.returncode:
                    sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
                if res.returncode:
                    raise RuntimeError(f"hipcc failed on {cmd[-3]}")
    objs = [s[:-4] + ".o" for s in srcs]
    relinked = bool(force or jobs or _newer(LIB, objs))
    # what this call actually did, for the logs (a box that ships prebuilt objects reuses all of them)
    print(f"poly_amd.build: rebuilt {len(jobs)} objects / reused {len(srcs) - len(jobs)}; libpolyhip.so "
          f"{'linked' if relinked else 'reused'}", file=sys.stderr, flush=True)
    if relinked:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
            raise RuntimeError("link of libpolyhip.so failed")
    return LIB


def build_abi_smoke() -> str:
    """tests/abi/abi_smoke: the torch-free C consumer of the ABI (plain gcc, links -lpolyhip only)."""
    src = os.path.join(ROOT, "tests", "abi", "abi_smoke.c")
    exe = os.path.join(ROOT, "tests", "abi", "abi_smoke")
    lib = build_lib()
    if _newer(exe, [src, lib, os.path.join(ROOT, "include", "polyhip.h")]):
        cmd = ["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", HERE, "-lpolyhip", "-lm",
               "-Wl,-rpath,$ORIGIN/../../poly_amd"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
            raise RuntimeError("build of tests/abi/abi_smoke failed")
    return exe


def build_abi_threads() -> str:
    """tests/abi/abi_threads: 8 pthreads in mixed entry points of the C ABI (plain gcc, links -lpolyhip -lpthread)."""
    src = os.path.join(ROOT, "tests", "abi", "abi_threads.c")
    exe = os.path.join(ROOT, "tests", "abi", "abi_threads")
    lib = build_lib()
    if _newer(exe, [src, lib, os.path.join(ROOT, "include", "polyhip.h")]):
        cmd = ["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", HERE, "-lpolyhip", "-lpthread",
               "-lm", "-Wl,-rpath,$ORIGIN/../../poly_amd"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
            raise RuntimeError("build of tests/abi/abi_threads failed")
    return exe


def build_abi_allgather() -> str:
    """tests/abi/abi_allgather: the torch-free multi-rank C host (libpolyhip + the HIP runtime for its device buffers)."""
    src = os.path.join(ROOT, "tests", "abi", "abi_allgather.c")
    exe = os.path.join(ROOT, "tests", "abi", "abi_allgather")
    lib = build_lib()
    if _newer(exe, [src, lib, os.path.join(ROOT, "include", "polyhip.h")]):
        cmd = ["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", src, "-o", exe,
               "-L", HERE, "-lpolyhip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
               "-Wl,-rpath,$ORIGIN/../../poly_amd", "-Wl,-rpath,/opt/rocm/lib"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
            raise RuntimeError("build of tests/abi/abi_allgather failed")
    return exe


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv))
