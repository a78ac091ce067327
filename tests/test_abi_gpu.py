"""The C ABI as the cgo shim sees it: tests/abi/abi_smoke.c is a plain C program linked against libpolyhip.so
only.  It runs in its own process, without Python and with nothing of PyTorch on a library path, so the HIP
runtime it gets is /opt/rocm's -- every other test here maps the copy bundled with the torch wheel first
(poly_amd/_lib.py).  Values are the reference's own test expectations (see the file header)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_harness_without_torch():
    from poly_amd import build
    exe = build.build_abi_smoke()
    env = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "LD_PRELOAD", "PYTHONPATH")}
    lp = [p for p in os.environ.get("LD_LIBRARY_PATH", "").split(":") if p and "torch" not in p and "site-packages" not in p]
    if lp:
        env["LD_LIBRARY_PATH"] = ":".join(lp)
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "puc19.seq")], env=env, capture_output=True, text=True,
                         timeout=300)
    print(res.stdout, res.stderr)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "abi_smoke: ok" in res.stdout
    rt = [ln for ln in res.stdout.splitlines() if "HIP runtime" in ln]
    assert rt and "torch" not in rt[0] and "/opt/rocm" in rt[0], rt


def test_eight_threads_in_mixed_entry_points():
    """tests/abi/abi_threads.c: cgo calls arrive on arbitrary OS threads, several at once -- 8 pthreads x 6 rounds x 7
    host-pointer entry points (one scoring handle shared by all), every result equal to the serial run"""
    from poly_amd import build
    exe = build.build_abi_threads()
    res = subprocess.run([exe, "8", "6"], capture_output=True, text=True, timeout=600)
    print(res.stdout, res.stderr)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "abi_threads ok: 8 threads" in res.stdout
