"""CPU-only checks of the drop-in boundary: libpolyhip.so builds for gfx950,
loads without a GPU, exports every symbol include/polyhip.h declares, and
refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "polyhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(polyhip_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    from poly_amd import build, _lib
    path = build.build_lib()
    assert os.path.exists(path)
    L = C.CDLL(path)
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(L, n), f"{n} declared in polyhip.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().polyhip_abi_version() == 1


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under poly_amd/ may load or link it."""
    pkg = os.path.join(ROOT, "poly_amd")
    banned = re.compile(r"import\s+oracle|from\s+oracle|libpolyoracle|poly_oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not banned.search(txt), f"{os.path.join(dirpath, f)} references the oracle"


def test_argument_errors_without_gpu():
    from poly_amd import _lib, mash
    import torch
    seq, offs = np.frombuffer(b"ACGT" * 10, np.uint8), np.array([0, 40], np.uint64)
    with pytest.raises(_lib.PolyhipError) as ei:  # SketchSize beyond 2^24: rejected before any device call
        mash.sketch_batch_packed(seq, offs, 3, (1 << 24) + 1, out=np.zeros((1, (1 << 24) + 1), np.uint32))
    assert ei.value.status == _lib.ERR_INVALID
    if not torch.cuda.is_available():
        # s < 2 is decided read by read ON THE DEVICE (mash.go:96,98): without one it is a device error, never a guess
        with pytest.raises(_lib.PolyhipError) as ei:
            mash.sketch_batch_packed(seq, offs, 3, 1)
        assert ei.value.status == _lib.ERR_HIP


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from poly_amd import _lib, mash
    with pytest.raises(_lib.PolyhipError) as ei:
        mash.New(21, 10).Sketch("ACGT" * 20)
    assert ei.value.status == _lib.ERR_HIP


def test_device_list_without_gpu():
    """polyhip_set_devices / polyhip_init (include/polyhip.h, "one host call over several GPUs"): without a usable device a
    non-empty list is refused with POLYHIP_ERR_HIP or _INVALID (never installed), the empty list is always fine, and a
    POLYHIP_DEVICES value that cannot be honoured is ignored at start-up -- the fan-out is no CPU fallback either."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from poly_amd import _lib, devices
    assert devices.get_devices() == []
    devices.set_devices([])
    for call in (lambda: devices.set_devices([0]), lambda: devices.set_devices([0, 0, 0]), lambda: devices.init(2)):
        with pytest.raises(_lib.PolyhipError) as ei:
            call()
        assert ei.value.status in (_lib.ERR_HIP, _lib.ERR_INVALID)
        assert devices.get_devices() == []
    with pytest.raises(_lib.PolyhipError):
        _lib.check(_lib.lib().polyhip_set_devices(None, 3))  # ids missing
    with pytest.raises(_lib.PolyhipError):
        _lib.check(_lib.lib().polyhip_set_devices(None, 65))  # more than the 64 a list holds
    code = ("from poly_amd import _lib, devices, mash\n"
            "print('list', devices.get_devices())\n"
            "try:\n    mash.New(21, 10).Sketch('ACGT' * 20)\nexcept _lib.PolyhipError as e:\n    print('status', e.status)\n")
    env = dict(os.environ, POLYHIP_DEVICES="0,0,0", PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "list []" in res.stdout and f"status {_lib.ERR_HIP}" in res.stdout, res.stdout + res.stderr
