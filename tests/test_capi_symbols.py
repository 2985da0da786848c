"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/sgg_hip.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import sgg_amd  # noqa: F401
from sgg_amd import build, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sgg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgg_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    assert os.path.exists(path)
    dll = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(dll, n), "libsgg_hip.so does not export %s" % n
    assert set(names) == set(lib.SIGNATURES), set(names) ^ set(lib.SIGNATURES)
    # exported == declared: nothing else with the library's prefix leaks out of the shared object (helpers are hidden)
    nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in nm.splitlines() if l.split() and l.split()[-1].startswith("sgg_")})
    assert exported == names, set(exported) ^ set(names)
    loaded = lib.load_library(path)
    assert loaded.sgg_version() == 100


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.SggError):
        lib.HipKernels()
