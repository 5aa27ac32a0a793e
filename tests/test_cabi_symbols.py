"""CPU-side checks of the drop-in boundary: the library builds, loads, exports every
symbol include/tensorops_hip.h declares, and refuses to compute without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

import tensor_ops_amd  # noqa: F401
from tensor_ops_amd import capi


@pytest.fixture(scope="module")
def built(repo_root):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_tops_build", os.path.join(repo_root, "tensor-ops_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b.build()


def declared_symbols(repo_root):
    text = open(os.path.join(repo_root, "include", "tensorops_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(to_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(built, repo_root):
    names = declared_symbols(repo_root)
    assert len(names) >= 55
    out = subprocess.check_output(["nm", "-D", "--defined-only", built]).decode()
    exported = set(re.findall(r" T (to_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    # nothing but the C ABI leaks out of the library
    leaked = [l for l in out.splitlines() if " T " in l and " T to_" not in l]
    assert not leaked, leaked


def test_ctypes_table_matches_header(built, repo_root):
    names = set(declared_symbols(repo_root))
    assert set(capi.SIGNATURES) | {"to_last_error"} == names


def test_no_cpu_fallback(built):
    L = capi.lib()
    n = C.c_int(-1)
    assert L.to_device_count(C.byref(n)) == 0
    if n.value == 0:
        assert L.to_init(0) != 0
        assert b"no CPU fallback" in L.to_last_error()
        t = capi.c_tensor()
        d = (C.c_int64 * 1)(4)
        assert L.to_alloc(0, 1, d, 0, C.byref(t)) != 0  # not initialised -> loud error


def test_kernels_are_gfx950(built):
    """The shared object carries gfx950 code objects only (no other arch, no generic fallback)."""
    blob = open(built, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets
