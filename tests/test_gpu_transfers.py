"""Caller memory <-> device (csrc/runtime.cpp host_to_device / device_to_host): the copy engine never touches pageable
caller memory -- the bytes go through the library's pinned staging chunks -- while memory the caller pinned itself is
transferred in place; TOPS_PINNED_STAGING=0 restores the runtime's own path.  Bit-exact round trips at every size class
around the 4 MiB chunk, both dtypes, views included; the counters of to_transfer_stats say which way the bytes went.
(Why: DESIGN_HISTORY.md 11.1 -- downloads that came back with pieces of the destination unwritten under eight processes.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK = 4 << 20


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_round_trips_are_bit_exact_at_every_size_class(T, dtype):
    from tensor_ops_amd.hipt import HipT
    Td = HipT(0, dtype=dtype) if dtype is np.float64 else T
    es = np.dtype(dtype).itemsize
    rng = np.random.default_rng(5)
    sizes = [2, 63, 1023, 65536 // es + 1, CHUNK // es - 1, CHUNK // es, CHUNK // es + 1, 2 * CHUNK // es, 2 * CHUNK // es + 3, 5 * CHUNK // es + 17]
    before = Td.transfer_stats()
    for n in sizes:
        x = rng.integers(-2 ** 20, 2 ** 20, n).astype(dtype)
        x[0], x[-1] = 1, 2                     # (<= 64 EQUAL numbers are a constant the planner knows: no transfer)
        got = Td.put(x).numpy()
        assert got.dtype == dtype and np.array_equal(got, x), n
    after = Td.transfer_stats()
    assert after["staged_calls"] - before["staged_calls"] == 2 * len(sizes)
    assert after["staged_bytes"] - before["staged_bytes"] == 2 * es * sum(sizes)
    assert after["direct_calls"] == before["direct_calls"]


def test_a_destination_full_of_other_data_is_overwritten_everywhere(T):
    """the failure this replaces left pieces of the destination unwritten: download into a buffer that holds a sentinel"""
    import ctypes as C
    from tensor_ops_amd.capi import check, lib
    rng = np.random.default_rng(6)
    x = rng.integers(1, 1000, (1537, 2049)).astype(np.float32)    # 12.6 MB: four chunks, ragged
    d = T.put(x)
    out = np.full(x.shape, -7.0, dtype=np.float32)
    check(lib().to_download(d.h, out.ctypes.data_as(C.c_void_p), out.nbytes))
    assert np.array_equal(out, x)
    # a transposed view is packed on the device first, then staged
    outT = np.full((2049, 1537), -7.0, dtype=np.float32)
    dT = T.transp(d)      # (held across the raw C call: a temporary's handle would be released before it)
    check(lib().to_download(dT.h, outT.ctypes.data_as(C.c_void_p), outT.nbytes))
    assert np.array_equal(outT, x.T)


def test_caller_pinned_memory_goes_in_place(T):
    import ctypes as C
    from tensor_ops_amd.capi import check, lib
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")     # (the runtime the library itself is linked against)
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    n = 3 * CHUNK // 4 + 5
    src_p, dst_p = C.c_void_p(), C.c_void_p()
    assert hip.hipHostMalloc(C.byref(src_p), n * 4, 0) == 0 and hip.hipHostMalloc(C.byref(dst_p), n * 4, 0) == 0
    try:
        src = np.ctypeslib.as_array(C.cast(src_p, C.POINTER(C.c_float)), shape=(n,))
        dst = np.ctypeslib.as_array(C.cast(dst_p, C.POINTER(C.c_float)), shape=(n,))
        src[:] = np.arange(n, dtype=np.float32)
        dst[:] = -1.0
        before = T.transfer_stats()
        d = T.konst((n,), 0.0)
        mid = T.transfer_stats()
        check(lib().to_upload(d.h, src_p, n * 4))
        check(lib().to_download(d.h, dst_p, n * 4))
        after = T.transfer_stats()
        assert np.array_equal(src, dst)
        assert after["direct_calls"] - mid["direct_calls"] == 2 and after["direct_bytes"] - mid["direct_bytes"] == 8 * n
        assert after["staged_calls"] == mid["staged_calls"]
        assert mid["direct_calls"] == before["direct_calls"]
    finally:
        hip.hipHostFree(src_p)
        hip.hipHostFree(dst_p)


def test_a_pinned_prefix_of_a_pageable_array_is_staged(T):
    """`caller_pinned` judged a range by its first byte.  A host that registers the first pages of a larger
    pageable array (hipHostRegister) and hands over the WHOLE array must get the staged path -- the rest of the range is
    pageable -- while a range inside the registration still goes in place."""
    import ctypes as C
    from tensor_ops_amd.capi import check, lib
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    n = CHUNK // 2                                     # floats: 8 MiB in all
    raw = np.zeros(n + 4096, dtype=np.float32)
    off = (-raw.ctypes.data % 4096) // 4               # a page-aligned start inside the numpy buffer
    x = raw[off:off + n]
    x[:] = np.arange(n, dtype=np.float32) % 9973
    reg_bytes = 1 << 20                                # pin the first MiB only
    assert hip.hipHostRegister(C.c_void_p(x.ctypes.data), reg_bytes, 0) == 0
    try:
        d = T.konst((n,), 0.0)
        s0 = T.transfer_stats()
        check(lib().to_upload(d.h, x.ctypes.data_as(C.c_void_p), n * 4))          # whole array: 1 MiB pinned, 7 MiB pageable
        s1 = T.transfer_stats()
        assert s1["staged_calls"] - s0["staged_calls"] == 1 and s1["direct_calls"] == s0["direct_calls"]
        out = np.full(n, -7.0, dtype=np.float32)
        check(lib().to_download(d.h, out.ctypes.data_as(C.c_void_p), n * 4))
        assert np.array_equal(out, x)
        m = reg_bytes // 8                               # a range well inside the registration: in place
        d2 = T.konst((m,), 0.0)
        s2 = T.transfer_stats()
        check(lib().to_upload(d2.h, x.ctypes.data_as(C.c_void_p), m * 4))
        s3 = T.transfer_stats()
        assert np.array_equal(d2.numpy(), x[:m])
        # (in place if the runtime can name the registration's extent; staged -- the safe answer -- if it cannot)
        assert (s3["direct_calls"] - s2["direct_calls"]) + (s3["staged_calls"] - s2["staged_calls"]) == 1
    finally:
        hip.hipHostUnregister(C.c_void_p(x.ctypes.data))


def test_index_arguments_and_scalars_take_the_same_route(T):
    """argMax / oneHot / batch_gather / `!` move int64 indices and single elements: small transfers, always staged"""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((300, 10)).astype(np.float32)
    d = T.put(x, batched=True)
    before = T.transfer_stats()
    am = T.arg_max(d)
    assert np.array_equal(np.asarray(am), x.argmax(axis=1))
    oh = T.one_hot(10, 1.0, 0.0, list(map(int, am)))
    assert np.array_equal(oh.numpy(), np.eye(10, dtype=np.float32)[x.argmax(axis=1)])
    after = T.transfer_stats()
    assert after["staged_calls"] > before["staged_calls"] and after["direct_calls"] == before["direct_calls"]


def test_the_switch_restores_the_runtime_path(repo_root):
    code = r'''
import json, numpy as np
from tensor_ops_amd.hipt import HipT
T = HipT(0)
x = np.arange(3_000_001, dtype=np.float32)
ok = bool(np.array_equal(T.put(x).numpy(), x))
print(json.dumps(dict(T.transfer_stats(), ok=ok)))
'''
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=repo_root, TOPS_PINNED_STAGING="0"),
                       capture_output=True, text=True, timeout=300, cwd=repo_root)
    assert r.returncode == 0, r.stderr[-2000:]
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["ok"] and st["staged_calls"] == 0 and st["direct_calls"] == 2
