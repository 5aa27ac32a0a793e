"""The wave-split GEMM kernels issue their LDS reads through inline asm, which the compiler takes for synchronous: a
register copy it places between such a read and the hand-written `s_waitcnt lgkmcnt(0)` moves stale data (it happened:
DESIGN.md section 3).  The kernels route every read through a temporary whose only consumer is the wait; this test
compiles them to gfx950 assembly (no GPU needed) and scans it for a touched in-flight register."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")


@pytest.mark.skipif(HIPCC is None, reason="hipcc not available")
@pytest.mark.parametrize("src,flags", [("gemm_kwave.hip", []), ("gemm_kwave_f64.hip", []),
                                       # the pinned 256x256 body of config 2, all four operand layouts (the development
                                       # build of the file: only those kernels; since its round-3 refit it is clean too)
                                       ("gemm_f32_mfma.hip", ["-DTOPS_GEMM_DEV=2"])])
def test_no_register_is_touched_while_its_lds_read_is_in_flight(tmp_path, src, flags):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_inflight_check
    out = tmp_path / (src + ".s")
    csrc = os.path.join(ROOT, "tensor-ops_amd", "csrc")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", "-o", str(out),
                        "-x", "hip", os.path.join(csrc, src), "-I", csrc, "-I", os.path.join(ROOT, "include")] + flags,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = out.read_text()
    assert text.count("ds_read_b") > 50 and "v_mfma_f" in text
    assert asm_inflight_check.check(str(out)) == 0
