"""The stand-alone probes behind DESIGN_HISTORY.md 11.1 and 11.6 (tools/probes/) are evidence that has to stay reproducible:
they must keep compiling for gfx950 / this host -- no GPU needed for that (hipcc cross-compiles)."""
import os
import shutil
import subprocess

import pytest

PROBES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes")


@pytest.mark.parametrize("src", ["preempt_lds_dma.hip", "dma_pageable.hip"])
def test_hip_probe_compiles_for_gfx950(tmp_path, src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / (src + ".o")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-c", "-o", str(out), os.path.join(PROBES, src)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 0


def test_the_lds_dma_probe_issues_the_instruction_it_is_about(tmp_path):
    """the DMA variant must contain `global_load_lds_dwordx4` with M0 set from a scalar, the control must not"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = tmp_path / "probe.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-S", "--cuda-device-only", "-o", str(asm),
                        os.path.join(PROBES, "preempt_lds_dma.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    bodies = {}
    name = None
    for line in text.splitlines():
        if line.startswith("_Z") and line.rstrip().endswith(":") or (line.startswith("_Z") and ":" in line.split(";")[0]):
            name = line.split(":")[0]
            bodies[name] = []
        elif name:
            bodies[name].append(line)
    dma = [k for k in bodies if "probeILb1E" in k]
    ctl = [k for k in bodies if "probeILb0E" in k]
    assert len(dma) == 1 and len(ctl) == 1, list(bodies)
    d, c = "\n".join(bodies[dma[0]]), "\n".join(bodies[ctl[0]])
    assert d.count("global_load_lds_dwordx4") >= 8 and "s_mov_b32 m0" in d
    assert "global_load_lds" not in c and ("ds_write_b128" in c or "ds_write2_b64" in c)


def test_sigprof_builds_and_exports_its_three_calls(tmp_path):
    so = tmp_path / "sigprof.so"
    r = subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", str(so), os.path.join(PROBES, "sigprof.c"), "-ldl"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout
    for s in ("sigprof_start", "sigprof_stop", "sigprof_dump"):
        assert s in syms
