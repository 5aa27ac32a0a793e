"""The kernels with inline-asm LDS reads, LDS DMA and MFMAs carry their own waits, barriers and wait states; the compiler
models none of it and schedules its own code around the asm statements.  tools/asm_inflight_check.py proves the six rules of
its header on the GENERATED gfx950 assembly of the PRODUCT build of every kernel file (the text the shipped objects were
assembled from: the build keeps it, tensor-ops_amd/build.py ASM_CHECKED), along every path of each kernel's control-flow
graph, back edges included.  Round 4 found three things this way that no GPU test had caught: the registers of the last,
unused fragment prefetch handed to the epilogue ahead of the post-loop wait (fp64 and fp32 bodies), and accumulator spills
/ AccVGPR shuffles placed a handful of scalar instructions behind the last inline-asm MFMA (fp32 body, stream-K).

The checker is itself under test: hand-written snippets that each contain exactly one hazard, and a mutation run that
weakens every hand-written wait / barrier of the clean assembly and requires the checker to object."""
import io
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _build_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_tops_build", os.path.join(ROOT, "tensor-ops_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _product_asm(src, tmp_path):
    """the device assembly of the product build of csrc/<src>: the file the build kept, or (stale / absent) a fresh -S"""
    b = _build_mod()
    kept = b.device_asm(src)
    spath = os.path.join(ROOT, "tensor-ops_amd", "csrc", src)
    csrc = os.path.dirname(spath)
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith((".hpp", ".h")) or f == src)
    if os.path.exists(kept) and os.path.getmtime(kept) >= newest:
        return kept
    out = tmp_path / (src + ".s")
    r = subprocess.run([HIPCC] + b.FLAGS + ["--cuda-device-only", "-S", "-o", str(out), "-x", "hip", spath],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


ASM_FILES = ["gemm_f32_mfma.hip", "gemm_f64.hip", "gemm_kwave.hip", "gemm_kw16.hip", "gemm_kwave_f64.hip", "gemm_skinnyk.hip",
             "gemm_skinnyk_f64.hip", "gemm_small.hip", "online_sgd.hip"]
ANNOTATED = {"gemm_f32_mfma.hip": "2 shared", "gemm_f64.hip": "2 shared", "gemm_kwave.hip": "2 private", "gemm_kw16.hip": "3 private",
             "gemm_kwave_f64.hip": "2 private"}


@pytest.mark.skipif(HIPCC is None, reason="hipcc not available")
@pytest.mark.parametrize("src", ASM_FILES)
def test_product_build_has_no_hazard(tmp_path, src):
    import asm_inflight_check
    assert src in _build_mod().ASM_CHECKED
    path = _product_asm(src, tmp_path)
    text = open(path).read()
    if src in ANNOTATED:  # the kernels say which image every asm access touches; the proof is about those statements
        assert text.count("@images") >= 4 and ("@images " + ANNOTATED[src]) in text
        assert text.count("; @rd ") > 50 and text.count("; @dma ") > 20 and text.count("@advance") >= 4
        assert "-DTOPS_GEMM_DEV" not in text
    sink = io.StringIO()
    n = asm_inflight_check.check(path, out=sink)
    assert n == 0, sink.getvalue()[:6000]


# kernels whose K loops are written for accumulators that live in AccVGPRs: (file, name substring) -> must have MFMA blocks
ACC_RESIDENT = [("gemm_f32_mfma.hip", "gemm_mfma"), ("gemm_kwave.hip", "gemm_kw_kernel"), ("gemm_kw16.hip", "gemm_kw16_kernel"), ("gemm_kwave_f64.hip", "gemm_kw64_kernel"),
                ("gemm_f64.hip", "gemm_f64_w4_kernel"), ("gemm_f64.hip", "gemm_f64_kernelILi128ELi128"),
                ("gemm_f64.hip", "gemm_f64_kernelILi64ELi64")]
# (not listed: the short-K streaming kernels, whose blocks leave THROUGH the MFMA stream -- AccVGPR reads are their design --
#  and the eight-wave 256x128 compiler-scheduled fp64 kernel, which has 256 registers a lane and spills when pinned)


@pytest.mark.skipif(HIPCC is None, reason="hipcc not available")
@pytest.mark.parametrize("src,kernel", ACC_RESIDENT, ids=[k for _, k in ACC_RESIDENT])
def test_k_loops_do_not_move_accumulators_between_register_files(tmp_path, src, kernel):
    """tools/asm_acc_lint.py on the product build: no block of 16 or more MFMAs carries a v_accvgpr move.  Round 4 found the
    fp64 wave-split kernel copying all 128 accumulator registers into AccVGPRs and back out every k-tile (256 moves next to
    64 MFMAs, every operand layout) and the compiler-scheduled fp64 kernel 137 -- the register allocator had parked the
    loop-carried accumulators in VGPRs; they are pinned now, and this keeps them pinned."""
    import asm_acc_lint
    rows = asm_acc_lint.lint(open(_product_asm(src, tmp_path)).read(), kernel)
    assert rows, "no MFMA block found for " + kernel
    bad = [r for r in rows if r[3] > 0]
    assert not bad, bad[:8]


def _snippet(body):
    return "k:\n" + body + "\n\ts_endpgm\n.Lfunc_end0:\n"


def _asm(*lines):
    return "\t;;#ASMSTART\n" + "".join("\t%s\n" % l for l in lines) + "\t;;#ASMEND\n"


LOOP_OK = (
    _asm("; @images 2 shared") + _asm("s_mov_b32 m0, s4", "s_nop 0") + _asm("global_load_lds_dwordx4 v1, s[2:3] offset:0 ; @dma 0")
    + _asm("global_load_lds_dwordx4 v1, s[2:3] offset:0 ; @dma 1") + _asm("s_waitcnt vmcnt(1)", "s_barrier")
    + _asm("ds_read_b128 v[4:7], v2 offset:0 ; @rd 0") + _asm("s_waitcnt lgkmcnt(0)")
    + ".LBB0_1:\n"
    + _asm("v_mfma_f32_32x32x2_f32 a[0:15], v4, v5, a[0:15]") + _asm("ds_read_b128 v[8:11], v2 offset:64 ; @rd 0")
    + _asm("s_waitcnt lgkmcnt(0)") + _asm("s_waitcnt vmcnt(0)", "s_barrier")
    + _asm("v_mfma_f32_32x32x2_f32 a[0:15], v8, v9, a[0:15]") + _asm("ds_read_b128 v[4:7], v3 offset:0 ; @rd 1")
    + _asm("s_mov_b32 m0, s4", "s_nop 0") + _asm("global_load_lds_dwordx4 v1, s[2:3] offset:0 ; @dma 2")
    + _asm("s_waitcnt lgkmcnt(0)") + _asm("; @advance")
    + "\ts_add_i32 s6, s6, 1\n\ts_cmp_lg_u32 s6, s7\n\ts_cbranch_scc1 .LBB0_1\n"
    + _asm("s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15") + "\ts_barrier\n\tv_accvgpr_read_b32 v20, a3\n\tds_write_b32 v21, v20\n")


def _findings(text, tmp_path, name="s.s"):
    import asm_inflight_check
    p = tmp_path / name
    p.write_text(text)
    sink = io.StringIO()
    n = asm_inflight_check.check(str(p), out=sink)
    return n, sink.getvalue()


def test_checker_accepts_a_correct_pipelined_loop(tmp_path):
    n, out = _findings(_snippet(LOOP_OK), tmp_path)
    assert n == 0, out


@pytest.mark.parametrize("what,old,new,kind", [
    ("the wait before the first fragments does not cover tile 0", "s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(2)", "read-before-landing"),
    ("no barrier behind the in-loop DMA wait: landed for this wave only", "\ts_waitcnt vmcnt(0)\n\ts_barrier\n\t;;#ASMEND\n\t;;#ASMSTART\n\tv_mfma",
     "\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\t;;#ASMSTART\n\tv_mfma", "read-before-landing"),
    ("the in-loop DMA wait leaves the next tile in flight (seen only through the back edge)", "\ts_waitcnt vmcnt(0)\n\ts_barrier\n\t;;#ASMEND\n\t;;#ASMSTART\n\tv_mfma",
     "\ts_waitcnt vmcnt(1)\n\ts_barrier\n\t;;#ASMEND\n\t;;#ASMSTART\n\tv_mfma", "read-before-landing"),
    ("an MFMA takes a fragment whose read is in flight", "ds_read_b128 v[8:11], v2 offset:64 ; @rd 0\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)",
     "ds_read_b128 v[8:11], v2 offset:64 ; @rd 0\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(1)", "inflight"),
    ("the last read of an iteration is still in flight at the top of the next (back edge) and at the exit",
     "s_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\t;;#ASMSTART\n\t; @advance", "s_nop 0\n\t;;#ASMEND\n\t;;#ASMSTART\n\t; @advance", "inflight"),
    ("the DMA overwrites an image the wave is still reading", "ds_read_b128 v[8:11], v2 offset:64 ; @rd 0\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\ts_barrier",
     "ds_read_b128 v[8:11], v2 offset:64 ; @rd 0\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)", "overwrite-before-read"),
    ("the epilogue reuses the LDS with the last DMA pending", "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15", "s_waitcnt lgkmcnt(0)\n\ts_nop 15", "lds-under-dma"),
    ("the epilogue reads an accumulator right behind the last MFMA", "\ts_nop 15\n\ts_nop 15\n", "\ts_nop 3\n", "mfma-result"),
    ("the compiler writes M0 between the asm's s_mov and its DMA", "\ts_nop 0\n\t;;#ASMEND\n\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v1, s[2:3] offset:0 ; @dma 2",
     "\ts_nop 0\n\t;;#ASMEND\n\ts_mov_b32 m0, s9\n\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v1, s[2:3] offset:0 ; @dma 2", "m0"),
    ("an asm DMA without a tag", "offset:0 ; @dma 2", "offset:0", "unannotated"),
])
def test_checker_finds_each_hazard(tmp_path, what, old, new, kind):
    assert LOOP_OK.count(old) >= 1, what
    broken = LOOP_OK.replace(old, new, 1) if kind != "read-before-landing" or "vmcnt(1)" in old else LOOP_OK.replace(old, new)
    n, out = _findings(_snippet(broken), tmp_path)
    assert n >= 1 and "[%s]" % kind in out, (what, out)


def test_private_images_need_no_barrier(tmp_path):
    private = LOOP_OK.replace("@images 2 shared", "@images 2 private").replace("\ts_barrier\n", "")
    n, out = _findings(_snippet(private), tmp_path)
    assert n == 0, out


@pytest.mark.skipif(HIPCC is None, reason="hipcc not available")
@pytest.mark.parametrize("src", ["gemm_kwave.hip", "gemm_kwave_f64.hip", "gemm_f64.hip"])
def test_weakening_any_wait_inside_a_k_loop_is_noticed(tmp_path, src):
    """Mutation run over the product assembly: every `vmcnt` wait and (shared images) every barrier that sits before the
    last `@advance` of its kernel -- the prologue and the K loop -- is needed on some path, so the checker has to object
    when it is weakened; `lgkmcnt` waits come in runs where one does the other's work, and what follows a loop (the drain, the
    barriers around the way out) is doubled by the next run's prologue in the stream-K kernels: most of them must be noticed."""
    import asm_mutate
    path = _product_asm(src, tmp_path)
    lines = open(path).read().split("\n")
    total = caught = essential = essential_caught = 0
    import asm_inflight_check as chk
    for name, lo, hi in asm_mutate.kernels_of(lines):
        body = lines[lo:hi + 1]
        # the prologue + K loop of every instance of a pinned body in this kernel: from its `@images` to its last `@advance`
        starts = [n for n, l in enumerate(body) if "@images" in l] + [len(body)]
        regions = [(a, max(n for n in range(a, b) if "@advance" in body[n])) for a, b in zip(starts, starts[1:])]
        for edits, what in asm_mutate.mutants(body, 0, len(body)):
            mutated = list(body)
            for n, repl in edits:
                mutated[n] = "" if repl is None else repl
            p = tmp_path / "m.s"
            p.write_text("\n".join(mutated))
            hit = chk.check(str(p), out=io.StringIO()) > 0
            total += 1
            caught += hit
            if any(a <= edits[0][0] <= b for a, b in regions) and ("vmcnt" in what or "barrier" in what):
                essential += 1
                essential_caught += hit
                assert hit, (name, lo + edits[0][0] + 1, what)
    assert essential >= 8 and essential_caught == essential
    assert caught >= 0.6 * total, (caught, total)
