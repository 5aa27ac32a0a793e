"""Scan gfx950 assembly for a use of a ds_read destination register before the next `s_waitcnt lgkmcnt(0)` (a linear scan
along the fall-through path of every block; an unconditional branch ends a trace).
The pinned GEMM kernels issue their LDS reads through inline asm, which the compiler takes for synchronous: a register
copy it places between such a read and the hand-written wait would move stale data.  usage: asm_inflight_check.py file.s

gemm_kwave*.hip and, since its round-3 refit, the pinned 256x256 body of gemm_f32_mfma.hip (-DTOPS_GEMM_DEV=2) are clean
(tests/test_pinned_asm.py).  gemm_f64.hip is flagged 32 times, every time for the SAME thing: the fragment prefetch of the
tile after the last one, whose results nobody uses and whose destination registers the epilogue reuses behind the loop's
final `s_waitcnt vmcnt(0) lgkmcnt(0)` (which the linear scan does not reach across the loop's back edge).
Benign (an LDS read lands within a few hundred cycles), but it is why this check is per file, not over the library."""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    bad = 0
    kernel = None
    inflight = {}
    in_asm = False
    for ln, line in enumerate(open(path), 1):
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            if t.endswith(":") and not t.startswith(".L"):
                kernel, inflight = t[:-1], {}
            continue
        if t.endswith(":"):
            if not t.startswith(".L"):
                kernel, inflight = t[:-1], {}
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
            inflight = {}
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):   # the text that follows is not reached by falling through
            inflight = {}
            continue
        used = set()
        for o in ops:
            used |= regs(o.split()[0]) if o else set()
        hit = used & set(inflight)
        if hit and not op.startswith("ds_read"):
            bad += 1
            print(f"{path}:{ln}: {kernel}: `{t}` touches v{sorted(hit)} while the ds_read of line {inflight[min(hit)]} is in flight")
        if op.startswith("ds_read") and ops and in_asm:  # (the compiler waits for the reads it issues itself)
            for r in regs(ops[0]):
                inflight[r] = ln
    return bad


if __name__ == "__main__":
    n = sum(check(p) for p in sys.argv[1:])
    print("in-flight uses:", n)
    sys.exit(1 if n else 0)
