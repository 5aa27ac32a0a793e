"""Mutation test of tools/asm_inflight_check.py: weaken, one at a time, the hand-written waits and barriers of a clean
assembly file and require the checker to object.  A wait is weakened by raising its count by one (`vmcnt(0)` -> `vmcnt(1)`,
`lgkmcnt(0)` -> `lgkmcnt(1)`) or deleting it; a barrier by deleting it.  Mutants the checker lets through are printed:
each is either a wait the kernel does not need on any path or a hole in the checker.
usage: asm_mutate.py file.s [max mutants per kernel]"""
import io
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_inflight_check as chk  # noqa: E402


def kernels_of(lines):
    """[(name, first line, last line)] of the functions that issue inline-asm LDS reads or DMA"""
    out, name, start, hot, in_asm = [], None, 0, False, False
    for n, l in enumerate(lines):
        t = l.split(";")[0].strip()
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        if t.endswith(":") and " " not in t and not t.startswith(".L"):
            name, start, hot = t[:-1], n, False
        elif t.startswith(".Lfunc_end") and name:
            if hot:
                out.append((name, start, n))
            name = None
        elif in_asm and (t.startswith("ds_read") or t.startswith("global_load_lds")):
            hot = True
    return out


def mutants(lines, lo, hi):
    """([(line index, replacement text or None to delete)], description): every run of hand-written waits (neighbours with
    nothing but scalar ALU code between them are one mutant: either alone would do the other's work) with one counter
    weakened, and -- in kernels whose images are shared by the waves -- every barrier deleted"""
    in_asm = False
    shared = any("@images" in l and "shared" in l for l in lines[lo:hi])
    runs, run = [], []
    for n in range(lo, hi):
        l = lines[n]
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        t = l.split(";")[0].strip()
        if not t:
            continue
        if t.startswith("s_waitcnt") and in_asm:
            run.append(n)
        elif t.startswith("s_") and not t.startswith(("s_barrier", "s_cbranch", "s_branch", "s_load", "s_buffer_load", "s_endpgm")):
            continue
        else:
            if run:
                runs.append(run)
            run = []
            if t == "s_barrier" and shared:
                yield [(n, None)], "s_barrier deleted"
    if run:
        runs.append(run)
    for run in runs:
        for cnt in ("vmcnt", "lgkmcnt"):
            edits = []
            for n in run:
                m = re.search(cnt + r"\((\d+)\)", lines[n])
                if m:
                    edits.append((n, lines[n].replace(m.group(0), "%s(%d)" % (cnt, int(m.group(1)) + 1))))
            if edits:
                yield edits, "%s + 1 in the wait(s) at line(s) %s" % (cnt, [n + 1 for n, _ in edits])


def main():
    path = sys.argv[1]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
    lines = open(path).read().split("\n")
    survived = total = 0
    for name, lo, hi in kernels_of(lines):
        body = lines[lo:hi + 1]
        sink = io.StringIO()
        with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
            f.write("\n".join(body))
            base = f.name
        assert chk.check(base, out=sink) == 0, "the unmutated kernel %s is not clean:\n%s" % (name, sink.getvalue())
        os.unlink(base)
        for k, (edits, what) in enumerate(mutants(body, 0, len(body))):
            if k >= limit:
                break
            mutated = list(body)
            for n, repl in edits:
                mutated[n] = "" if repl is None else repl
            n = edits[0][0]
            with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
                f.write("\n".join(mutated))
                mp = f.name
            total += 1
            if chk.check(mp, out=io.StringIO()) == 0:
                survived += 1
                print("SURVIVED %s: line %d %s   `%s`" % (name[:90], lo + n + 1, what, body[n].strip()))
            os.unlink(mp)
    print("mutants", total, "survived", survived)
    return 0


if __name__ == "__main__":
    sys.exit(main())
