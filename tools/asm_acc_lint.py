"""Performance lint on generated gfx950 assembly: a basic block with 16 or more MFMAs (a K-loop body) must not move
accumulators between AccVGPRs and VGPRs.  With the MFMA builtin -- and even with inline-asm MFMAs whose accumulators are
"+a" operands -- the register allocator may carry the accumulators across the loop's back edge in VGPRs and copy them in and
out around the MFMAs of every k-tile (round 4 found 256 such moves next to the 64 MFMAs of gemm_kw64_kernel's loop, all
four operand layouts, and 137 in the compiler-scheduled fp64 kernel); an empty asm statement with the accumulators as "+a"
operands at the loop's head keeps them where the MFMAs want them.

usage: asm_acc_lint.py file.s [kernel-name-substring]      prints offending blocks, exit status 1 when there are any."""
import re
import sys


def lint(text, pattern=""):
    """[(kernel, block label, MFMAs, accvgpr moves)] for every block with >= 16 MFMAs"""
    out = []
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', text, re.M | re.S):
        name, body = m.group(1), m.group(2)
        if pattern not in name:
            continue
        blocks = re.split(r'^(\.LBB\d+_\d+):', body, flags=re.M)
        for i in range(1, len(blocks), 2):
            b = blocks[i + 1]
            nm = len(re.findall(r'^\s+v_mfma', b, re.M))
            na = len(re.findall(r'^\s+v_accvgpr', b, re.M))
            if nm >= 16:
                out.append((name, blocks[i], nm, na))
    return out


if __name__ == "__main__":
    rows = lint(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
    bad = [r for r in rows if r[3] > 0]
    for r in bad:
        print("%s %s: %d MFMAs, %d v_accvgpr moves" % r)
    print("%d MFMA blocks, %d with accumulator moves" % (len(rows), len(bad)))
    sys.exit(1 if bad else 0)
