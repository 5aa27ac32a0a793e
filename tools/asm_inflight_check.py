"""Static hazard check of gfx950 assembly (hipcc -S) for the kernels that issue LDS reads and LDS DMA through inline asm.

The compiler takes an inline-asm `ds_read` for synchronous and knows nothing of an inline-asm `global_load_lds`: every
wait and barrier that orders them is written by hand in the kernels, and the compiler is free to move its own code between
them.  This tool proves, on the GENERATED code and along EVERY path of each kernel's control-flow graph (a forward dataflow
analysis iterated to its fixed point, so a loop's back edge carries what is in flight into the next iteration), that

 1. no instruction touches a VGPR while the inline-asm `ds_read` that fills it is in flight (`s_waitcnt lgkmcnt(n)` retires
    all but the n youngest LDS operations: LDS operations of a wave complete in order; scalar-memory operations share the
    counter but return out of order, so they are never counted as "younger");
 2. an LDS image is read only after the DMA that fills it has landed: the issuing wave's `s_waitcnt vmcnt(n)` (vector-memory
    loads return in order; stores are not counted as younger, which errs on the safe side) and -- for images the waves of a
    workgroup share -- an `s_barrier` behind that wait;
 3. an LDS image is overwritten by a DMA only after the reads of what it held have completed (`lgkmcnt`) and -- shared
    images -- every wave has passed a barrier behind that;
 4. no other LDS access happens while an LDS DMA is pending (the epilogues reuse the images as staging strips);
 5. every LDS DMA takes its M0 from an inline-asm `s_mov_b32 m0` with no compiler-written M0 in between (the asm
    statements do not declare M0 clobbered);
 6. the result registers of an inline-asm MFMA are not read or written by anything but another MFMA until MFMA_STATES
    wait states have passed (an instruction is one wait state, `s_nop n` is n + 1): the compiler pads the hazards of the
    MFMAs it issues itself, it does not look inside an asm string, and it may well place its own `v_accvgpr_read` -- a
    spill, the first instruction of an epilogue -- right behind one.

Which image an access touches cannot be read off an address register, so the kernels SAY it, in comments inside their asm
strings that cost no instruction and travel with the code through unrolling, peeling and rotation:

    ; @images N shared|private     N LDS images per operand; shared by the workgroup's waves or private to a wave
    ; @dma K                       this DMA fetches k-tile (t + K) of the current loop iteration t, into image (t + K) mod N
    ; @rd K                        this ds_read reads the image of k-tile (t + K)
    ; @advance                     end of a K-loop iteration: t becomes t + 1 (every outstanding K drops by one)

usage: asm_inflight_check.py [-v] file.s ...      exit status 1 when anything is reported.
tests/test_pinned_asm.py runs it over the product build of every kernel file with inline asm and over hand-written
snippets that each contain one of the hazards (the checker has to find them)."""
import re
import sys
from collections import defaultdict

CAP = 64          # counters saturate here (vmcnt is 6 bits on gfx9, lgkmcnt 4)
STALE = -9        # tags that fell this far behind are merged
MFMA_STATES = 19  # XDL write -> VALU / memory access of the result, 16-pass MFMA (the longest); s_nop 15 + s_nop 2

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
_ANN = re.compile(r"@(images|dma|rd|advance)\b\s*(-?\d+)?\s*(shared|private)?")
_BR = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\S+)")


def vregs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def aregs(text):
    """[(lo, hi)] of the AccVGPR operands"""
    out = []
    for m in _AREG.finditer(text):
        if m.group(1) is not None:
            out.append((int(m.group(1)), int(m.group(1))))
        else:
            out.append((int(m.group(2)), int(m.group(3))))
    return out


class Ins:
    __slots__ = ("ln", "text", "op", "rest", "asm", "ann", "kind", "uses", "dst", "aops", "states")

    def __init__(self, ln, text, asm, ann):
        self.ln, self.text, self.asm, self.ann = ln, text, asm, ann
        self.op, _, self.rest = text.partition(" ")
        self.uses = None
        self.dst = None
        self.aops = None
        op = self.op
        self.states = 1
        if op == "s_nop":
            try:
                self.states = int(self.rest.strip(), 0) + 1
            except ValueError:
                pass
        if op == "s_waitcnt":
            self.kind = "wait"
        elif op == "s_barrier":
            self.kind = "barrier"
        elif op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in self.rest):
            self.kind = "dma"
        elif op.startswith("ds_"):
            self.kind = "lds"
        elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "global_atomic", "buffer_atomic", "flat_atomic")):
            self.kind = "vmload"
        elif ann and not text:
            self.kind = "ann"
        elif op.startswith(("v_mfma", "v_smfmac")):
            self.kind = "mfma"
            # the matrix pipe takes the next MFMA only when this one's passes are through: an MFMA behind an MFMA is worth
            # its predecessor's passes in wait states (32x32 shapes: 16 passes, the 16x16 ones we use: 8)
            self.states = 16 if "_32x32" in op else 8
        else:
            self.kind = "plain"

    def regs(self):
        if self.uses is None:
            self.uses = vregs(self.rest)
        return self.uses


class State:
    """lgkm: VGPR -> LDS operations issued since its ds_read (the smaller, the more recent: worst case at a join);
    rd / dma: tag -> ('f', younger) in flight | ('l', 0) complete for this wave, no barrier since;  m0: who wrote M0 last."""
    __slots__ = ("lgkm", "rd", "dma", "m0", "images", "shared", "mfma")

    def __init__(self):
        self.lgkm, self.rd, self.dma, self.m0, self.images, self.shared = {}, {}, {}, "entry", 0, False
        self.mfma = {}   # (lo, hi) of an inline-asm MFMA's result -> wait states still owed

    def copy(self):
        s = State()
        s.lgkm, s.rd, s.dma, s.m0, s.images, s.shared = dict(self.lgkm), dict(self.rd), dict(self.dma), self.m0, self.images, self.shared
        s.mfma = dict(self.mfma)
        return s

    def key(self):
        return (tuple(sorted(self.lgkm.items())), tuple(sorted(self.rd.items())), tuple(sorted(self.dma.items())), self.m0, self.images, self.shared,
                tuple(sorted(self.mfma.items())))

    @staticmethod
    def _worse(a, b):
        if a is None:
            return b
        if b is None:
            return a
        if a[0] == "f" and b[0] == "f":
            return ("f", min(a[1], b[1]))
        return a if a[0] == "f" else b

    def join(self, o):
        """self <- the worst of both; True when self changed"""
        before = self.key()
        for r, y in o.lgkm.items():
            self.lgkm[r] = min(self.lgkm.get(r, CAP), y)
        for t, v in o.rd.items():
            self.rd[t] = State._worse(self.rd.get(t), v)
        for t, v in o.dma.items():
            self.dma[t] = State._worse(self.dma.get(t), v)
        if self.m0 != o.m0:
            self.m0 = "compiler" if "compiler" in (self.m0, o.m0) else ("entry" if "entry" in (self.m0, o.m0) else self.m0)
        self.images = max(self.images, o.images)
        self.shared = self.shared or o.shared
        for r, n in o.mfma.items():
            self.mfma[r] = max(self.mfma.get(r, 0), n)
        return self.key() != before


def _cnt(rest, name):
    m = re.search(name + r"\((\d+)\)", rest)
    return int(m.group(1)) if m else None


def step(st, ins, report):
    """the effect of one instruction on the state; hazards go to report(ins, kind, message)"""
    k = ins.kind
    ann = ins.ann
    if ann and ann[0] == "images":
        st.images, st.shared = int(ann[1]), ann[2] == "shared"
    if ann and ann[0] == "advance":
        for d in (st.rd, st.dma):
            moved = {}
            for t, v in d.items():
                nt = max(t - 1, STALE)
                moved[nt] = State._worse(moved.get(nt), v)
            d.clear()
            d.update(moved)
    if k == "ann":
        return
    # 6. results of inline-asm MFMAs
    if st.mfma:
        if k != "mfma" and ("a" in ins.rest):
            if ins.aops is None:
                ins.aops = aregs(ins.rest)
            for lo, hi in ins.aops:
                for (rlo, rhi), owed in st.mfma.items():
                    if lo <= rhi and rlo <= hi:
                        report(ins, "mfma-result", "accesses a[%d:%d], the result of an inline-asm MFMA, %d wait state(s) too early" % (rlo, rhi, owed))
                        break
        st.mfma = {r: n - ins.states for r, n in st.mfma.items() if n > ins.states}
    if k == "mfma" and ins.asm:
        d = aregs(ins.rest.split(",")[0])
        if d:
            st.mfma[d[0]] = MFMA_STATES
    # 1. registers in flight
    if st.lgkm and k != "wait":
        used = ins.regs()
        if k == "lds" and ins.op.startswith("ds_read") and ins.asm:
            # (the destination of a further read is not a use: LDS results return in order; its address is)
            parts = ins.rest.split(",")
            used = vregs(",".join(parts[1:])) if len(parts) > 1 else set()
        hit = used & st.lgkm.keys()
        if hit:
            report(ins, "inflight", "touches v%s while an inline-asm ds_read of it is in flight" % sorted(hit))
    if k == "mfma":
        return
    if k == "wait":
        n = _cnt(ins.rest, "lgkmcnt")
        if n is not None:
            st.lgkm = {r: y for r, y in st.lgkm.items() if y < n}
            for t, v in list(st.rd.items()):
                if v[0] == "f" and v[1] >= n:
                    if st.shared:
                        st.rd[t] = ("l", 0)
                    else:
                        del st.rd[t]
        n = _cnt(ins.rest, "vmcnt")
        if n is not None:
            for t, v in list(st.dma.items()):
                if v[0] == "f" and v[1] >= n:
                    if st.shared:
                        st.dma[t] = ("l", 0)
                    else:
                        del st.dma[t]
        return
    if k == "barrier":
        st.rd = {t: v for t, v in st.rd.items() if v[0] == "f"}
        st.dma = {t: v for t, v in st.dma.items() if v[0] == "f"}
        return
    if k == "lds":
        mem = not ins.op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle", "ds_nop"))
        tag = int(ann[1]) if ann and ann[0] == "rd" and ann[1] is not None else None
        if mem and tag is None and st.dma:
            if ins.asm and st.images and ins.op.startswith("ds_read"):
                report(ins, "unannotated", "inline-asm ds_read without an @rd tag in a kernel that declares @images")
            else:
                report(ins, "lds-under-dma", "LDS access while the DMA of tile(s) %s is pending (%s)" % (
                    sorted(st.dma), "no wait" if any(v[0] == "f" for v in st.dma.values()) else "landed for this wave, no barrier since"))
        if tag is not None:
            v = st.dma.get(tag)
            if v is not None:
                report(ins, "read-before-landing", "reads the image of tile t%+d whose DMA %s" % (
                    tag, "is still in flight (no vmcnt wait covers it)" if v[0] == "f" else "has landed for this wave only: no s_barrier behind the wait"))
        # an LDS operation: everything older gets one operation further from the tail of the queue
        for r in st.lgkm:
            st.lgkm[r] = min(st.lgkm[r] + 1, CAP)
        for t, v in st.rd.items():
            if v[0] == "f":
                st.rd[t] = ("f", min(v[1] + 1, CAP))
        if ins.asm and ins.op.startswith("ds_read"):
            for r in vregs(ins.rest.split(",")[0]):
                st.lgkm[r] = 0
            if tag is not None:
                st.rd[tag] = ("f", 0)
        return
    if k == "dma":
        tag = int(ann[1]) if ann and ann[0] == "dma" and ann[1] is not None else None
        if ins.asm:
            if st.m0 != "asm":
                report(ins, "m0", "LDS DMA with M0 last written by %s, not by the kernel's inline asm" % st.m0)
            if tag is None and st.images:
                report(ins, "unannotated", "inline-asm LDS DMA without a @dma tag in a kernel that declares @images")
        if tag is not None and st.images:
            for t, v in st.rd.items():
                if t < tag and (tag - t) % st.images == 0 and t > STALE:
                    report(ins, "overwrite-before-read", "DMA of tile t%+d overwrites the image of tile t%+d whose reads %s" % (
                        tag, t, "are still in flight (no lgkmcnt wait)" if v[0] == "f" else "are complete for this wave only: no s_barrier since"))
        for t, v in st.dma.items():
            if v[0] == "f":
                st.dma[t] = ("f", min(v[1] + 1, CAP))
        if tag is not None:
            st.dma[tag] = ("f", 0)
        elif ins.asm:
            st.dma[STALE] = ("f", 0)
        return
    if k == "vmload":
        for t, v in st.dma.items():
            if v[0] == "f":
                st.dma[t] = ("f", min(v[1] + 1, CAP))
        return
    # plain: M0
    if "m0" in ins.rest:
        first = ins.rest.split(",")[0].strip()
        if first == "m0":
            st.m0 = "asm" if ins.asm else "compiler"


def parse(path):
    """-> [(kernel name, [blocks]), ...]; a block = {label, ins: [Ins], succ: [labels], fall: bool}"""
    kernels = []
    cur = None
    blocks = None
    blk = None
    in_asm = False

    def new_block(label):
        nonlocal blk
        blk = {"label": label, "ins": [], "succ": [], "fall": True}
        blocks.append(blk)

    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            if "#ASMSTART" in raw:
                in_asm = True
                continue
            if "#ASMEND" in raw:
                in_asm = False
                continue
            code, _, comment = raw.partition(";")
            t = code.strip()
            ann = None
            if in_asm and "@" in comment:
                m = _ANN.search(comment)
                if m:
                    ann = (m.group(1), m.group(2), m.group(3))
            if not t:
                if ann and cur is not None:
                    blk["ins"].append(Ins(ln, "", True, ann))
                continue
            if t.endswith(":") and " " not in t:
                name = t[:-1]
                if name.startswith(".Lfunc_end"):
                    cur = None
                elif name.startswith(".L"):
                    if cur is not None:
                        new_block(name)
                else:
                    cur = name
                    blocks = []
                    kernels.append((name, blocks))
                    new_block(name)
                continue
            if t.startswith(".") or cur is None:
                continue
            ins = Ins(ln, t, in_asm, ann)
            blk["ins"].append(ins)
            m = _BR.match(t)
            if m:
                blk["succ"].append(m.group(2))
                if m.group(1) == "s_branch":
                    blk["fall"] = False
                new_block(None)
            elif ins.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
                blk["fall"] = False
                new_block(None)
    return kernels


def check_kernel(name, blocks, path, verbose=False, out=sys.stdout):
    if not any(i.asm and i.kind in ("lds", "dma") for b in blocks for i in b["ins"]):
        return 0
    index = {b["label"]: n for n, b in enumerate(blocks) if b["label"]}
    succ = []
    for n, b in enumerate(blocks):
        s = [index[l] for l in b["succ"] if l in index]
        if b["fall"] and n + 1 < len(blocks):
            s.append(n + 1)
        succ.append(s)
    entry = [None] * len(blocks)
    entry[0] = State()
    work = [0]
    queued = {0}
    found = {}
    passes = 0

    def report(ins, kind, msg):
        found.setdefault((ins.ln, kind), msg + "   `" + ins.text[:90] + "`")

    while work:
        n = work.pop()
        queued.discard(n)
        passes += 1
        if passes > 200 * len(blocks) + 1000:
            found[(0, "fixpoint")] = "no fixed point after %d block visits" % passes
            break
        st = entry[n].copy()
        for ins in blocks[n]["ins"]:
            if ins.kind == "plain" and not ins.ann and not st.lgkm and not st.mfma and "m0" not in ins.rest:
                continue
            step(st, ins, report)
        for s in succ[n]:
            if entry[s] is None:
                entry[s] = st.copy()
                changed = True
            else:
                changed = entry[s].join(st)
            if changed and s not in queued:
                queued.add(s)
                work.append(s)
    for (ln, kind), msg in sorted(found.items()):
        out.write("%s:%d: %s: [%s] %s\n" % (path, ln, name, kind, msg))
    if verbose:
        n_asm = sum(1 for b in blocks for i in b["ins"] if i.asm and i.kind in ("lds", "dma"))
        out.write("  %s: %d blocks, %d block visits, %d inline-asm LDS reads / DMAs, %d findings\n" % (name[:100], len(blocks), passes, n_asm, len(found)))
    return len(found)


def check(path, verbose=False, out=sys.stdout):
    return sum(check_kernel(name, blocks, path, verbose, out) for name, blocks in parse(path))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-v"]
    try:
        total = sum(check(p, "-v" in sys.argv) for p in args)
        print("hazards:", total)
    except BrokenPipeError:
        sys.exit(1)
    sys.exit(1 if total else 0)
