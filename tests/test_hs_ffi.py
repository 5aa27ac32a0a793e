"""hs/ cannot be type-checked in this image (no GHC), so the one thing that can be checked mechanically is: every
`foreign import ccall` of the Haskell shim (and of the snippets in INTEGRATION.md) names an entry point that
include/tensorops_hip.h declares, with the same arity and the same scalar / pointer types -- and no entry point that can
block, copy to the host, compile (hiprtc), plan a recorded graph or talk to other ranks is imported `unsafe` (an `unsafe`
call holds its capability and stalls every other one at the next GC for as long as it runs).  The per-entry justification
is the table in INTEGRATION.md ("Import modes"); the test requires the table, the shim and MUST_BE_SAFE to agree."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tensorops_hip.h")
HS_DIR = os.path.join(ROOT, "hs")

# C parameter / return types -> the Haskell FFI type a maintainer has to write
C2HS = {
    "int": "CInt", "int32_t": "CInt", "to_status": "CInt", "unsigned": "CUInt", "int64_t": "Int64", "uint64_t": "Word64",
    "double": "CDouble", "float": "CFloat",
    "to_tensor": "Ptr ToTensor", "to_expr": "Ptr ToExpr", "to_graph": "Ptr ToGraph",
    "to_tensor*": "Ptr (Ptr ToTensor)", "to_expr*": "Ptr (Ptr ToExpr)", "to_graph*": "Ptr (Ptr ToGraph)",
    "int*": "Ptr CInt", "int32_t*": "Ptr Int32", "int64_t*": "Ptr Int64", "uint64_t*": "Ptr Word64", "double*": "Ptr CDouble",
    "float*": "Ptr CFloat", "void*": "Ptr ()", "void**": "Ptr (Ptr ())", "char*": "CString", "uint8_t*": "Ptr Word8",
    "unsigned char*": "Ptr Word8",
}

# Entry points that may run for much longer than an enqueue, and why.  Everything else is bounded by microseconds on every
# path (a query, handle bookkeeping, recording an op, or enqueueing one kernel; a pool miss costs one hipMalloc).
MUST_BE_SAFE = {
    "to_init": "creates the context, the stream and the pool",
    "to_shutdown": "synchronises and frees the pool",
    "to_sync": "waits for the stream",
    "to_upload": "host -> device copy; first runs every recorded op that still reads the destination",
    "to_download": "device -> host copy: demands the value and waits",
    "to_from_host": "host -> device copy",
    "to_index": "demands the value, waits, reads one element back",
    "to_arg_max": "demands the value, waits, reads the index back",
    "to_arg_min": "demands the value, waits, reads the index back",
    "to_one_hot": "uploads the index vector",
    "to_blas_dot": "returns a host scalar: waits",
    "to_blas_trace": "returns a host scalar: waits",
    "to_blas_sum": "returns a host scalar: waits",
    "to_batch_gather": "uploads the index vector",
    "to_lift": "outside a scope the closure's kernel is specialised with hiprtc on first use (hundreds of ms)",
    "to_expr_compile": "classifies the program and may compile it with hiprtc",
    "to_force": "plans the recorded graph (row programs compile with hiprtc) and launches it",
    "to_force_many": "plans the recorded graph (row programs compile with hiprtc) and launches it",
    "to_graph_end": "plans what the capture recorded, instantiates the hipGraph",
    "to_graph_launch": "replays a whole captured step: as many launches as the step has, and the capture may hold a collective",
    "to_graph_release": "destroys the hipGraph and releases what the capture retained",
    "to_graph_online_sgd": "runs the persistent kernel over the whole sample stream and waits for its verdict",
    "to_sgd_step_inplace": "first runs every recorded op that still reads the memory it overwrites (a plan, its launches)",
    "to_copy_into": "first runs every recorded op that still reads the memory it overwrites (a plan, its launches)",
    "to_copy_into_many": "first runs every recorded op that still reads the memory it overwrites (a plan, its launches)",
    "to_comm_unique_id": "loads RCCL",
    "to_comm_init": "RCCL rendezvous with the other ranks",
    "to_comm_allreduce_sum": "a collective: progress depends on the other ranks",
    "to_comm_shutdown": "destroys the communicator",
    "to_p2p_create": "allocates and exports the exchange buffers",
    "to_p2p_connect": "maps the peers' buffers (hipIpc)",
    "to_p2p_allreduce_sum": "a collective: the launch waits for the peers' flags (watchdog seconds)",
    "to_p2p_allreduce_sgd": "a collective: the launch waits for the peers' flags (watchdog seconds)",
    "to_batch_sum": "outside a scope it plans and launches what its operand recorded (the per-sample outer products' lowering: a plan, its launches, possibly a row program compiled with hiprtc)",
    "to_gmul": "an operand that is still deferred (a per-sample outer product recorded outside any scope, round 4) is produced first: a plan, its launches",
    "to_sum": "an operand that is still deferred (a per-sample outer product recorded outside any scope, round 4) is produced first: a plan, its launches",
    "to_scale": "an operand that is still deferred (a per-sample outer product recorded outside any scope, round 4) is produced first: a plan, its launches",
}


def c_prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"\b(to_status|const\s+char\s*\*|void)\s+(to_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = re.sub(r"\bconst\b", " ", a)
                a = a.replace("*", " * ")
                toks = a.split()
                stars = toks.count("*")
                toks = [t for t in toks if t != "*"]
                # the last identifier is the parameter's name unless the declaration has none
                base = " ".join(toks[:-1]) if len(toks) > 1 else toks[0]
                params.append(base + "*" * stars)
        protos[name] = ("CString" if "char" in ret else ("()" if ret == "void" else "CInt"), params)
    return protos


def hs_type_list(sig):
    """'A -> Ptr (Ptr B) -> IO C' -> ['A', 'Ptr (Ptr B)', 'IO C'] (split on top-level arrows)"""
    out, depth, cur = [], 0, ""
    i = 0
    while i < len(sig):
        c = sig[i]
        if c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
        if depth == 0 and sig[i:i + 2] == "->":
            out.append(" ".join(cur.split()))
            cur = ""
            i += 2
            continue
        cur += c
        i += 1
    out.append(" ".join(cur.split()))
    return out


IMPORT = re.compile(r'foreign\s+import\s+ccall\s+(safe|unsafe)?\s*"(?:tensorops_hip\.h\s+)?(&?)(\w+)"\s+(\S+)\s*::\s*([^\n]*)')


def hs_imports(path):
    out = []
    for m in IMPORT.finditer(open(path).read()):
        mode, amp, cname, hsname, sig = m.group(1) or "safe", m.group(2), m.group(3), m.group(4), m.group(5)
        sig = sig.split("--")[0].strip()
        out.append({"file": os.path.relpath(path, ROOT), "mode": mode, "addr": bool(amp), "c": cname, "hs": hsname, "sig": sig})
    return out


def all_hs_files():
    return [os.path.join(r, f) for r, _, fs in os.walk(HS_DIR) for f in fs if f.endswith(".hs")]


def test_header_parses():
    p = c_prototypes()
    assert len(p) >= 95 and p["to_gmul"] == ("CInt", ["int", "int", "int", "to_tensor", "to_tensor", "to_tensor*"])
    assert p["to_last_error"] == ("CString", []) and p["to_rand"][1][-2:] == ["uint64_t", "to_tensor*"]


def _check_import(imp, protos):
    assert imp["c"] in protos, "%s: `%s` is not declared in include/tensorops_hip.h" % (imp["file"], imp["c"])
    ret, params = protos[imp["c"]]
    want = [C2HS[p] for p in params]
    if imp["addr"]:   # a finaliser: FunPtr (Ptr a -> IO ()); the C function's status is dropped by the RTS
        m = re.fullmatch(r"FunPtr \((.*)\)", imp["sig"])
        assert m, imp
        got = hs_type_list(m.group(1))
        assert got[:-1] == want and got[-1] == "IO ()", (imp, want)
        return
    got = hs_type_list(imp["sig"])
    # (`void*` is any `Ptr a`: the pointee type is the caller's business)
    got = [("Ptr ()" if w == "Ptr ()" and re.fullmatch(r"Ptr (\w+|\(\))", g) else g) for g, w in zip(got, want + [None] * len(got))] if len(got) == len(want) + 1 else got
    assert got[:-1] == want, "%s: %s :: %s  but the header says (%s)" % (imp["file"], imp["c"], imp["sig"], ", ".join(params))
    assert got[-1] == "IO " + ret, (imp, ret)


def test_every_foreign_import_matches_the_header():
    protos = c_prototypes()
    imports = [i for f in all_hs_files() for i in hs_imports(f)]
    assert len(imports) >= 75
    for imp in imports:
        _check_import(imp, protos)


def test_integration_md_snippets_match_the_header_and_the_shim():
    protos = c_prototypes()
    doc = hs_imports(os.path.join(ROOT, "INTEGRATION.md"))
    assert len(doc) >= 10
    shim = {}
    for f in all_hs_files():
        for i in hs_imports(f):
            shim.setdefault(i["c"], i["mode"])
    for imp in doc:
        _check_import(imp, protos)
        if not imp["addr"]:
            assert imp["mode"] == shim.get(imp["c"], imp["mode"]), "INTEGRATION.md imports %s %s, hs/ imports it %s" % (imp["c"], imp["mode"], shim.get(imp["c"]))


def test_nothing_that_can_block_or_compile_is_imported_unsafe():
    for f in all_hs_files() + [os.path.join(ROOT, "INTEGRATION.md")]:
        for imp in hs_imports(f):
            if imp["addr"]:
                continue
            if imp["c"] in MUST_BE_SAFE:
                assert imp["mode"] == "safe", "%s imports %s unsafe: %s" % (imp["file"], imp["c"], MUST_BE_SAFE[imp["c"]])
    assert set(MUST_BE_SAFE) <= set(c_prototypes())


def test_the_import_mode_table_in_integration_md_is_complete():
    """INTEGRATION.md 'Import modes': one row per entry point the shim imports -- `| to_x | safe/unsafe | why |`."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rows = dict((m.group(1), m.group(2)) for m in re.finditer(r"^\|\s*`(to_\w+)`\s*\|\s*(safe|unsafe)\s*\|", text, flags=re.M))
    shim = {}
    for f in all_hs_files():
        for i in hs_imports(f):
            if not i["addr"]:
                assert shim.setdefault(i["c"], i["mode"]) == i["mode"], "%s imported with two modes" % i["c"]
    assert set(shim) == set(rows), (sorted(set(shim) - set(rows)), sorted(set(rows) - set(shim)))
    for name, mode in shim.items():
        assert rows[name] == mode, (name, mode, rows[name])
        assert (mode == "safe") or name not in MUST_BE_SAFE
