"""What a bit-exact comparison writes down when it fails: WHICH elements differ, in the coordinates of the kernels that
could have produced them -- output tile (256x256 / 128x128 / 64x64), wave quadrant, row / column residues modulo the MFMA
tile, and, when the operands are at hand and integer-valued, the k-range whose products are missing from (or doubled in) a
wrong element.  `same(got, want, ...)` is `np.array_equal` with that side effect; the reports go to $TOPS_MISMATCH_DIR
(default gpurun_out/mismatch/) as one JSON file per failure, and a one-line summary to stderr.
Test infrastructure (tests/, tools/*_fuzz.py, tools/stress_suite.py); nothing in the product imports it."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiles(idx, size, limit=24):
    """[(tile_row, tile_col, mismatches)] for square tiles of `size`, worst first."""
    t = idx // size
    keys, counts = np.unique(t, axis=0, return_counts=True)
    order = np.argsort(-counts)[:limit]
    return {"tile": size, "distinct": int(len(keys)),
            "worst": [[int(keys[i][0]), int(keys[i][1]), int(counts[i])] for i in order]}


def _k_ranges(a_row, b_col, missing, step=16, limit=6):
    """k-ranges [k0, k1) on `step` boundaries whose products sum to `missing` (= want - got): a range the kernel skipped;
    or to -missing: a range it added twice.  Shortest first.  Exact on integer-valued operands only."""
    p = a_row.astype(np.float64) * b_col.astype(np.float64)
    K = len(p)
    cuts = list(range(0, K, step)) + [K]
    pre = np.concatenate([[0.0], np.cumsum(p)])[cuts]
    out = []
    for sign, what in ((1.0, "skipped"), (-1.0, "added twice")):
        d = pre[None, :] - pre[:, None]          # d[i, j] = sum over [cuts[i], cuts[j])
        ii, jj = np.nonzero((d == sign * missing) & (np.arange(len(cuts))[None, :] > np.arange(len(cuts))[:, None]))
        for i, j in sorted(zip(ii.tolist(), jj.tolist()), key=lambda ij: ij[1] - ij[0])[:limit]:
            out.append({"k0": cuts[i], "k1": cuts[j], "what": what})
    return out


def describe(got, want, a=None, b=None, **ctx):
    got, want = np.asarray(got), np.asarray(want)
    rep = {"ctx": {k: (v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)) for k, v in ctx.items()},
           "shape_got": list(got.shape), "shape_want": list(want.shape), "dtype": str(got.dtype)}
    if got.shape != want.shape:
        rep["kind"] = "shape"
        return rep
    neq = got != want
    nan_both = np.isnan(got) & np.isnan(want) if got.dtype.kind == "f" else np.zeros_like(neq)
    neq &= ~nan_both
    rep["mismatches"] = int(neq.sum())
    rep["elements"] = int(neq.size)
    if not rep["mismatches"]:
        return rep
    g2 = got.reshape(-1, got.shape[-1]) if got.ndim >= 2 else got.reshape(1, -1)
    w2 = want.reshape(g2.shape)
    idx = np.argwhere(neq.reshape(g2.shape))
    rep["rows"] = [int(idx[:, 0].min()), int(idx[:, 0].max())]
    rep["cols"] = [int(idx[:, 1].min()), int(idx[:, 1].max())]
    rep["first"] = [[int(r), int(c), float(g2[r, c]), float(w2[r, c])] for r, c in idx[:12]]
    rep["by_tile"] = [_tiles(idx, s) for s in (256, 128, 64, 32)]
    # where inside a 256x256 workgroup tile: wave quadrant (128x128), rows / columns modulo the 32x32 MFMA tile
    q = (idx % 256) // 128
    rep["wave_quadrants_of_256"] = {"%d%d" % (i, j): int(((q[:, 0] == i) & (q[:, 1] == j)).sum()) for i in (0, 1) for j in (0, 1)}
    rep["row_mod_32"] = sorted(set((idx[:, 0] % 32).tolist()))
    rep["col_mod_32"] = sorted(set((idx[:, 1] % 32).tolist()))
    diff = (g2.astype(np.float64) - w2.astype(np.float64))[neq.reshape(g2.shape)]
    vals, cnt = np.unique(diff, return_counts=True)
    order = np.argsort(-cnt)[:12]
    rep["got_minus_want"] = [[float(vals[i]), int(cnt[i])] for i in order]
    rep["nonfinite_got"] = int((~np.isfinite(g2)).sum()) if g2.dtype.kind == "f" else 0
    if a is not None and b is not None and np.asarray(a).ndim == 2 and np.asarray(b).ndim == 2:
        a, b = np.asarray(a), np.asarray(b)
        if a.shape[0] == g2.shape[0] and b.shape[1] == g2.shape[1]:
            ks = []
            for r, c in idx[:: max(1, len(idx) // 8)][:8]:
                ks.append({"row": int(r), "col": int(c),
                           "ranges": _k_ranges(a[r, :], b[:, c], float(w2[r, c]) - float(g2[r, c]))})
            rep["k_ranges"] = ks
    return rep


def write(rep, tag="mismatch"):
    d = os.environ.get("TOPS_MISMATCH_DIR") or os.path.join(ROOT, "gpurun_out", "mismatch")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "%s_%d_%d.json" % (tag, int(time.time() * 1000), os.getpid()))
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    return path


def same(got, want, a=None, b=None, **ctx):
    """np.array_equal(got, want); a failure is described (tiles, waves, k-ranges) and written down before False returns."""
    got, want = np.asarray(got), np.asarray(want)
    if got.shape == want.shape and np.array_equal(got, want):
        return True
    rep = describe(got, want, a=a, b=b, **ctx)
    rep["test"] = os.environ.get("PYTEST_CURRENT_TEST")
    rep["env"] = {k: v for k, v in os.environ.items() if k.startswith("TOPS_") or k.startswith("FUZZ_")}
    path = write(rep)
    worst = rep.get("by_tile", [{}])[0].get("worst", [])[:4]
    sys.stderr.write("BIT-EXACT MISMATCH %s: %s of %s elements, rows %s cols %s, 256-tiles %s -> %s\n" % (
        rep["ctx"], rep.get("mismatches"), rep.get("elements"), rep.get("rows"), rep.get("cols"), worst, path))
    return False
