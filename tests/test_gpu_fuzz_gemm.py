"""Randomised bit-exact sweeps of the contraction entry point (the tools do the work so that longer runs are one
command: `python tools/gemm_fuzz.py 2000 5`, `python tools/gmul_fuzz.py 5000 5`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("seed", [11, 12])
def test_random_gemm_extents_and_layouts_bit_exact(seed):
    """run_gemm's routing (tile shapes, carves, edge tiles, split-K, K tails, the short-K and small kernels) on random
    M, K, N -- tile boundaries +-1 over-represented -- and all four operand layouts: exact on small integers."""
    out = _run("gemm_fuzz.py", 120, seed)
    assert "mismatches 0" in out, out[-3000:]


def test_random_gemm_extents_and_layouts_bit_exact_fp64():
    """... and the fp64 kernels behind the same routing (tiled, wave-split, small, fallback)."""
    out = _run("gemm_fuzz.py", 80, 13, env={"FUZZ_DTYPE": "f64"})
    assert "mismatches 0" in out, out[-3000:]


@pytest.mark.parametrize("dtype,cases", [("f32", 14), ("f64", 8)])
def test_random_large_extents_on_the_pinned_bodies_bit_exact(dtype, cases):
    """Extents of 3000 .. 6400 (whole tiles, multiples of 4, anything) with one to a few dozen k-tiles: the routes of the
    pinned 256x256 / 256x128 bodies -- whole rounds, edge tiles run whole, carved blocks + border strips, hybrid stream-K,
    the short-K row streams -- on all four operand layouts: exact on small integers."""
    out = _run("pinned_fuzz.py", cases, 51, env={"FUZZ_DTYPE": dtype})
    assert "mismatches 0" in out, out[-3000:]


def test_large_layers_leave_the_pinned_body_with_bias_and_logistic_in_one_launch():
    """A recorded `W x + b`, alone and under logistic, on layers large enough for the pinned 256x256 body (full tiles, edge
    tiles, a hybrid stream-K shape, a short-K row stream): exact pre-activations, logistic at 2e-6."""
    out = _run("pinned_epilogue_check.py")
    assert "mismatches 0" in out and "MISMATCH" not in out, out[-3000:]


def test_random_layers_of_few_tiles_and_a_long_k_with_fused_epilogues():
    """The same on extents of 10 .. 170 output tiles with a K of 600 .. 4200: gemm_kwave.hip with two to eight workgroups
    per tile (round 4), whose last arriver adds the partial tiles in k order and carries the epilogue."""
    out = _run("kw_epilogue_fuzz.py", 30, 43, env={"FUZZ_SPLIT": "1"})
    assert "mismatches 0" in out, out[-3000:]


@pytest.mark.parametrize("seed", [21, 22])
def test_random_gmul_ranks_and_batches_bit_exact(seed):
    """`gmul lM lO lN` with ranks 0..3 on each side (`Reverse os` on the right operand, TOp.hs:81-88), a hidden batch on
    either operand or both, the batch-summed form, fp32 and fp64: numpy einsum, exact on small integers."""
    out = _run("gmul_fuzz.py", 400, seed)
    assert "mismatches 0" in out, out[-3000:]


@pytest.mark.parametrize("jit", ["1", "0"])
def test_random_closures_match_numpy(jit):
    """liftT over random expression trees of the whole symbolic vocabulary (arity 1-3, kinks of abs/signum/max/min
    included): the run-time specialised kernels (or a pre-fused functor when the classifier recognises one) and the
    bytecode VM against numpy in double, 1e-5 / 1e-11 (plus numpy's own drift in the element type)."""
    out = _run("expr_fuzz.py", 60 if jit == "1" else 150, 31, env={"TOPS_EXPR_JIT": jit})
    assert "mismatches 0" in out, out[-3000:]


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_random_mid_size_layers_with_fused_epilogues(dtype):
    """A recorded `W x + b` alone, under logistic and under tanh, on random mid-size extents (the wave-split kernels'
    final reduction carries the epilogue; ragged tiles, K tails): exact pre-activations, activations at 2e-6 / 1e-12."""
    out = _run("kw_epilogue_fuzz.py", 25, 41, env={"FUZZ_DTYPE": dtype})
    assert "mismatches 0" in out, out[-3000:]


def test_four_wave_32x32_tiles_on_the_steps_shapes_bit_exact():
    """gemm_t32.hip (round 5: four DMA-fed waves per 32x32 tile, the training step's two big contractions): about one
    round of tiles with a long K in all four operand layouts, ragged K / M / N, and the recorded `W x + b` with its
    logistic -- exact on small integers / 2e-6 on the activation."""
    out = _run("t32_check.py")
    assert "t32_check mismatches 0" in out, out[-3000:]


def test_tile_menu_kernel_on_ragged_extents_and_k_tails_bit_exact():
    """gemm_kw16.hip (round 6) through the product's routing: extents around the sizes its tile menu serves (multiples of 48
    and 80 +- a ragged edge, K tails, all four operand layouts) -- tools/kw16_check.py's list runs whichever route the
    library picks for each (a development build forces one menu entry per run: TOPS_GEMM_KW16=2 TOPS_GEMM_KW16_TILE=i)."""
    out = _run("kw16_check.py", "check")
    assert "kw16_check mismatches 0" in out, out[-3000:]


def test_learn_layer_shapes_tall_narrow_and_very_long_k_bit_exact():
    """Round 6, last: the reference's own network (784 -> 300 -> 100 -> 10, app/MNIST.hs) under a whole data set or a big batch --
    60000 x 784 x 300, 60000 x 300 x 100, 8192 x 300 x 100 forward (a tile per wave / per workgroup of gemm_kwave.hip), the
    cotangent 8192 x 100 x 300 (K < 128), the weight gradients 100 x 8192 x 300, 100 x 60000 x 300, 10 x 60000 x 100, 128 x 16384 x
    256 (a handful of tiles under a very long K: stream-K over 256 / 128 workgroups, up to 64 contributors a tile added in k
    order) -- exact on small integers, all four operand layouts, one launch each (tools/learn_check.py)."""
    out = _run("learn_check.py", "60000", "784", "300", "60000", "300", "100", "60000", "100", "12", "8192", "300", "100", "8200", "300", "100",
               "8192", "100", "300", "100", "8192", "300", "100", "60000", "300", "128", "16384", "256", "64", "8192", "784", "100", "3072", "100",
               "16", "4096", "2000", "40", "5000", "72", "10", "60000", "100", "12", "8192", "100", "16384", "64", "256", "300", "784", "60000")
    assert "learn_check mismatches 0" in out, out[-3000:]


@pytest.mark.parametrize("dtype_env,cases,seed", [({}, 300, 21), ({"ROUTE_DTYPE": "f64"}, 200, 22)], ids=["f32", "f64"])
def test_routing_boundaries_bit_exact(dtype_env, cases, seed):
    """Round 6 (last) moved many routing rules (tools/gemm_scan.py): shapes drawn AROUND their thresholds -- tile counts at the
    rounds of 256 workgroups / 2,048 waves, K at 16 ... 8,192, extents of 1, 8, 16, 96, 128 -- in all four layouts, with and without
    `beta * C`: exact on small integers (tools/route_fuzz.py)."""
    out = _run("route_fuzz.py", cases, seed, env=dtype_env)
    assert "mismatches 0" in out, out[-3000:]
