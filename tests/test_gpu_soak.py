"""GPU tier: a seeded slice of the builder's soak, so that the driver's own GPUTEST carries randomised-scoring evidence.

The long runs (hundreds of thousands of alignments; profiles/r0N/*fuzz_e2e*.txt, *x2_check*.txt) are records of
seq-align_amd/tools/fuzz_e2e.py and x2_check.py.  These tests run the SAME loops, fixed seeds, bounded by trial count
(so the cases are the same on every box) with a wall-clock cap as a backstop:

  * fuzz_e2e: random scorings (all five flags, wildcards, case sensitivity) x random / related / tandem-repeat pairs;
    seqalign_nw_batch scores + strings and seqalign_sw_batch hit lists, through every sweep form and both traceback
    kernels, against the ORACLE (reference semantics: src/needleman_wunsch.c:34-146, src/smith_waterman.c:137-277).
  * x2_check: the packed int16 fills (two / four pairs per wave) against the 32-bit one-pair kernels on uniform, mostly
    uniform and ragged batches, incl. scorings at the edge of the int16 admission bound, plus oracle spot checks.

A mismatch raises SystemExit(1) inside the tool with the failing case printed.
"""
import importlib.util
from pathlib import Path

import pytest

import seqalign_amd as S

pytestmark = pytest.mark.gpu

TOOLS = Path(__file__).resolve().parent.parent / "seq-align_amd" / "tools"


def load(name):
    spec = importlib.util.spec_from_file_location(f"soak_{name}", TOOLS / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ctx():
    with S.Context(0) as c:
        yield c


@pytest.mark.parametrize("seed", [20260929, 7])
def test_fuzz_e2e_slice(ctx, seed):
    """Random scorings: NW strings and SW hit lists equal the oracle's."""
    fz = load("fuzz_e2e")
    try:
        r = fz.run(seconds=40.0, seed=seed, max_trials=60, ctx=ctx)
    except SystemExit as e:
        pytest.fail(f"fuzz_e2e mismatch (seed {seed}); the failing case is in the captured output (exit {e.code})")
    assert r["trials"] >= 20, r                    # 60 on a healthy box; the wall-clock cap is a backstop only
    assert r["sw_checked"] == 12 * r["trials"] and r["nw_checked"] > 0, r


def test_packed_fills_slice(ctx):
    """Packed int16 fills == the 32-bit kernels == the oracle, on uniform / mostly-uniform / ragged batches, NW and SW."""
    x2 = load("x2_check")
    x2.setup(seed=77, context=ctx)
    saved = {k: ctx.get_option(k) for k in ("pack16", "quad", "subbatches")}
    try:
        u = x2.check_uniform(seconds=20.0, max_trials=45)
        m = x2.check_mixed(seconds=12.0, max_trials=10)
        s = x2.check_sw(seconds=20.0, max_trials=30)
        g = x2.check_ragged(seconds=12.0, max_trials=8)
    except SystemExit as e:
        pytest.fail(f"x2_check mismatch; the failing case is in the captured output (exit {e.code})")
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
    assert u["batches"] >= 19 and u["quad_batches"] > 0 and u["oracle_pairs"] > 0, u      # every listed shape once at least
    assert m["batches"] >= 3 and m["oracle_pairs"] > 0, m
    assert s["batches"] >= 10 and s["oracle_pairs"] > 0, s
    assert g["batches"] >= 2 and g["oracle_pairs"] > 0, g
