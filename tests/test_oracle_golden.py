"""The oracle (oracle/liboracle.so) against the committed golden vectors.

kat.json holds the reference's OWN known answers (tests.c, README.md); the other
files were produced by the compiled reference (tests/golden/make_golden.py).
This is what pins the oracle; the GPU parity tests then compare against it.
"""
import json
from pathlib import Path

import numpy as np
import pytest

import orclib as O
from seqalign_amd import workloads as W

GOLD = Path(__file__).resolve().parent / "golden"


def load(name):
    return json.loads((GOLD / name).read_text())


def oracle_scoring(spec):
    """Oracle builders only.  A preset is resolved through presets.json, which
    stores each scoring_system_* as data (penalties + dense letter table) taken
    from the compiled reference, together with a digest of the reference struct."""
    if "preset" in spec:
        entry = load("presets.json")[spec["preset"]]
        sc = O.build_scoring(entry["spec"], "oracle")
        digest = O.fnv(np.frombuffer(O.scoring_defined_bytes(sc), np.uint8))
        assert f"{digest:016x}" == entry["digest"]
        return sc
    return O.build_scoring(spec, "oracle")


def test_prng_vectors():
    g = load("prng.json")
    assert [int(x) for x in W.splitmix64(g["seed"], 8)] == g["first8"]
    b = W.dna_nw_150(2, 1)
    assert [b.seq_a(0).decode(), b.seq_b(0).decode()] == g["dna_nw_150_seed1_pair0"]


@pytest.mark.parametrize("v", load("kat.json")["nw"], ids=lambda v: v["src"])
def test_reference_known_answers_nw(v):
    sc = oracle_scoring(v["scoring"])
    a, b = v["a"].encode(), v["b"].encode()
    rc, M, A, B = O.oracle_fill(sc, a, b, 0)
    assert rc == 0
    if "match_scores" in v:  # README.md:118-145 pins every cell incl. the NW floor
        assert M.tolist() == sum(v["match_scores"], [])
        assert A.tolist() == sum(v["gap_a_scores"], [])
        assert B.tolist() == sum(v["gap_b_scores"], [])
    rc, score, ra, rb = O.oracle_nw_traceback(sc, a, b, M, A, B)
    assert rc == 0
    assert (ra.decode(), rb.decode()) == (v["result_a"], v["result_b"])
    if "score" in v:
        assert score == v["score"]


@pytest.mark.parametrize("v", load("kat.json")["sw"], ids=lambda v: v["src"])
def test_reference_known_answers_sw(v):
    sc = oracle_scoring(v["scoring"])
    rc, hits = O.oracle_sw(sc, v["a"].encode(), v["b"].encode(), min_score=0)
    assert rc == 0
    got = [[h["a"], h["b"]] for h in hits[:len(v["hits"])]]
    assert got == v["hits"]


def test_fill_small_all_flag_combinations():
    n_mat = 0
    for case in load("fill_small.json")["cases"]:
        sc = oracle_scoring(case["scoring"])
        for pair in case["pairs"]:
            a, b = pair["a"].encode(), pair["b"].encode()
            for key, is_sw in (("nw", 0), ("sw", 1)):
                if key not in pair:
                    continue
                rc, M, A, B = O.oracle_fill(sc, a, b, is_sw)
                assert rc == 0
                g = pair[key]
                assert M.tolist() == g["M"] and A.tolist() == g["A"] and B.tolist() == g["B"], (case["scoring"], pair["a"], pair["b"], key)
                n_mat += 3
                if "score" in g:
                    rc, score, ra, rb = O.oracle_nw_traceback(sc, a, b, M, A, B)
                    assert (rc, score, ra.decode(), rb.decode()) == (0, g["score"], g["result_a"], g["result_b"])
    assert n_mat > 700


@pytest.mark.parametrize("name", ["C2", "C2_related", "C3", "C4", "C5", "C5_rank7"])
def test_config_vectors(name):
    cfg = load("configs.json")[name]
    sc = oracle_scoring(cfg["scoring"])
    batch = W.make(cfg["gen"], cfg["n"], cfg["kwargs"])
    for p, g in enumerate(cfg["pairs"]):
        a, b = batch.seq_a(p), batch.seq_b(p)
        assert f"{O.fnv(np.frombuffer(a + b'|' + b, np.uint8)):016x}" == g["input"]
        rc, M, A, B = O.oracle_fill(sc, a, b, cfg["is_sw"])
        assert rc == 0
        assert (f"{O.fnv(M):016x}", f"{O.fnv(A):016x}", f"{O.fnv(B):016x}") == (g["M"], g["A"], g["B"])
        if cfg["is_sw"]:
            assert int(M.max()) == g["max"]
        else:
            rc, score, ra, rb = O.oracle_nw_traceback(sc, a, b, M, A, B)
            assert (rc, score, ra.decode(), rb.decode()) == (0, g["score"], g["result_a"], g["result_b"])


@pytest.mark.parametrize("name", O.PRESETS)
def test_preset_lookup_probes(name):
    import ctypes as C
    entry = load("presets.json")[name]
    sc = oracle_scoring({"preset": name})
    for x, y, want_s, want_m in entry["lookup"]:
        s, m = C.c_int(0), C.c_int(0)
        rc = O.oracle().orc_scoring_lookup(C.byref(sc), C.c_char(x.encode()), C.c_char(y.encode()), C.byref(s), C.byref(m))
        assert (rc, s.value, m.value) == (0, want_s, want_m), (x, y)


def test_unknown_pair_is_reported_not_fatal():
    """alignment_scoring.c:178-181 exit()s; the oracle returns a code instead."""
    sc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0], "use_match_mismatch": 0}, "oracle")
    rc, *_ = O.oracle_fill(sc, b"AC", b"AG", 0)
    assert rc == 1


@pytest.mark.parametrize("name", ["C3", "C4"])
def test_oracle_derived_sw_hit_lists_are_current(name):
    """tests/golden/sw_hits_oracle.json (ORACLE-derived, see make_sw_hits.py) equals a live oracle run."""
    g = load("sw_hits_oracle.json")[name]
    spec = load("presets.json")["BLOSUM62"]["spec"] if g["scoring"] == "BLOSUM62" else g["scoring"]
    sc = O.build_scoring(spec, "oracle")
    batch = W.make(g["gen"], g["n"], g["kwargs"])
    for p in range(0, batch.n_pairs, 4):
        rc, hits = O.oracle_sw(sc, batch.seq_a(p), batch.seq_b(p), g["min_score"])
        assert rc == 0
        assert [[h["score"], h["pos_a"], h["pos_b"], h["len_a"], h["len_b"], h["a"], h["b"]] for h in hits] == g["hits"][p]


def _refwalk_cases():
    """(label, oracle-side scoring spec, pairs as (a, b) bytes, min_score, expected hit rows per pair) of sw_hits_refwalk.json."""
    g = load("sw_hits_refwalk.json")
    for name in ("C3", "C4", "C3_low", "C4_low"):
        e = g[name]
        spec = load("presets.json")[e["scoring"]["preset"]]["spec"] if "preset" in e["scoring"] else e["scoring"]
        batch = W.make(e["gen"], e["of"], e["kwargs"])
        yield name, spec, [(batch.seq_a(p), batch.seq_b(p)) for p in range(e["n"])], e["min_score"], e["hits"]
    for k, r in enumerate(g["repeats"]):
        yield f"repeats[{k}]", r["scoring"], [(a.encode(), b.encode()) for a, b in r["pairs"]], r["min_score"], r["hits"]


def test_oracle_hit_lists_equal_the_reference_walked_ones():
    """the restatement's SW enumeration (orc_sw_hits) against tests/golden/sw_hits_refwalk.json -- hit lists whose
    every number the COMPILED REFERENCE computed (aligner_align + alignment_reverse_move per step; only candidate order and visited
    mask restated, in Python: orclib.ref_sw_hits).  Two independent enumerations of smith_waterman.c:137-277 must agree on every
    hit of every pair: all hits >= 60 of the 64 C3 / C4 pairs, tens of hits per pair at a quarter of that threshold, and tandem
    repeats under twelve scorings (ties on score AND column)."""
    total = 0
    for label, spec, pairs, thr, want in _refwalk_cases():
        sc = O.build_scoring(spec, "oracle")
        for p, (a, b) in enumerate(pairs):
            rc, hits = O.oracle_sw(sc, a, b, thr)
            assert rc == 0
            got = [[h["score"], h["pos_a"], h["pos_b"], h["len_a"], h["len_b"], h["a"], h["b"]] for h in hits]
            assert got == want[p], (label, p, len(got), len(want[p]))
            total += len(got)
    assert total > 3000
