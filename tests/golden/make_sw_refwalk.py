#!/usr/bin/env python3
"""REFERENCE-PINNED Smith-Waterman hit lists.

tests/golden/sw_hits_oracle.json comes from our own restatement (oracle/seqalign_oracle.c: orc_sw_hits).  This file is the
second, independent enumeration: every number in it was computed by the COMPILED REFERENCE (oracle/_ref: aligner_align(is_sw = 1)
for the three matrices, alignment_reverse_move for every step of every walk); only the candidate order and the visited mask of
smith_waterman.c:71-86,137-277 -- which hold no arithmetic, and which cannot be compiled here (un-vendored sort_r) -- are
restated, in ~40 lines of Python: tests/orclib.py: ref_sw_hits.  Both the oracle (CPU tier, tests/test_oracle_golden.py) and
seqalign_sw_batch (`-m gpu`, tests/test_gpu_parity.py) are compared with it, hit for hit.

Sections:
  C3, C4          the 64 seeded pairs of BASELINE configs[2] / configs[3], every hit >= the command line's default --minscore (60)
  C3_low, C4_low  16 of the same pairs at a quarter of that threshold: tens of hits per pair, i.e. walks abandoned at cells that
                  earlier walks marked, marks left behind by abandoned walks, equal-score candidates in several columns
  repeats         tandem repeats and short related pairs under 12 scorings (gap flags, wildcard, free end gaps): many candidates
                  with equal score and equal column -- the ties the comparator leaves to the sort

    python tests/golden/make_sw_refwalk.py        ->  tests/golden/sw_hits_refwalk.json      (needs oracle/_ref, i.e. /root/reference)
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))

import orclib as O  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

CONFIGS = {
    # name: generator, kwargs, pairs, scoring spec ("ref" side), min_score
    "C3": dict(gen="dna_sw_read_vs_ref", kwargs=dict(seed=2), n=64, scoring={"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}, min_score=60),
    "C4": dict(gen="protein_sw_300", kwargs=dict(seed=3), n=64, scoring={"preset": "BLOSUM62"}, min_score=60),
    "C3_low": dict(gen="dna_sw_read_vs_ref", kwargs=dict(seed=2), n=16, scoring={"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}, min_score=15),
    "C4_low": dict(gen="protein_sw_300", kwargs=dict(seed=3), n=16, scoring={"preset": "BLOSUM62"}, min_score=15),
}

REPEAT_SCORINGS = [
    # match, mismatch, gap_open, gap_extend, no_start_gap, no_end_gap, no_gaps_in_a, no_gaps_in_b, no_mismatches, case_sensitive
    [1, -2, -4, -1, 0, 0, 0, 0, 0, 0], [2, -2, -2, -1, 0, 0, 0, 0, 0, 0], [1, -1, 0, -1, 0, 0, 0, 0, 0, 0], [3, -1, -1, 0, 0, 0, 0, 0, 0, 0],
    [2, -3, -5, -2, 1, 1, 0, 0, 0, 0], [2, -2, -2, -1, 0, 0, 1, 0, 0, 0], [2, -2, -2, -1, 0, 0, 0, 1, 0, 0], [2, -4, -2, -1, 0, 0, 1, 1, 0, 0],
    [2, -2, -3, -1, 0, 0, 0, 0, 1, 0], [1, 0, -1, -1, 0, 0, 0, 0, 0, 0], [4, -4, -6, -2, 0, 1, 0, 0, 0, 1], [2, -1, -2, 0, 1, 0, 0, 0, 0, 0],
]


def repeat_pairs(seed: int):
    rng = W.Rng(seed)

    def rand(n, alpha=b"ACGT"):
        return bytes(alpha[i] for i in rng.below(len(alpha), n)) if n else b""
    pairs = []
    for k in range(6):
        unit = rand(int(2 + rng.below(6, 1)[0]))
        pairs.append((unit * int(3 + rng.below(12, 1)[0]), rand(3) + unit * int(2 + rng.below(14, 1)[0]) + rand(2)))
    for k in range(4):
        a = rand(int(30 + rng.below(60, 1)[0]))
        cut = int(rng.below(20, 1)[0])
        pairs.append((a, rand(int(rng.below(25, 1)[0])) + a[cut:cut + 40] + rand(8) + a[cut + 5:cut + 35] + rand(int(rng.below(25, 1)[0]))))
    pairs.append((b"acgtNNacgtACGT", b"ACGTacgtnnACGTACGT"))     # case folding + the wildcard the scorings below add
    pairs.append((b"AAAAAAAAAAAAAAAA", b"AAAAAAAAAAAA"))         # every cell a candidate; every score shared by a whole anti-diagonal
    return pairs


def hit_rows(hits):
    return [[h["score"], h["pos_a"], h["pos_b"], h["len_a"], h["len_b"], h["a"], h["b"]] for h in hits]


def main():
    assert O.ref() is not None, "oracle/_ref is not built (needs /root/reference): make -C oracle ref"
    out = {"_provenance": "every score, matrix cell and walk step computed by the compiled reference (oracle/_ref: aligner_align, "
                          "alignment_reverse_move); candidate order + visited mask of smith_waterman.c:71-86,137-277 restated in "
                          "tests/orclib.py: ref_sw_hits; see make_sw_refwalk.py"}
    for name, cfg in CONFIGS.items():
        sc = O.build_scoring(cfg["scoring"], "ref")
        batch = W.make(cfg["gen"], 64, cfg["kwargs"])          # (the _low sections: the first 16 of the same 64 pairs)
        pairs = [hit_rows(O.ref_sw_hits(sc, batch.seq_a(p), batch.seq_b(p), cfg["min_score"])) for p in range(cfg["n"])]
        out[name] = dict(gen=cfg["gen"], kwargs=cfg["kwargs"], of=64, n=cfg["n"], scoring=cfg["scoring"], min_score=cfg["min_score"],
                         hits=pairs)
        print(name, "hits per pair: min", min(map(len, pairs)), "max", max(map(len, pairs)), "total", sum(map(len, pairs)))
    reps = []
    for k, init in enumerate(REPEAT_SCORINGS):
        spec = {"init": init, "wildcards": [["N", 0 if k % 2 else -1]]}
        sc = O.build_scoring(spec, "ref")
        pairs = repeat_pairs(9000 + k)
        thr = (3 + k % 3) * init[0]
        reps.append(dict(scoring=spec, min_score=thr, pairs=[[a.decode(), b.decode()] for a, b in pairs],
                         hits=[hit_rows(O.ref_sw_hits(sc, a, b, thr)) for a, b in pairs]))
    out["repeats"] = reps
    print("repeats:", sum(len(h) for r in reps for h in r["hits"]), "hits over", sum(len(r["pairs"]) for r in reps), "pairs")
    path = HERE / "sw_hits_refwalk.json"
    path.write_text(json.dumps(out, separators=(",", ":")) + "\n")
    print(path.name, path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
