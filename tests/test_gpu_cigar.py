"""GPU tier: CIGAR as the batch calls' output (seqalign_nw_batch_cigar / seqalign_sw_batch_cigar).

north_star asks for "identical CIGAR/alignment strings".  The reference prints the two gapped strings only
(src/tools/nw_cmdline.c:78-149, src/alignment.h:33-40); CIGAR is the derived format defined in include/seqalign_hip.h: seq_a
is the query, seq_b the reference -- '-' in result_b = I, '-' in result_a = D, M or '=' / 'X' otherwise.  So the expectation is
always a run-length encoding, done here in Python, of REFERENCE strings: the golden alignments the compiled reference produced
(tests/golden/configs.json), the reference-walked SW hits (sw_hits_refwalk.json), the oracle's strings on ragged / flagged
batches.  On plain scorings the device walks come home as bit planes and the CIGAR is made from those without any string
(host/sa_moves.c); the three-matrix and host paths encode their strings -- every path must say the same.
"""
import itertools
import json
from pathlib import Path

import numpy as np
import pytest

import orclib as O
import seqalign_amd as S
from seqalign_amd import workloads as W

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"
M, EQX = 1, 2


def load(name):
    return json.loads((GOLDEN / name).read_text())


@pytest.fixture(scope="module")
def ctx():
    with S.Context(0) as c:
        yield c


@pytest.fixture
def opts(ctx):
    changed = {}

    def set_(**kv):
        for k, v in kv.items():
            changed.setdefault(k, ctx.get_option(k))
            ctx.set_option(k, v)
    yield set_
    for k, v in changed.items():
        ctx.set_option(k, v)


def py_cigar(ra, rb, fmt, fold=True) -> str:
    """include/seqalign_hip.h's definition, independently: seq_a = query, seq_b = reference."""
    if isinstance(ra, str):
        ra, rb = ra.encode(), rb.encode()
    ops = []
    for x, y in zip(ra, rb):
        if x == 45: op = "D"
        elif y == 45: op = "I"
        elif fmt == M: op = "M"
        else: op = "=" if (bytes([x]).lower() == bytes([y]).lower() if fold else x == y) else "X"
        if ops and ops[-1][0] == op: ops[-1][1] += 1
        else: ops.append([op, 1])
    return "".join(f"{n}{op}" for op, n in ops)


def consumed(cigar: str):
    """(query characters, reference characters, columns) a CIGAR accounts for."""
    q = r = cols = 0
    num = ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
            continue
        n = int(num); num = ""
        cols += n
        q += n if ch in "M=XI" else 0
        r += n if ch in "M=XD" else 0
    return q, r, cols


@pytest.mark.parametrize("where", ["device", "device-three-matrices", "host"])
@pytest.mark.parametrize("name", ["C2", "C2_related", "C5", "C5_rank7"])
def test_nw_cigar_of_the_golden_alignments(ctx, opts, name, where):
    """The 64 golden NW pairs of each config (strings produced by the COMPILED REFERENCE's needleman_wunsch_align2): their CIGAR, both
    formats, from the direction-byte path (planes -> run lengths, no strings), the three-matrix walkers and the host traceback."""
    if where == "device-three-matrices":
        opts(nw_dirs=0)
    elif where == "host":
        opts(traceback="host")
    cfg = load("configs.json")[name]
    sc = S.make_scoring(cfg["scoring"])
    batch = W.make(cfg["gen"], cfg["n"], cfg["kwargs"])
    for fmt in (M, EQX):
        got = ctx.nw_batch_cigar(batch, sc, fmt)
        ran = ctx.last_call()
        if where == "device":
            assert "fill_nw_dirs" in ran and not any(k.startswith("fill_stream") for k in ran), ran
        for p, g in enumerate(cfg["pairs"]):
            want = py_cigar(g["result_a"], g["result_b"], fmt)
            assert got[p] == (g["score"], want.encode()), (name, where, fmt, p)
            assert consumed(want) == (int(batch.len_a[p]), int(batch.len_b[p]), len(g["result_a"]))


def test_nw_cigar_ragged_flags_empty_and_slots(ctx):
    """Every flag combination (free end gaps put I / D runs at the ends), wildcards, mixed case (EQX folds unless the scoring is case
    sensitive), ragged lengths with empty sequences: against the oracle's strings.  Then the slots: a CIGAR that does not fit its
    pair's slot is SEQALIGN_E_NOMEM, one that fits exactly is delivered, and nothing is written outside a slot."""
    for idx, flags in enumerate(itertools.product([0, 1], repeat=5)):
        case_sensitive = idx & 1
        spec = {"init": [1, -6 if (flags[2] and flags[3]) else -2, -4, -1, *flags, case_sensitive], "wildcards": [["N", -1]] if idx % 3 == 0 else []}
        sc = S.make_scoring(spec)
        osc = O.Scoring.from_buffer_copy(bytes(sc))
        r = W.ragged(40, seed=idx + 900, max_len=90, lower_frac=0.3, extra=b"N" if spec["wildcards"] else b"")
        pairs = [(b"", b""), (b"ACGT", b""), (b"", b"TT")] + [(r.seq_a(p), r.seq_b(p)) for p in range(r.n_pairs)]
        batch = W.from_pairs(pairs)
        for fmt in (M, EQX):
            got = ctx.nw_batch_cigar(batch, sc, fmt)
            for p, (a, b) in enumerate(pairs):
                rc, s, ra, rb = O.oracle_nw(osc, a, b)
                assert rc == 0 and got[p] == (s, py_cigar(ra, rb, fmt, fold=not case_sensitive).encode()), (spec, fmt, p, a, b)
    # slots
    sc = S.make_scoring({"preset": "default"})
    batch = W.dna_nw_150(300, seed=77, related=True)
    full = ctx.nw_batch_cigar(batch, sc, M)
    longest = max(len(c) for _, c in full)
    assert longest > 4                                            # some pair has gaps: "150M" is 4
    off, out, out_len, out_score = ctx.nw_batch_cigar(batch, sc, M, slot=longest + 1, raw=True)
    for p, (s, c) in enumerate(full):
        assert out[int(off[p]):int(off[p]) + int(out_len[p]) + 1].tobytes() == c + b"\0" and out_score[p] == s
    with pytest.raises(S.SeqAlignError) as e:
        ctx.nw_batch_cigar(batch, sc, M, slot=longest)
    assert e.value.code == S.E_NOMEM


def test_nw_cigar_full_size_c2_equals_the_strings(ctx):
    """BASELINE configs[1] at full size (10 000 pairs, the related variant: real gaps): the CIGAR call says what the string call says
    for every pair, scores included, and every CIGAR accounts for both sequences completely."""
    sc = S.make_scoring({"preset": "default"})
    batch = W.dna_nw_150(10000, seed=1, related=True)
    strings = ctx.nw_batch(batch, sc)
    for fmt in (M, EQX):
        got = ctx.nw_batch_cigar(batch, sc, fmt)
        ran = ctx.last_call()      # the direction-byte path (packed fills at this size): planes home, no strings anywhere
        assert all(k.startswith("fill_nw_dirs") or k.startswith("walk_moves") for k in ran) and any(k.startswith("walk_moves") for k in ran), ran
        gaps = 0
        for p, (s, ra, rb) in enumerate(strings):
            want = py_cigar(ra, rb, fmt)
            assert got[p] == (s, want.encode()), (fmt, p)
            gaps += "I" in want or "D" in want
        assert gaps > 1000


def _refwalk_sections():
    g = load("sw_hits_refwalk.json")
    for name in ("C3", "C4", "C3_low", "C4_low"):
        e = g[name]
        batch = W.make(e["gen"], e["of"], e["kwargs"])
        yield name, e["scoring"], batch.slice(0, e["n"]), e["min_score"], e["hits"]
    for k, r in enumerate(g["repeats"]):
        yield f"repeats[{k}]", r["scoring"], W.from_pairs([(a.encode(), b.encode()) for a, b in r["pairs"]]), r["min_score"], r["hits"]


@pytest.mark.parametrize("where", ["device", "device-three-matrices", "host"])
def test_sw_cigar_of_the_reference_walked_hits(ctx, opts, where):
    """Every hit of tests/golden/sw_hits_refwalk.json (computed by the compiled reference's fill and reverse moves) and of
    sw_hits_oracle.json: its CIGAR, both formats, and the hit's other fields, through max_hits = 1 (the packed best-hit call: planes),
    4 (the one-trip multi-hit call: planes) and unlimited (the three-trip path: strings on the device), on direction bytes, on three
    matrices and through the host enumeration."""
    if where == "device-three-matrices":
        opts(sweep_dirs=0, nw_dirs=0)
    elif where == "host":
        opts(traceback="host")
    total = 0
    sections = list(_refwalk_sections())
    orc = load("sw_hits_oracle.json")
    for name in ("C3", "C4"):
        e = orc[name]
        sections.append((f"oracle:{name}", {"preset": "BLOSUM62"} if e["scoring"] == "BLOSUM62" else e["scoring"],
                         W.make(e["gen"], e["n"], e["kwargs"]), e["min_score"], e["hits"]))
    for label, spec, batch, thr, rows in sections:
        sc = S.make_scoring(spec)
        fold = not sc.case_sensitive
        for max_hits, fmt in ((1, M), (4, EQX), (1 << 20, M), (4, M)):
            got = ctx.sw_batch_cigar(batch, sc, thr, max_hits=max_hits, fmt=fmt, hit_cap=1 << 16, cigar_cap=1 << 22)
            for p in range(batch.n_pairs):
                want = [dict(score=h[0], pos_a=h[1], pos_b=h[2], len_a=h[3], len_b=h[4], length=len(h[5]), cigar=py_cigar(h[5], h[6], fmt, fold))
                        for h in rows[p][:max_hits]]
                assert got[p] == want, (label, where, max_hits, fmt, p)
                for h in want:
                    assert consumed(h["cigar"]) == (h["len_a"], h["len_b"], h["length"])
                total += len(want)
    assert total > 5000


def test_sw_cigar_hits_lie_back_to_back_and_nomem_delivers_a_prefix(ctx):
    """The CIGARs of a call lie back to back in the caller's buffer (str_off of hit h + 1 = end of hit h's text + NUL): their exact
    lengths are counted from the planes before anything is placed.  A buffer too small for all of them: SEQALIGN_E_NOMEM after a
    prefix of the hits, in pair order, has been delivered intact."""
    e = load("sw_hits_refwalk.json")["C3_low"]
    sc = S.make_scoring(e["scoring"])
    batch = W.make(e["gen"], e["of"], e["kwargs"]).slice(0, e["n"])
    for max_hits in (1, 4):
        rc, n, hits, out = ctx.sw_batch_cigar(batch, sc, e["min_score"], max_hits=max_hits, fmt=M, hit_cap=4096, cigar_cap=1 << 20, raw=True)
        assert rc == 0 and n == sum(min(max_hits, len(r)) for r in e["hits"])
        at, texts = 0, []
        for k in range(n):
            assert hits[k].str_off == at
            end = at
            while out[end]:
                end += 1
            texts.append((hits[k].pair, out[at:end].tobytes()))
            at = end + 1
        used = at
        rc2, n2, hits2, out2 = ctx.sw_batch_cigar(batch, sc, e["min_score"], max_hits=max_hits, fmt=M, hit_cap=4096, cigar_cap=used // 2, raw=True)
        assert rc2 == S.E_NOMEM and 0 < n2 < n
        for k in range(n2):
            o = hits2[k].str_off
            assert (hits2[k].pair, out2[o:o + len(texts[k][1]) + 1].tobytes()) == (texts[k][0], texts[k][1] + b"\0")


# ------------------------------------------------------------------ the command-line tools: --cigar / --cigarx ---
import subprocess

ROOT = Path(__file__).resolve().parent.parent
NW_BIN = ROOT / "seq-align_amd" / "bin" / "seqalign_nw"
SW_BIN = ROOT / "seq-align_amd" / "bin" / "seqalign_sw"


def run_cli(exe, *args, stdin=None):
    assert exe.exists(), f"{exe} missing: run make -C seq-align_amd"
    p = subprocess.run([str(exe), *args], input=stdin, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_cli_nw_cigar_lines(tmp_path):
    """seqalign_nw --cigar / --cigarx: one line per pair -- index, names, score, CIGAR -- against the reference's own known answer
    (README.md:65-74: CAGACGT x CGATA -> C-AGACGT / CGATA---, score -11) and the oracle on a FASTA file of 300 related pairs, in input
    order; a pair whose CIGAR outgrows the 64-byte slot (alternating gaps) sends the batch through the worst-case slots."""
    assert run_cli(NW_BIN, "--cigar", "CAGACGT", "CGATA") == "0\t*\t*\t-11\t1M1D3M3I\n"
    assert run_cli(NW_BIN, "--cigarx", "CAGACGT", "CGATA") == "0\t*\t*\t-11\t1=1D1=1X1=3I\n"
    osc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    batch = W.dna_nw_150(300, seed=31, related=True)
    pairs = [(batch.seq_a(p), batch.seq_b(p)) for p in range(300)]
    pairs[17] = (b"ACGT" * 40, b"TTTTTTTT")                       # mostly one long gap
    pairs[18] = (b"AC" * 60, b"AGC" * 40)
    fa = tmp_path / "pairs.fa"
    fa.write_text("".join(f">a{p} x\n{a.decode()}\n>b{p}\n{b.decode()}\n" for p, (a, b) in enumerate(pairs)))
    for flag, fmt in (("--cigar", M), ("--cigarx", EQX)):
        lines = run_cli(NW_BIN, flag, "--file", str(fa)).splitlines()
        assert len(lines) == 300
        for p, (a, b) in enumerate(pairs):
            rc, s, ra, rb = O.oracle_nw(osc, a, b)
            assert rc == 0 and lines[p] == f"{p}\ta{p} x\tb{p}\t{s}\t{py_cigar(ra, rb, fmt)}", (flag, p)
    # a CIGAR longer than its 64-byte slot: the second pass (worst-case slots) delivers it
    a, b = b"ACGTTGCA" * 30, (b"ACGTGCA" + b"ACGTTTGCA") * 15
    rc, s, ra, rb = O.oracle_nw(O.build_scoring({"init": [2, -3, 0, -1, 0, 0, 0, 0, 0, 0]}, "oracle"), a, b)
    want = py_cigar(ra, rb, M)
    assert len(want) > 64
    out = run_cli(NW_BIN, "--cigar", "--match", "2", "--mismatch", "-3", "--gapopen", "0", "--gapextend", "-1", a.decode(), b.decode())
    assert out == f"0\t*\t*\t{s}\t{want}\n"
    p = subprocess.run([str(NW_BIN), "--cigar", "--pretty", "A", "C"], capture_output=True, text=True)
    assert p.returncode != 0 and "--cigar" in p.stderr


def test_cli_sw_cigar_lines(tmp_path):
    """seqalign_sw --cigar: one line per hit -- pair, hit index, score, pos / len in both sequences, CIGAR -- against the oracle's hit
    lists (defaults 2 / -2 / -2 / -1, --minscore as the reference computes it, sw_cmdline.c:192-197), with --maxhits and without (the
    device path's cap of 16 hits is lifted through the per-pair API)."""
    osc = O.build_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    batch = W.dna_sw_read_vs_ref(40, seed=41, read_len=80, ref_len=300)
    pairs = [(batch.seq_a(p), batch.seq_b(p)) for p in range(40)]
    pairs.append((b"ACGTACGTAC" * 6, b"TT" + b"ACGTACGTAC" * 12))     # tandem repeat: many hits
    fa = tmp_path / "reads.fa"
    fa.write_text("".join(f">r{p}\n{a.decode()}\n>w{p}\n{b.decode()}\n" for p, (a, b) in enumerate(pairs)))
    for extra, max_hits, thr in ((("--maxhits", "3"), 3, None), ((), 1 << 30, None), (("--minscore", "6"), 1 << 30, 6)):
        lines = run_cli(SW_BIN, "--cigar", *extra, "--file", str(fa)).splitlines()
        want = []
        for p, (a, b) in enumerate(pairs):
            t = thr if thr is not None else W.default_minscore(2, len(a), len(b))
            rc, hits = O.oracle_sw(osc, a, b, t, max_hits)
            assert rc == 0
            want += [f"{p}\t{k}\t{h['score']}\t{h['pos_a']}\t{h['len_a']}\t{h['pos_b']}\t{h['len_b']}\t{py_cigar(h['a'], h['b'], M)}"
                     for k, h in enumerate(hits)]
        assert lines == want, (extra, len(lines), len(want))
    assert any(int(l.split("\t")[1]) >= 16 for l in run_cli(SW_BIN, "--cigar", "--minscore", "6", "--file", str(fa)).splitlines())
