"""GPU tier: the batch command-line front-end (seq-align_amd/bin/seqalign_nw,
seqalign_sw) against the reference's documented output text (README.md) and the
oracle.  The Perl wrappers' regular expressions (perl/SmithWaterman.pm:239-275,
perl/NeedlemanWunsch.pm) define what the text must look like to a consumer."""
import re
import subprocess
from pathlib import Path

import pytest

import orclib as O
from seqalign_amd import workloads as W

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
NW = ROOT / "seq-align_amd" / "bin" / "seqalign_nw"
SW = ROOT / "seq-align_amd" / "bin" / "seqalign_sw"


def run(exe, *args, stdin=None):
    assert exe.exists(), f"{exe} missing: run make -C seq-align_amd"
    p = subprocess.run([str(exe), *args], input=stdin, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    return p.stdout


def spacer(a, b):
    return "".join(" " if "-" in (x, y) else "|" if x.lower() == y.lower() else "*" for x, y in zip(a, b))


def test_readme_basic_examples():
    # README.md:65-74
    assert run(NW, "CAGACGT", "CGATA") == "C-AGACGT\nCGATA---\n\n"
    assert run(NW, "--printscores", "CAGACGT", "CGATA") == "C-AGACGT\nCGATA---\nscore: -11\n\n"


def test_readme_printmatrices_text():
    """README.md:118-145: the three matrices, the parameter line, then the alignment."""
    out = run(NW, "--printmatrices", "ACAGGT", "AAGGT")
    want = """seq_a: ACAGGT
seq_b: AAGGT
match_scores:
  0:    0 -2147483643 -2147483643 -2147483643 -2147483643 -2147483643 -2147483643
  1:  -2147483643   1  -7  -5  -9 -10 -11
  2:  -2147483643  -4  -1  -3  -7  -8  -9
  3:  -2147483643  -8  -6  -3  -2  -6 -10
  4:  -2147483643  -9  -7  -8  -2  -1  -8
  5:  -2147483643 -10  -8  -9 -10  -4   0
gap_a_scores:
  0:    0 -2147483643 -2147483643 -2147483643 -2147483643 -2147483643 -2147483643
  1:   -5 -10 -11 -12 -13 -14 -15
  2:   -6  -4  -9 -10 -11 -12 -13
  3:   -7  -5  -6  -8 -12 -13 -14
  4:   -8  -6  -7  -8  -7 -11 -13
  5:   -9  -7  -8  -9  -7  -6 -11
gap_b_scores:
  0:    0  -5  -6  -7  -8  -9 -10
  1:  -2147483643 -10  -4  -5  -6  -7  -8
  2:  -2147483643 -11  -9  -6  -7  -8  -9
  3:  -2147483643 -12 -10 -11  -8  -7  -8
  4:  -2147483643 -13 -11 -12 -13  -7  -6
  5:  -2147483643 -14 -12 -13 -14 -12  -9
match: 1 mismatch: -2 gapopen: -4 gapexend: -1

ACAGGT
A-AGGT
"""
    assert out.split() == want.split()                    # README flattens the tabs
    assert "  1:\t-2147483643\t  1\t -7" in out           # the real separators are tabs (alignment.c:368-373)
    assert out.endswith("\nACAGGT\nA-AGGT\n\n")


def test_fasta_file_stdin_and_formats(tmp_path):
    fa = tmp_path / "dna.fa"       # README.md:79-88
    fa.write_text(">seqA\nACAATAGAC\n>seqB\nACGAATAGAT\n>seqC\nACGTGA\nCAGAT\n>seqD\nGTGGACG\nAGTA\n")
    want = "AC-AATAGAC\nACGAATAGAT\nscore: 1\n\nACGTGAC-AGAT\nGTG-GACGAGTA\nscore: -12\n\n"   # scores: SURVEY A.3-1
    assert run(NW, "--printscores", "--file", str(fa)) == want
    assert run(NW, "--printscores", "--stdin", stdin=fa.read_text()) == want
    assert run(NW, "--printscores", "--file", "-", stdin=fa.read_text()) == want
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    a.write_text("ACAATAGAC\nACGTGACAGAT\n"); b.write_text("ACGAATAGAT\nGTGGACGAGTA\n")
    assert run(NW, "--printscores", "--files", str(a), str(b)) == want
    out = run(NW, "--pretty", "--printfasta", "--file", str(fa))
    assert out == (">seqA\n>seqB\nAC-AATAGAC\n|| ||||||*\nACGAATAGAT\n\n"
                   ">seqC\n>seqD\nACGTGAC-AGAT\n" + spacer("ACGTGAC-AGAT", "GTG-GACGAGTA") + "\nGTG-GACGAGTA\n\n")
    out = run(NW, "--printfasta", "--file", str(fa))
    assert out.startswith(">seqA\nAC-AATAGAC\n>seqB\nACGAATAGAT\n\n")
    col = run(NW, "--colour", "ACGT", "AGT")
    assert "\033[91m" in col and "\033[0m" in col


def test_nw_options_match_oracle(tmp_path):
    batch = W.ragged(300, seed=5, max_len=80, lower_frac=0.2, extra=b"N")
    f = tmp_path / "pairs.txt"
    f.write_text("".join(f"{batch.seq_a(p).decode() or 'A'}\n{batch.seq_b(p).decode() or 'C'}\n" for p in range(300)))
    pairs = [((batch.seq_a(p) or b"A"), (batch.seq_b(p) or b"C")) for p in range(300)]
    for args, spec in (
            ([], {"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}),
            (["--freestartgap", "--freeendgap", "--wildcard", "N", "0"],
             {"init": [1, -2, -4, -1, 1, 1, 0, 0, 0, 0], "wildcards": [["N", 0]]}),
            (["--match", "2", "--mismatch", "-3", "--gapopen", "-5", "--gapextend", "-2", "--nogapsin1"],
             {"init": [2, -3, -5, -2, 0, 0, 1, 0, 0, 0]}),
            (["--scoring", "BLOSUM62", "--case_sensitive"], None)):
        out = run(NW, "--printscores", *args, "--file", str(f))
        blocks = out.strip("\n").split("\n\n")
        assert len(blocks) == 300
        if spec is None:
            import seqalign_amd as S
            sc = O.Scoring.from_buffer_copy(bytes(S.make_scoring({"preset": "BLOSUM62"})))
        else:
            sc = O.build_scoring(spec, "oracle")
            sc.min_penalty = min(sc.min_penalty, -5)      # the CLI starts from the default scoring's range
        for (a, b), blk in zip(pairs, blocks):
            rc, score, ra, rb = O.oracle_nw(sc, a, b)
            assert rc == 0 and blk == f"{ra.decode()}\n{rb.decode()}\nscore: {score}", (args, a, b)


def test_plain_c_batch_example():
    """seq-align_amd/examples/batch_example.c: the batch C-ABI from plain C (README.md:79-88's pairs)."""
    exe = ROOT / "seq-align_amd" / "bin" / "batch_example"
    want = "AC-AATAGAC\nACGAATAGAT\nscore: 1\n\nACGTGAC-AGAT\nGTG-GACGAGTA\nscore: -12\n\nC-AGACGT\nCGATA---\nscore: -11\n\n"
    assert run(exe) == want


def test_zam_output():
    """--zam (nw_cmdline.c:36-76): '_' for gaps, spacer line, mismatch and indel counts."""
    sc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    a, b = b"ACGTTTGACCA", b"ACGATTGCA"
    rc, score, ra, rb = O.oracle_nw(sc, a, b)
    ra, rb = ra.decode(), rb.decode()
    sp = "".join(" " if "-" in (x, y) else "|" if x.lower() == y.lower() else "*" for x, y in zip(ra, rb))
    want = f"Br1:{ra.replace('-', '_')}\n    {sp}\nBr2:{rb.replace('-', '_')}\n{sp.count('*')} {sp.count(' ')}\n\n"
    assert run(NW, "--zam", a.decode(), b.decode()) == want


def test_long_pair_takes_the_strip_pipeline(tmp_path):
    """One pair much longer than a wave's row (3 000 x 2 700, related): the fill runs as a
    pipeline of column strips (sa_fill_strips.hip); alignment and score equal the oracle's."""
    batch = W.dna_nw_150(1, seed=31, length=3000, related=True)
    a, b = batch.seq_a(0), batch.seq_b(0)[:2700]
    f = tmp_path / "long.txt"
    f.write_text(f"{a.decode()}\n{b.decode()}\n")
    sc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    rc, score, ra, rb = O.oracle_nw(sc, a, b)
    assert rc == 0
    assert run(NW, "--printscores", "--file", str(f)) == f"{ra.decode()}\n{rb.decode()}\nscore: {score}\n\n"


def expected_sw_text(index, a, b, hits, context=0, pretty=False):
    out = [f"== Alignment {index} lengths ({len(a)}, {len(b)}):", ""]
    for k, h in enumerate(hits):
        out.append(f"hit {index}.{k} score: {h['score']}")
        rem_a, rem_b = len(a) - (h["pos_a"] + h["len_a"]), len(b) - (h["pos_b"] + h["len_b"])
        cl = min(max(h["pos_a"], h["pos_b"]), context)
        cr = min(max(rem_a, rem_b), context)
        ls_a, ls_b = max(cl - h["pos_a"], 0), max(cl - h["pos_b"], 0)
        rs_a, rs_b = max(cr - rem_a, 0), max(cr - rem_b, 0)

        def part(s, pos, ln, whole, ls, rs):
            return ("  " + " " * ls + whole[pos - (cl - ls):pos] + s + whole[pos + ln:pos + ln + (cr - rs)]
                    + " " * rs + f"  [pos: {pos}; len: {ln}]")
        out.append(part(h["a"], h["pos_a"], h["len_a"], a, ls_a, rs_a))
        if pretty:
            ml, mr = max(ls_a, ls_b), max(rs_a, rs_b)
            out.append("  " + " " * ml + "." * (cl - ml) + spacer(h["a"], h["b"]) + "." * (cr - mr) + " " * mr)
        out.append(part(h["b"], h["pos_b"], h["len_b"], b, ls_b, rs_b))
        out.append("")
    out.append("==")
    return "\n".join(out) + "\n"


def test_sw_cli_text_and_perl_wrapper_grammar(tmp_path):
    sc = O.build_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}, "oracle")   # sw_cmdline.c:37-46
    batch = W.dna_sw_read_vs_ref(40, seed=8, read_len=60, ref_len=200)
    f = tmp_path / "reads.fa"
    f.write_text("".join(f">r{p}\n{batch.seq_a(p).decode()}\n>ref{p}\n{batch.seq_b(p).decode()}\n" for p in range(40)))
    for extra, ctx, pretty in (([], 0, False), (["--pretty"], 0, True), (["--context", "5", "--pretty"], 5, True),
                               (["--maxhits", "2", "--context", "300"], 300, False)):
        out = run(SW, *extra, "--file", str(f))
        want = ""
        for p in range(40):
            a, b = batch.seq_a(p), batch.seq_b(p)
            thr = W.default_minscore(2, len(a), len(b))
            rc, hits = O.oracle_sw(sc, a, b, thr, 2 if "--maxhits" in extra else 1 << 30)
            want += expected_sw_text(p, a.decode(), b.decode(), hits, ctx, pretty)
        assert out == want, extra
    # what perl/SmithWaterman.pm:239-275 expects to parse from `--pretty`
    out = run(SW, "--pretty", "--minscore", "20", batch.seq_a(0).decode(), batch.seq_b(0).decode()).split("\n")
    assert out[0].startswith("== Alignment 0 lengths (60, 200):")
    i = 2
    n_hits = 0
    while not out[i].startswith("=="):
        assert re.match(r"^hit \d+\.(\d+) score: (\d+)$", out[i])
        assert re.match(r"^  (.*)  \[pos: (\d+); len: (\d+)\]$", out[i + 1])
        assert re.match(r"^  ([\|\* ]+)$", out[i + 2])
        assert re.match(r"^  (.*)  \[pos: (\d+); len: (\d+)\]$", out[i + 3])
        assert out[i + 4] == ""
        i += 5
        n_hits += 1
    assert n_hits >= 1


def _matrix_text(letters, score):
    """A substitution-matrix file as alignment_scoring_load.c:39-220 reads it: header line of column
    characters, then one row per character."""
    return ("# test matrix\n   " + "  ".join(letters) + "\n"
            + "".join(a + " " + " ".join(str(score(a, b)) for b in letters) + "\n" for a in letters))


def _sw_blocks(out):
    """The CLI's text, one string per pair (each block ends with the '==' line)."""
    return [b + "==\n" for b in out.split("==\n") if b]


def test_sw_cli_substitution_matrix_file_vs_oracle(tmp_path):
    """SURVEY 8f-4 (alignment_scoring_load.c:39-220 + the fill it feeds): scores loaded from a matrix FILE
    by the CLI give the hits of an oracle scoring built independently from the same numbers -- (1) the
    BLOSUM62 table rendered from tests/golden/presets.json (values extracted from the compiled reference),
    (2) an asymmetric made-up DNA matrix with large entries, upper-case file vs mixed-case sequences."""
    import json
    spec62 = json.loads((ROOT / "tests" / "golden" / "presets.json").read_text())["BLOSUM62"]["spec"]
    table = {(a, b): s for a, b, s in spec62["mutations"]}
    letters = "ARNDCQEGHILKMFPSTWYVBZX*"
    m = tmp_path / "b62.txt"
    m.write_text(_matrix_text(letters, lambda a, b: table[(a.lower(), b.lower())]))
    batch = W.protein_sw_300(12, seed=4, length=80)
    f = tmp_path / "prot.txt"
    f.write_text("".join(f"{batch.seq_a(p).decode()}\n{batch.seq_b(p).decode()}\n" for p in range(12)))
    # the CLI starts from the SW defaults 2/-2/-2/-1 (sw_cmdline.c:37-46); a matrix without --match switches
    # the match/mismatch fallback off (alignment_cmdline.c: use_match_mismatch = 0)
    osc = O.build_scoring({"init": [2, -2, -10, -1, 0, 0, 0, 0, 0, 0], "mutations": spec62["mutations"],
                           "use_match_mismatch": 0}, "oracle")
    for maxhits in (1, 3, 1 << 20):
        out = run(SW, "--substitution_matrix", str(m), "--gapopen", "-10", "--gapextend", "-1", "--minscore", "15",
                  *( ["--maxhits", str(maxhits)] if maxhits < (1 << 20) else []), "--file", str(f))
        blocks = _sw_blocks(out)
        assert len(blocks) == 12
        for p, blk in enumerate(blocks):
            a, b = batch.seq_a(p), batch.seq_b(p)
            rc, hits = O.oracle_sw(osc, a, b, 15, maxhits)
            assert rc == 0 and blk == expected_sw_text(p, a.decode(), b.decode(), hits), (maxhits, p)

    def dna_score(a, b):
        return 37 if a == b else -((ord(a) * 5 + ord(b) * 3) % 9) - 1   # asymmetric: (A,C) != (C,A)
    m2 = tmp_path / "dna.txt"
    m2.write_text(_matrix_text("ACGTN", dna_score))
    rb = W.ragged(40, seed=12, max_len=90, alphabet=b"ACGT", lower_frac=0.3, extra=b"N")
    pairs = [((rb.seq_a(p) or b"A"), (rb.seq_b(p) or b"c")) for p in range(40)]
    f2 = tmp_path / "dna_pairs.txt"
    f2.write_text("".join(f"{a.decode()}\n{b.decode()}\n" for a, b in pairs))
    muts = [[a.lower(), b.lower(), dna_score(a, b)] for a in "ACGTN" for b in "ACGTN"]
    osc = O.build_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0], "mutations": muts, "use_match_mismatch": 0}, "oracle")
    blocks = _sw_blocks(run(SW, "--substitution_matrix", str(m2), "--minscore", "40", "--file", str(f2)))
    assert len(blocks) == 40
    for p, ((a, b), blk) in enumerate(zip(pairs, blocks)):
        rc, hits = O.oracle_sw(osc, a, b, 40)
        assert rc == 0 and blk == expected_sw_text(p, a.decode(), b.decode(), hits), p
    # the same file through the global aligner: score + strings (min_penalty follows the loaded scores)
    onw = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0], "mutations": muts, "use_match_mismatch": 0}, "oracle")
    blocks = run(NW, "--printscores", "--substitution_matrix", str(m2), "--file", str(f2)).strip("\n").split("\n\n")
    assert len(blocks) == 40
    for (a, b), blk in zip(pairs, blocks):
        rc, score, ra, rb_ = O.oracle_nw(onw, a, b)
        assert rc == 0 and blk == f"{ra.decode()}\n{rb_.decode()}\nscore: {score}", (a, b)


def test_cli_substitution_pairs_file_vs_oracle(tmp_path):
    """alignment_scoring_load.c:223-306: a pair list ("a b score" per line, or with a one-character
    separator) on top of --match/--mismatch; pairs not listed fall back to match/mismatch."""
    listed = [["a", "c", -1], ["c", "a", 4], ["g", "t", 1], ["t", "t", 7], ["n", "a", 0], ["a", "n", 0]]
    f1 = tmp_path / "pairs_ws.txt"
    f1.write_text("# pairs\n" + "".join(f"{a.upper()} {b.upper()} {s}\n" for a, b, s in listed))
    f2 = tmp_path / "pairs_sep.txt"
    f2.write_text("".join(f"{a},{b},{s}\n" for a, b, s in listed))
    rb = W.ragged(60, seed=13, max_len=70, alphabet=b"ACGT", lower_frac=0.2, extra=b"N")
    pairs = [((rb.seq_a(p) or b"T"), (rb.seq_b(p) or b"t")) for p in range(60)]
    seqs = tmp_path / "seqs.txt"
    seqs.write_text("".join(f"{a.decode()}\n{b.decode()}\n" for a, b in pairs))
    osw = O.build_scoring({"init": [3, -3, -2, -1, 0, 0, 0, 0, 0, 0], "mutations": listed}, "oracle")
    onw = O.build_scoring({"init": [3, -3, -4, -1, 0, 0, 0, 0, 0, 0], "mutations": listed}, "oracle")
    for pf in (f1, f2):
        blocks = _sw_blocks(run(SW, "--substitution_pairs", str(pf), "--match", "3", "--mismatch", "-3", "--minscore", "12",
                                "--file", str(seqs)))
        assert len(blocks) == 60
        for p, ((a, b), blk) in enumerate(zip(pairs, blocks)):
            rc, hits = O.oracle_sw(osw, a, b, 12)
            assert rc == 0 and blk == expected_sw_text(p, a.decode(), b.decode(), hits), (pf.name, p)
        blocks = run(NW, "--printscores", "--substitution_pairs", str(pf), "--match", "3", "--mismatch", "-3",
                     "--file", str(seqs)).strip("\n").split("\n\n")
        for (a, b), blk in zip(pairs, blocks):
            rc, score, ra, rb_ = O.oracle_nw(onw, a, b)
            assert rc == 0 and blk == f"{ra.decode()}\n{rb_.decode()}\nscore: {score}", (pf.name, a, b)


def test_pipeline_keeps_input_order_across_batches(tmp_path):
    """Round 5: the tools run as a pipeline (reader | GPU | printer) over a ring of three batches of 65 536 pairs.  150 000 short
    pairs cross two batch boundaries and reuse the first batch's buffers: every alignment must come out in input order, with
    its own FASTA name, and equal the oracle's (reference driver: src/alignment_cmdline.c:578-640, one pair at a time)."""
    rng = W.Rng(515)
    n = 150_000
    lens = 8 + rng.below(12, 2 * n).astype(int)
    bases = rng.below(4, int(lens.sum()))
    seqs, at = [], 0
    for ln in lens:
        seqs.append(bytes(b"ACGT"[i] for i in bases[at:at + ln]).decode())
        at += int(ln)
    fa = tmp_path / "pairs.fa"
    with open(fa, "w") as f:
        for k, s in enumerate(seqs):
            f.write(f">r{k}\n{s}\n")
    out = run(NW, "--printfasta", "--printscores", "--file", str(fa))
    recs = out.strip("\n").split("\n\n")
    assert len(recs) == n
    osc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    for p in list(range(0, n, 997)) + [65535, 65536, 131071, 131072, n - 1]:
        name_a, ra, name_b, rb, sc_line = recs[p].split("\n")      # nw_cmdline.c:78-149 with --printfasta --printscores
        assert (name_a, name_b) == (f">r{2 * p}", f">r{2 * p + 1}"), (p, name_a, name_b)
        rc, s_, wa, wb = O.oracle_nw(osc, seqs[2 * p].encode(), seqs[2 * p + 1].encode())
        assert rc == 0 and (ra, rb, sc_line) == (wa.decode(), wb.decode(), f"score: {s_}"), p
    # the local tool through the same pipeline: hit headings count alignments in input order
    out = run(SW, "--minscore", "12", "--maxhits", "1", "--file", str(fa))
    idx = [int(m) for m in re.findall(r"^== Alignment (\d+) lengths", out, re.M)]
    assert idx == list(range(n))
