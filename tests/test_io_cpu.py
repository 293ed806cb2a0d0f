"""CPU tier: scoring-file loaders and the sequence reader (include/seqalign_io.h),
the data formats either side of the hot path."""
import ctypes as C
import os
from pathlib import Path

import numpy as np
import pytest

import orclib as O
import seqalign_amd as S

libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]
LETTERS = "ARNDCQEGHILKMFPSTWYVBZX*"


def load(kind, path, case_sensitive=0, base=None):
    sc = base or S.make_scoring({"init": [1, -4, -10, -1, 0, 0, 0, 0, 0, case_sensitive]})
    f = libc.fopen(str(path).encode(), b"r")
    assert f
    err = C.create_string_buffer(256)
    fn = getattr(S.lib(), f"seqalign_scoring_load_{kind}")
    rc = fn(C.c_void_p(f), C.byref(sc), C.c_int(case_sensitive), err, C.c_size_t(256))
    libc.fclose(f)
    return rc, err.value.decode(), sc


def table_of(sc, letters=LETTERS.lower()):
    return [[sc.swap_scores[ord(a)][ord(b)] for b in letters] for a in letters]


def write_ncbi(path, sc, letters=LETTERS, sep=None):
    """Render the preset in the NCBI text layout (scoring/BLOSUM62.txt style)."""
    low = letters.lower()
    with open(path, "w") as f:
        f.write("#  Matrix rendered from scoring_system_BLOSUM62\n#  comment line\n")
        if sep is None:
            f.write("   " + "  ".join(letters) + "\n")
            for a, la in zip(letters, low):
                f.write(a + " " + " ".join(f"{sc.swap_scores[ord(la)][ord(lb)]:2d}" for lb in low) + "\n")
        else:
            f.write("".join(sep + c for c in letters) + "\n")
            for a, la in zip(letters, low):
                f.write(a + "".join(f"{sep}{sc.swap_scores[ord(la)][ord(lb)]}" for lb in low) + "\n")


@pytest.mark.parametrize("sep", [None, ",", "\t"])
def test_matrix_loader_round_trips_the_blosum62_preset(tmp_path, sep):
    preset = S.make_scoring({"preset": "BLOSUM62"})
    path = tmp_path / "m.txt"
    write_ncbi(path, preset, sep=sep if sep != "\t" else None)
    rc, err, sc = load("matrix", path)
    assert rc == 0, err
    assert table_of(sc) == table_of(preset)
    assert sc.min_penalty == -11 and sc.max_penalty == 11       # same as the preset (alignment_scoring.c:349-360)
    # identical lookups for every letter pair, upper and lower case
    s1, m1, s2, m2 = C.c_int(0), C.c_bool(False), C.c_int(0), C.c_bool(False)
    for a in LETTERS + LETTERS.lower():
        for b in LETTERS:
            S.lib().scoring_lookup(C.byref(sc), C.c_char(a.encode()), C.c_char(b.encode()), C.byref(s1), C.byref(m1))
            S.lib().scoring_lookup(C.byref(preset), C.c_char(a.encode()), C.c_char(b.encode()), C.byref(s2), C.byref(m2))
            assert (s1.value, m1.value) == (s2.value, m2.value)


def test_matrix_loader_case_sensitive_and_errors(tmp_path):
    p = tmp_path / "cs.txt"
    p.write_text("  A c\nA 1 -2\nc -3 4\n")
    rc, err, sc = load("matrix", p, case_sensitive=1)
    assert rc == 0 and sc.swap_scores[ord("A")][ord("c")] == -2 and sc.swap_scores[ord("c")][ord("A")] == -3
    rc, err, sc = load("matrix", p, case_sensitive=0)
    assert rc == 0 and sc.swap_scores[ord("a")][ord("c")] == -2
    for text, what in (("", "empty"), ("  A C\nA 1\n", "expected whitespace"), ("  A C\nA 1 x\n", "missing number"), ("  A C\nA 1 2 3\n", "too many"),
                       ("-A-C\nA-1-2\n", "separator"), ("  A C\n", "no rows")):
        p.write_text(text)
        rc, err, _ = load("matrix", p)
        assert rc == -1 and what.split()[0] in err.lower(), (text, err)


def test_pair_loader(tmp_path):
    p = tmp_path / "pairs.txt"
    p.write_text("# comment\nA C -3\nc,a,2\nG\tT\t 7\n\n")
    rc, err, sc = load("pairs", p)
    assert rc == 0, err
    assert sc.swap_scores[ord("a")][ord("c")] == -3 and sc.swap_scores[ord("c")][ord("a")] == 2
    assert sc.swap_scores[ord("g")][ord("t")] == 7 and sc.min_penalty == -11 and sc.max_penalty == 7
    p.write_text("A C x\n")
    assert load("pairs", p)[0] == -1
    p.write_text("# nothing\n")
    assert load("pairs", p)[0] == -1


@pytest.mark.skipif(not Path("/root/reference/scoring/BLOSUM62.txt").exists(), reason="reference data files absent")
def test_real_ncbi_files_load_and_blosum62_equals_the_preset():
    """Authoring container only: the reference's own scoring/ files parse, and its
    BLOSUM62.txt equals its compiled-in BLOSUM62 table."""
    preset = S.make_scoring({"preset": "BLOSUM62"})
    rc, err, sc = load("matrix", "/root/reference/scoring/BLOSUM62.txt")
    assert rc == 0, err
    assert table_of(sc) == table_of(preset)
    n = 0
    for path in sorted(Path("/root/reference/scoring").glob("*.txt")):
        rc, err, sc = load("matrix", path)
        assert rc == 0, (path.name, err)
        n += 1
    assert n >= 70


def read_all(path):
    lib = S.lib()
    lib.seqalign_reader_open.restype = C.c_void_p
    r = C.c_void_p(lib.seqalign_reader_open(str(path).encode()))
    assert r
    out = []
    name, seq, ln = C.c_char_p(), C.c_char_p(), C.c_size_t(0)
    while lib.seqalign_reader_next(r, C.byref(name), C.byref(seq), C.byref(ln)):
        out.append((name.value.decode(), seq.value.decode(), ln.value))
    lib.seqalign_reader_close(r)
    return out


def test_sequence_reader_fasta_fastq_plain(tmp_path):
    p = tmp_path / "dna.fa"
    # README.md:79-88: multi-line FASTA records
    p.write_text(">seqA\nACAATAGAC\n>seqB\nACGAATAGAT\n>seqC\nACGTGA\nCAGAT\n>seqD\nGTGGACG\nAGTA\n")
    assert read_all(p) == [(">seqA", "ACAATAGAC", 9), (">seqB", "ACGAATAGAT", 10),
                           (">seqC", "ACGTGACAGAT", 11), (">seqD", "GTGGACGAGTA", 11)]
    p.write_text("@r1\nACGT\n+\nIIII\n@r2 desc\nTTGA\n+r2\n@@II\n")
    assert read_all(p) == [("@r1", "ACGT", 4), ("@r2 desc", "TTGA", 4)]
    # a FASTQ file that ends right after a sequence line (truncated record): no phantom record follows
    p.write_text("@r1\nACGT\n+\nIIII\n@r2\nTTGA\n")
    assert read_all(p) == [("@r1", "ACGT", 4), ("@r2", "TTGA", 4)]
    p.write_text("@r1\nACGT")
    assert read_all(p) == [("@r1", "ACGT", 4)]
    p.write_text("ACGT\n\nTTGCA\r\nGG")
    assert read_all(p) == [("", "ACGT", 4), ("", "TTGCA", 5), ("", "GG", 2)]
    # gzip-compressed input, as the reference reads through zlib (alignment_cmdline.c / seq_file)
    import gzip
    gz = tmp_path / "dna.fa.gz"
    with gzip.open(gz, "wt") as f:
        f.write(">seqA\nACAATAGAC\n>seqB\nACGAATAGAT\n" + ">long\n" + "ACGT" * 50000 + "\n")
    got = read_all(gz)
    assert got[:2] == [(">seqA", "ACAATAGAC", 9), (">seqB", "ACGAATAGAT", 10)]
    assert got[2][0] == ">long" and got[2][2] == 200000 and got[2][1] == "ACGT" * 50000


def test_sequence_reader_lines_across_block_boundaries(tmp_path):
    """The reader takes its lines out of 4 MiB blocks (host/sa_io.c: read_line_block): records whose lines straddle a block's
    end, a line longer than a block, CR LF ends, blank lines and a last line without a newline -- against a parse in Python;
    plain and gzip-compressed."""
    import gzip
    import random
    rnd = random.Random(11)
    recs, parts = [], []
    big = "".join(rnd.choice("ACGT") for _ in range(1 << 16)) * 80          # 5 MiB in one line
    for k in range(3000):
        name = ">r%d %s" % (k, "x" * rnd.randrange(0, 40))
        if k == 1500:
            lines = [big]
        else:
            lines = ["".join(rnd.choice("ACGTN") for _ in range(rnd.randrange(1, 9000))) for _ in range(rnd.randrange(1, 4))]
        eol = "\r\n" if k % 7 == 0 else "\n"
        parts.append(name + eol + eol.join(lines) + eol + ("\n" if k % 11 == 0 else ""))
        recs.append((name, "".join(lines)))
    text = "".join(parts)
    text = text[:-1] if text.endswith("\n") else text                       # no newline at the end of the file
    assert len(text) > 3 * (4 << 20)
    p = tmp_path / "big.fa"
    p.write_text(text, newline="")
    got = read_all(p)
    assert len(got) == len(recs)
    assert all(g == (n, s, len(s)) for g, (n, s) in zip(got, recs))
    gz = tmp_path / "big.fa.gz"
    with gzip.open(gz, "wt", newline="", compresslevel=1) as f:
        f.write(text)
    assert read_all(gz) == got
