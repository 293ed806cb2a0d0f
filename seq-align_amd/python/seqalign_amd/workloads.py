"""Deterministic synthetic batches for the BASELINE.json configs.

Own PRNG (splitmix64, vectorised with numpy uint64 wrap-around) so that the same
seed gives the same bytes on every numpy version and on both boxes.

A batch is the host-side picture of what the C-ABI consumes
(include/seqalign_hip.h `seqalign_batch_t`): one byte arena holding every
sequence, plus per-pair (off_a, len_a, off_b, len_b).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n successive outputs of splitmix64 started at `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Rng:
    """Small stream wrapper over splitmix64 blocks."""

    def __init__(self, seed: int):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.counter = 0

    def u64(self, n: int) -> np.ndarray:
        base = (self.seed + self.counter * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        self.counter += n
        return splitmix64(base, n)

    def below(self, bound: int, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)) % np.uint64(bound)

    def unit(self, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
AMINO20 = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)


@dataclass
class Batch:
    arena: np.ndarray   # uint8, every sequence back to back
    off_a: np.ndarray   # uint64 [n]
    len_a: np.ndarray   # uint32 [n]
    off_b: np.ndarray   # uint64 [n]
    len_b: np.ndarray   # uint32 [n]

    @property
    def n_pairs(self) -> int:
        return int(self.len_a.shape[0])

    def cells(self) -> int:
        """GCUPS numerator: sum of len_a*len_b (interior DP cells)."""
        return int((self.len_a.astype(np.int64) * self.len_b.astype(np.int64)).sum())

    def matrix_cells(self) -> np.ndarray:
        return (self.len_a.astype(np.int64) + 1) * (self.len_b.astype(np.int64) + 1)

    def seq_a(self, p: int) -> bytes:
        o, n = int(self.off_a[p]), int(self.len_a[p])
        return self.arena[o:o + n].tobytes()

    def seq_b(self, p: int) -> bytes:
        o, n = int(self.off_b[p]), int(self.len_b[p])
        return self.arena[o:o + n].tobytes()

    def slice(self, lo: int, hi: int) -> "Batch":
        return Batch(self.arena, self.off_a[lo:hi].copy(), self.len_a[lo:hi].copy(),
                     self.off_b[lo:hi].copy(), self.len_b[lo:hi].copy())

    def shard(self, rank: int, world: int) -> "Batch":
        """Contiguous pair-index block for one rank, cut at equal DP CELLS (balance
        ragged batches by W*H, keep pair order); no collective."""
        return self.slice(*shard_range_cells(self.matrix_cells(), rank, world))


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Pairs [lo, hi) owned by `rank` when every pair costs the same: block g gets [g*n/G, (g+1)*n/G)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_edges_cells(cells, world: int) -> list[int]:
    """world+1 pair indices cutting a batch into contiguous blocks of (nearly) equal total cells:
    edge k is the pair boundary nearest to k/world of the cell total (ties: the later boundary).
    The same rule as for_each_shard in seq-align_amd/csrc/sa_multi.hip."""
    cells = np.asarray(cells, dtype=np.uint64)
    n = int(cells.shape[0])
    cum = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cells, dtype=np.uint64)])
    total = int(cum[-1])
    edges = [0]
    for k in range(1, world):
        t = total * k // world
        i = int(np.searchsorted(cum, np.uint64(t), side="left"))
        if i > 0 and t - int(cum[i - 1]) < int(cum[min(i, n)]) - t:
            i -= 1
        edges.append(max(min(i, n), edges[-1]))
    edges.append(n)
    return edges


def shard_range_cells(cells, rank: int, world: int) -> tuple[int, int]:
    e = shard_edges_cells(cells, world)
    return e[rank], e[rank + 1]


def from_pairs(pairs: list[tuple[bytes, bytes]]) -> Batch:
    chunks, off_a, len_a, off_b, len_b = [], [], [], [], []
    pos = 0
    for a, b in pairs:
        off_a.append(pos); len_a.append(len(a)); chunks.append(a); pos += len(a)
        off_b.append(pos); len_b.append(len(b)); chunks.append(b); pos += len(b)
    arena = np.frombuffer(b"".join(chunks) + b"\0", dtype=np.uint8).copy()
    return Batch(arena, np.asarray(off_a, np.uint64), np.asarray(len_a, np.uint32),
                 np.asarray(off_b, np.uint64), np.asarray(len_b, np.uint32))


def _fixed_batch(a_mat: np.ndarray, b_mat: np.ndarray) -> Batch:
    n, la = a_mat.shape
    lb = b_mat.shape[1]
    stride = la + lb
    arena = np.empty(n * stride + 1, dtype=np.uint8)
    view = arena[:-1].reshape(n, stride)
    view[:, :la] = a_mat
    view[:, la:] = b_mat
    arena[-1] = 0
    off_a = np.arange(n, dtype=np.uint64) * np.uint64(stride)
    return Batch(arena, off_a, np.full(n, la, np.uint32),
                 off_a + np.uint64(la), np.full(n, lb, np.uint32))


def _mutate(src: np.ndarray, out_len: int, alphabet: np.ndarray, rng: Rng,
            p_sub: float, p_indel: float) -> np.ndarray:
    """Per-row substitutions + short indels, result forced to out_len columns."""
    n, L = src.shape
    k = len(alphabet)
    sub = rng.unit(n * L).reshape(n, L) < p_sub
    repl = alphabet[rng.below(k, n * L).astype(np.int64)].reshape(n, L)
    mutated = np.where(sub, repl, src)
    out = np.empty((n, out_len), dtype=np.uint8)
    ind = rng.unit(n * L).reshape(n, L)
    ind_len = 1 + rng.below(3, n * L).reshape(n, L).astype(np.int64)
    pad = alphabet[rng.below(k, n * out_len).astype(np.int64)].reshape(n, out_len)
    for r in range(n):
        row = mutated[r]
        events = np.nonzero(ind[r] < p_indel)[0]
        if len(events) == 0:
            res = row
        else:
            pieces, prev = [], 0
            for e in events:
                pieces.append(row[prev:e])
                if (int(ind_len[r, e]) + e) & 1:       # deletion
                    prev = min(L, e + int(ind_len[r, e]))
                else:                                    # insertion
                    pieces.append(pad[r, :int(ind_len[r, e])])
                    prev = e
            pieces.append(row[prev:])
            res = np.concatenate(pieces)
        if len(res) >= out_len:
            out[r] = res[:out_len]
        else:
            out[r, :len(res)] = res
            out[r, len(res):] = pad[r, :out_len - len(res)]
    return out


def dna_nw_150(n_pairs: int, seed: int = 1, related: bool = False,
               length: int = 150) -> Batch:
    """C2 / C5: DNA `length` x `length`, iid uniform ACGT (or b = mutated a)."""
    rng = Rng(seed)
    a = DNA[rng.below(4, n_pairs * length).astype(np.int64)].reshape(n_pairs, length)
    if related:
        b = _mutate(a, length, DNA, rng, 0.05, 0.01)
    else:
        b = DNA[rng.below(4, n_pairs * length).astype(np.int64)].reshape(n_pairs, length)
    return _fixed_batch(a, b)


def dna_nw_indexed(first: int, n_pairs: int, seed: int = 5, length: int = 150) -> Batch:
    """C5: pairs [first, first + n_pairs) of an unbounded stream of iid ACGT pairs.  Pair p draws
    its 2*length letters from splitmix64 counter positions [p*2*length, (p+1)*2*length), so a
    rank generates exactly its own shard of the 1 M-pair batch (contiguous pair-index
    blocks) without materialising the rest, and every shard size sees the same pair p."""
    rng = Rng(seed)
    rng.counter = first * 2 * length
    m = DNA[rng.below(4, n_pairs * 2 * length).astype(np.int64)].reshape(n_pairs, 2 * length)
    return _fixed_batch(m[:, :length], m[:, length:])


def dna_sw_read_vs_ref(n_pairs: int, seed: int = 2, read_len: int = 150,
                       ref_len: int = 1000) -> Batch:
    """C3: a = read (150) cut from b = ref (1000) with 5% subs + 1% indels."""
    rng = Rng(seed)
    ref = DNA[rng.below(4, n_pairs * ref_len).astype(np.int64)].reshape(n_pairs, ref_len)
    start = rng.below(ref_len - read_len - 8, n_pairs).astype(np.int64)
    cols = start[:, None] + np.arange(read_len + 8)[None, :]
    window = np.take_along_axis(ref, cols, axis=1)
    read = _mutate(window, read_len, DNA, rng, 0.05, 0.01)
    return _fixed_batch(read, ref)


def protein_sw_300(n_pairs: int, seed: int = 3, length: int = 300) -> Batch:
    """C4: a iid over 20 amino acids, b = a with 40% subs + 2% indels."""
    rng = Rng(seed)
    a = AMINO20[rng.below(20, n_pairs * length).astype(np.int64)].reshape(n_pairs, length)
    b = _mutate(a, length, AMINO20, rng, 0.40, 0.02)
    return _fixed_batch(a, b)


def ragged(n_pairs: int, seed: int, max_len: int, alphabet: bytes = b"ACGT",
           lower_frac: float = 0.0, extra: bytes = b"") -> Batch:
    """Parity-test batch: lengths 0..max_len, optional lower case / extra chars."""
    rng = Rng(seed)
    alpha = np.frombuffer(alphabet + extra, dtype=np.uint8)
    la = rng.below(max_len + 1, n_pairs).astype(np.int64)
    lb = rng.below(max_len + 1, n_pairs).astype(np.int64)
    pairs = []
    for p in range(n_pairs):
        seqs = []
        for n in (int(la[p]), int(lb[p])):
            s = alpha[rng.below(len(alpha), n).astype(np.int64)] if n else np.empty(0, np.uint8)
            if lower_frac > 0 and n:
                low = rng.unit(n) < lower_frac
                is_up = (s >= 65) & (s <= 90)
                s = np.where(low & is_up, s + 32, s).astype(np.uint8)
            seqs.append(s.tobytes())
        pairs.append((seqs[0], seqs[1]))
    return from_pairs(pairs)


def make(gen: str, n_pairs: int, kwargs: dict) -> Batch:
    """Generator by name, as the golden fixtures record it (tests/golden/configs.json)."""
    kw = dict(kwargs)
    if gen == "dna_nw_indexed":
        return dna_nw_indexed(kw.pop("first", 0), n_pairs, **kw)
    return globals()[gen](n_pairs, **kw)


def default_minscore(match: int, len_a: int, len_b: int) -> int:
    """sw_cmdline.c:192-197: match * MAX2(0.2 * MIN2(len_a,len_b), 2), as int."""
    return int(match * max(0.2 * min(len_a, len_b), 2))
