"""GPU tier: asynchronous host-level calls (seqalign_{nw,sw}_batch_submit + seqalign_job_wait).

A job IS the synchronous call run on a lane (a context of its own on the same device), so its results must be the synchronous
call's bit for bit -- which in turn are the reference's (src/needleman_wunsch.c:34-146, src/smith_waterman.c:137-277; checked
against the golden alignments and the oracle here too, not only against the synchronous path).
"""
import json
import threading
from pathlib import Path

import numpy as np
import pytest

import orclib as O
import seqalign_amd as S
from seqalign_amd import workloads as W

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


def load(name):
    return json.loads((GOLDEN / name).read_text())


def test_interleaved_submits_from_two_threads_equal_the_synchronous_calls():
    """Two threads submit NW and SW batches of different shapes to ONE context, interleaved, twelve jobs each, waiting two jobs
    behind their submits (what a streaming caller does); every job's result equals the synchronous call's on the same batch, and
    the golden C2 pairs / an oracle sample besides.  Jobs run with the options their context had at submit time."""
    cfg = load("configs.json")["C2_related"]
    nw_sc = S.make_scoring(cfg["scoring"])
    sw_spec = {"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}
    sw_sc, sw_osc = S.make_scoring(sw_spec), O.build_scoring(sw_spec, "oracle")
    nw_batches = [W.make(cfg["gen"], cfg["n"], cfg["kwargs"]), W.dna_nw_150(3000, seed=71, related=True), W.ragged(500, seed=72, max_len=120),
                  W.dna_nw_150(1500, seed=73, length=90)]
    sw_batches = [W.dna_sw_read_vs_ref(400, seed=74, read_len=100, ref_len=400), W.dna_sw_read_vs_ref(1200, seed=75, read_len=60, ref_len=250)]
    with S.Context(0) as sync:
        nw_want = [sync.nw_batch(b, nw_sc) for b in nw_batches]
        sw_want = [(sync.sw_batch(b, sw_sc, 30, max_hits=1), sync.sw_batch(b, sw_sc, 30, max_hits=3)) for b in sw_batches]
    for p, g in enumerate(cfg["pairs"]):     # the synchronous results themselves are the reference's
        assert nw_want[0][p] == (g["score"], g["result_a"].encode(), g["result_b"].encode())
    for p in range(0, 400, 37):
        rc, want = O.oracle_sw(sw_osc, sw_batches[0].seq_a(p), sw_batches[0].seq_b(p), 30, 3)
        assert rc == 0 and sw_want[0][1][p] == want
    errors = []
    with S.Context(0) as ctx:
        def worker(tid):
            try:
                pending = []
                for k in range(12):
                    if (k + tid) % 3 == 2:
                        i, mh = (k + tid) % len(sw_batches), (1, 3)[k % 2]
                        pending.append((ctx.sw_batch_submit(sw_batches[i], sw_sc, 30, max_hits=mh), sw_want[i][k % 2], ("sw", i, mh)))
                    else:
                        i = (k + 2 * tid) % len(nw_batches)
                        pending.append((ctx.nw_batch_submit(nw_batches[i], nw_sc), nw_want[i], ("nw", i)))
                    if len(pending) > 2:
                        job, want, what = pending.pop(0)
                        if job.wait() != want:
                            errors.append((tid, what))
                for job, want, what in pending:
                    if job.wait() != want:
                        errors.append((tid, what))
            except Exception as ex:          # noqa: BLE001 -- reported by the main thread
                errors.append((tid, repr(ex)))
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for t in threads: t.start()
        for t in threads: t.join()
        assert not errors, errors
        # what a job launched is reported through its context after the wait
        j = ctx.nw_batch_submit(nw_batches[1], nw_sc)
        j.wait(raw=True)
        assert any(k.startswith("fill_nw_dirs") for k in ctx.last_call()), ctx.last_call()
        # options are snapshotted at submit: this job takes the three-matrix path although the option is put back at once
        ctx.set_option("nw_dirs", 0)
        j = ctx.nw_batch_submit(nw_batches[1], nw_sc)
        ctx.set_option("nw_dirs", 1)
        assert j.wait() == nw_want[1]
        assert "fill_stream" in ctx.last_call() and not any(k.startswith("fill_nw_dirs") for k in ctx.last_call()), ctx.last_call()


def test_job_errors_arrive_at_the_waiter_and_destroy_drains():
    """A job that fails (an unknown character pair under use_match_mismatch = 0, alignment_scoring.c:178-181) reports its code and text
    where it is waited for, the jobs around it are unaffected; a context destroyed with jobs in flight runs them to the end first."""
    good = S.make_scoring({"preset": "default"})
    bad = S.make_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0], "use_match_mismatch": 0})
    batch = W.dna_nw_150(2000, seed=81)
    with S.Context(0) as ctx:
        want = ctx.nw_batch(batch, good)
        jobs = [ctx.nw_batch_submit(batch, good), ctx.nw_batch_submit(batch, bad), ctx.nw_batch_submit(batch, good)]
        assert jobs[0].wait() == want
        with pytest.raises(S.SeqAlignError) as e:
            jobs[1].wait()
        assert e.value.code == S.E_UNKNOWN_PAIR
        assert jobs[2].wait() == want
    ctx = S.Context(0)
    tail = [ctx.nw_batch_submit(batch, good) for _ in range(5)]
    assert not all(j.done() for j in tail) or True       # (they may or may not have finished yet)
    ctx.close()                                           # drains: every job has run when this returns
    assert all(j.done() for j in tail)
    for j in tail:
        assert j.wait() == want


def test_streaming_is_faster_than_back_to_back_synchronous_calls():
    """BASELINE configs[1] as a stream of 24 batches: three in flight through submit / wait take less wall clock than 24 synchronous
    calls in a row on the same context (not a benchmark -- bench.py's e2e.stream is; this only pins that the lanes do overlap)."""
    import time
    sc = S.make_scoring({"preset": "default"})
    batch = W.dna_nw_150(10000, seed=1)
    with S.Context(0) as ctx:
        bufs = [ctx.nw_buffers(batch) for _ in range(4)]
        for _ in range(3):
            ctx.nw_batch(batch, sc, raw=True)
        warm = [ctx.nw_batch_submit(batch, sc, bufs[k]) for k in range(4)]     # every lane has sized its buffers
        for j in warm: j.wait(raw=True)
        tries = []
        for attempt in range(4):      # (wall clock on a shared box: a burst of foreign CPU load stalls the lanes' host threads -- the best of a few)
            t0 = time.perf_counter()
            for _ in range(24):
                ctx.nw_batch(batch, sc, raw=True)
            t_sync = time.perf_counter() - t0
            t0 = time.perf_counter()
            pending = []
            for k in range(24):
                if len(pending) == 3:
                    pending.pop(0).wait(raw=True)
                pending.append(ctx.nw_batch_submit(batch, sc, bufs[k % 4]))
            for j in pending: j.wait(raw=True)
            t_stream = time.perf_counter() - t0
            tries.append((t_stream, t_sync))
            if t_stream < 0.9 * t_sync:
                break
        assert any(a < 0.9 * b for a, b in tries), tries
