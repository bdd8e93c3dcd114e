"""World-size-2 CPU test (gloo) of the multi-GPU plumbing: contiguous table shards and the
hit-list gather.  The data path itself needs no collective (every buffer is independent);
per-rank hit lists here come from the CPU oracle standing in for a rank's GPU results."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_centres, out_dir, silent_rank=-1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from scanner_amd import capi, sweep, synth

    n, fs = 1024, 8000000
    first, fc = capi.frequency_table(fs, 0.0, n_centres * 0.75 * fs, shard=rank, n_shards=world)
    lo, hi = sweep.shard_range(n_centres, rank, world)
    assert (first, first + len(fc)) == (lo, hi)
    # every rank generates the WHOLE sweep deterministically and keeps its shard
    x = synth.cfloat_batch(n, n_centres, seed=77, sigma=0.1)[lo:hi]
    seq = np.arange(lo, hi, dtype=np.uint64)
    _, hits, _ = O.Oracle(n, fs, 9.0).run(x, fc, seq)
    if rank == silent_rank:  # a shard that saw nothing: its (empty) list still takes part in the gather
        hits = hits[:0]
    allh = sweep.gather_hits(hits.astype(capi.HIT_DTYPE), torch.device("cpu"))
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), allh)
    else:
        assert len(allh) == 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_centres", [7, 64])
def test_sharded_sweep_gather_matches_single_process(tmp_path, oracle_mod, built_lib, n_centres):
    from scanner_amd import capi, synth

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_centres, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    n, fs = 1024, 8000000
    _, fc = capi.frequency_table(fs, 0.0, n_centres * 0.75 * fs)
    x = synth.cfloat_batch(n, n_centres, seed=77, sigma=0.1)
    _, ref, _ = oracle_mod.Oracle(n, fs, 9.0).run(x, fc, np.arange(n_centres, dtype=np.uint64))
    assert len(ref) > 0
    for f in ("seq_id", "i", "power_db", "freq_hz"):
        assert np.array_equal(got[f], ref[f]), f          # rank-major concatenation == global order


def test_gather_world_three_with_an_empty_shard(tmp_path, oracle_mod, built_lib):
    """World size 3, the middle rank's shard holds no detection (a rank whose part of the band is quiet): the
    gathered list is the concatenation of ranks 0 and 2, in order, and nobody waits for records that never come."""
    from scanner_amd import capi, sweep, synth

    world, n_centres, n, fs = 3, 10, 1024, 8000000
    mp.spawn(_worker, args=(world, _free_port(), n_centres, str(tmp_path), 1), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    _, fc = capi.frequency_table(fs, 0.0, n_centres * 0.75 * fs)
    x = synth.cfloat_batch(n, n_centres, seed=77, sigma=0.1)
    _, ref, _ = oracle_mod.Oracle(n, fs, 9.0).run(x, fc, np.arange(n_centres, dtype=np.uint64))
    lo, hi = sweep.shard_range(n_centres, 1, world)
    keep = (ref["seq_id"] < lo) | (ref["seq_id"] >= hi)
    assert keep.any() and not keep.all()
    assert got.tobytes() == ref[keep].astype(capi.HIT_DTYPE).tobytes()


def test_shard_ranges_partition():
    from scanner_amd import sweep

    for count in (0, 1, 7, 16384):
        for world in (1, 2, 3, 8):
            r = [sweep.shard_range(count, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == count
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


@pytest.mark.parametrize("san", ["", "thread"])
def test_gather_protocol_every_rank_returns(tmp_path, san):
    """The gather's control flow -- scanner_amd/csrc/scn_gather_protocol.h, the very code scn_gather.hip runs over RCCL -- with
    threads for ranks (worlds of 3 and 8) and a failure injected on each rank in turn: its part cannot be prepared, the staging
    of its announce words fails, it cannot read an announce back, the root cannot make room.  Every rank must return, with an
    error, within the transport's deadline, and nothing may be transferred (tests/cpp/test_gather_protocol.cpp)."""
    import subprocess

    exe = tmp_path / "test_gather_protocol"
    cmd = ["g++", "-std=gnu++11", "-O1", "-g", "-pthread", "-I", os.path.join(ROOT, "scanner_amd", "csrc")] + (
        [f"-fsanitize={san}"] if san else []) + [os.path.join(ROOT, "tests", "cpp", "test_gather_protocol.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and san and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available")
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if out.returncode != 0 and "unexpected memory mapping" in out.stderr:
        pytest.skip("TSan cannot run under this kernel's address-space layout")
    assert out.returncode == 0 and "gather protocol tests ok" in out.stdout, out.stderr[-2000:]


def _failing_worker(rank, world, port, bad_rank, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scanner_amd import capi, sweep

    hits = np.zeros(5 + rank, capi.HIT_DTYPE)
    hits["seq_id"] = 100 * rank + np.arange(len(hits))
    import time

    t0 = time.time()
    try:
        with sweep.HitGather(torch.device("cpu")) as g:
            g.gather(hits, local_status=capi.E_NOMEM if rank == bad_rank else capi.OK)
        result = "ok"
    except capi.ScannerError as e:
        result = f"error {e.status}"
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(f"{result} {time.time() - t0:.3f}")
    dist.barrier()
    dist.destroy_process_group()


def test_gather_with_a_failing_rank_over_gloo(tmp_path, built_lib):
    """The CPU twin of the same protocol (sweep.HitGather over torch.distributed): rank 1 of 3 cannot stage its part; every
    rank -- the root included -- returns an error promptly instead of waiting for records that never come."""
    world = 3
    mp.spawn(_failing_worker, args=(world, _free_port(), 1, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        word, status, seconds = open(tmp_path / f"rank{r}.txt").read().split()
        assert word == "error" and float(seconds) < 20.0, (r, word, status, seconds)
        assert int(status) == (3 if r == 1 else 7)   # E_NOMEM on the failing rank, E_COMM on its peers


def _stream_worker(rank, world, port, out_dir):
    """Six sweeps, a list per sweep posted asynchronously (HitGather.post / wait over gloo: the message format and the root's
    reading of the headers are those of scn_gather_post / scn_gather_wait), up to GATHER_TICKETS in flight."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scanner_amd import capi, sweep

    def part(sweep_no, r):   # rank r's ordered records of sweep `sweep_no`: r + 2 * sweep_no + 1 of them (rank 1: none on odd sweeps)
        k = 0 if (r == 1 and sweep_no % 2) else r + 2 * sweep_no + 1
        h = np.zeros(k, capi.HIT_DTYPE)
        h["seq_id"] = 1000 * r + np.arange(k)
        h["i"] = 600 + sweep_no
        h["power_db"] = 10.0 + r
        h["freq_hz"] = 7 * r + sweep_no
        return h

    cap = 16
    g = sweep.HitGather(torch.device("cpu"))
    tickets, results = [], []
    for sw in range(6):
        if len(tickets) == capi.GATHER_TICKETS:
            results.append(g.wait(tickets.pop(0))[0])
        tickets.append(g.post(part(sw, rank), cap_per_rank=cap))
    if rank == 0:                                   # a fifth post while four are in flight is refused before anything is sent
        try:
            g._post_torch(part(0, 0), cap, 0, capi.OK)
            raise AssertionError("the ring should be full")
        except capi.ScannerError as e:
            assert e.status == capi.E_STATE
    for t in tickets:
        results.append(g.wait(t)[0])
    if rank == 0:
        for sw, got in enumerate(results):
            want = np.concatenate([part(sw, r) for r in range(world)])
            assert got.tobytes() == want.tobytes(), sw
    else:
        assert all(len(r) == 0 for r in results)
    # sweep 7: rank 2's list does not fit its message -- it learns so from its own post, the root from its wait (true count reported)
    big = part(9, rank)
    try:
        tk = g.post(big, cap_per_rank=18)
        assert len(big) <= 18
    except capi.ScannerError as e:
        assert e.status == capi.E_TRUNCATED and len(big) > 18
        tk = g.last_ticket
    try:
        g.wait(tk)
        assert rank != 0
    except capi.ScannerError as e:
        assert rank == 0 and e.status == capi.E_TRUNCATED and g.last_per_rank.tolist() == [len(part(9, r)) for r in range(world)]
        want = np.concatenate([part(9, r)[:18] for r in range(world)])
        assert g.last_list.tobytes() == want.tobytes()
    # sweep 8: the last rank could not prepare its part -- it still posts (marked), the root names it, nobody hangs
    try:
        tk = g.post(part(1, rank), cap_per_rank=cap, local_status=capi.E_NOMEM if rank == world - 1 else capi.OK)
        assert rank != world - 1
    except capi.ScannerError as e:
        assert rank == world - 1 and e.status == capi.E_NOMEM
        tk = g.last_ticket
    try:
        g.wait(tk)
        assert rank != 0
    except capi.ScannerError as e:
        assert rank == 0 and e.status == capi.E_COMM and f"rank {world - 1}" in str(e)
    dist.barrier()
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 3])
def test_gather_post_wait_over_gloo(tmp_path, built_lib, world):
    mp.spawn(_stream_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").read_text() == "ok"
