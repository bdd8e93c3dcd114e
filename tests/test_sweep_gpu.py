"""BASELINE config 4 on the GPU: the full frequency table FrequencyTable(8e6, 0, 16384*6e6) (frequencyTable.cpp:9-37)
-> 16384 centres x 4096-pt, range-sharded 8 ways the way the 8 ranks of a node take it (scan.cpp:211-239 per shard).
One GPU here, so the 8 shards run one after the other: their concatenation must equal the unsharded sweep and the
closed-form expectation of the planted emitters (tests/test_bench_launcher.py pins that form to the oracle); the
gather itself goes through the C-ABI's RCCL path with a one-rank communicator, and bench.py's C4 mode is run end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from scanner_amd import Plan, capi, sweep, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS, N, CENTRES, THR = 8000000, 4096, 16384, 10.0


@pytest.fixture(scope="module")
def torch_cuda(built_lib):
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU; refusing to skip silently"
    return torch


def _sweep(plan, x, fc, first, max_batch):
    """Run table entries [first, first+len(fc)) through the plan in launches of max_batch, double-buffered."""
    parts, pending = [], []
    for j, lo in enumerate(range(0, len(fc), max_batch)):
        hi = min(lo + max_batch, len(fc))
        if len(pending) == 2:
            parts.append(plan.collect(pending.pop(0), want_power=False)[1])
        plan.submit_device(j & 1, x[lo:hi], hi - lo, fc[lo:hi], np.arange(first + lo, first + hi, dtype=np.uint64),
                           sync_producer=False)
        pending.append(j & 1)
    for s in pending:
        parts.append(plan.collect(s, want_power=False)[1])
    return np.concatenate(parts)


def test_c4_eight_shards_equal_unsharded_and_expectation(torch_cuda):
    torch = torch_cuda
    dev = torch.device("cuda", 0)
    _, fc_all = capi.frequency_table(FS, 0.0, CENTRES * 0.75 * FS)
    assert len(fc_all) == CENTRES and fc_all[0] == 3e6 and fc_all[-1] == 3e6 + (CENTRES - 1) * 6e6
    centres, i0 = synth.c4_emitters(CENTRES, N)
    x = synth.c4_shard_torch(N, 0, CENTRES, centres, i0, seed=4, device=dev)   # the whole sweep, 512 MiB, in HBM
    torch.cuda.synchronize()
    with Plan(N, FS, THR, max_batch=8192, flags=capi.OUT_HITS) as plan:
        whole = _sweep(plan, x, fc_all, 0, 8192)
        window = plan.window()
    shards = []
    with Plan(N, FS, THR, max_batch=2048, flags=capi.OUT_HITS) as plan:         # what one of 8 ranks runs
        for r in range(8):
            first, fc = capi.frequency_table(FS, 0.0, CENTRES * 0.75 * FS, shard=r, n_shards=8)
            assert (first, len(fc)) == (2048 * r, 2048) == (sweep.shard_range(CENTRES, r, 8)[0], 2048)
            shards.append(_sweep(plan, x[first:first + 2048], fc, first, 2048))
    cat = np.concatenate(shards)                                               # rank-major concatenation
    assert cat.tobytes() == whole.tobytes()                                    # identical, bit for bit
    off = capi.gather_layout([len(s) for s in shards])
    assert off[-1] == len(cat) and all(np.array_equal(cat[int(off[r]):int(off[r + 1])], shards[r]) for r in range(8))
    want = synth.c4_expected_hits(window, fc_all, centres, i0, N, FS, THR)
    assert len(want) == 7 * 4096 == len(whole)
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(whole[f], want[f]), f
    assert np.abs(whole["power_db"] - want["power_db"]).max() < 0.15
    sid, i = whole["seq_id"].astype(np.int64), whole["i"].astype(np.int64)
    assert np.all((np.diff(sid) > 0) | ((np.diff(sid) == 0) & (np.diff(i) > 0)))   # global (centre, i) order


def test_gather_hits_through_rccl_one_rank(torch_cuda):
    """scn_comm_* / scn_gather_hits with a one-rank communicator: RCCL loads (dlopen), the count all-gather and the
    root's copy run on the GPU, the list comes back unchanged."""
    dev = torch_cuda.device("cuda", 0)
    hits = np.zeros(1000, capi.HIT_DTYPE)
    hits["seq_id"] = np.repeat(np.arange(100), 10)
    hits["i"] = np.tile(np.arange(600, 610), 100)
    hits["power_db"] = np.linspace(10, 30, 1000)
    hits["freq_hz"] = hits["seq_id"] * 6000000 + hits["i"] * 1953
    with sweep.HitGather(dev) as g:
        assert g._comm, "the GPU path must use the C-ABI communicator"
        got, per_rank = g.gather(hits)
        assert per_rank.tolist() == [1000] and got.tobytes() == hits.tobytes()
        got, per_rank = g.gather(hits[:0])                                     # an empty list is a valid sweep result
        assert per_rank.tolist() == [0] and len(got) == 0


def test_gather_from_the_slots_device_list_and_fetch_windows(torch_cuda, oracle_mod):
    """scn_gather_hits_device sends a collected slot's ordered list from where the compaction kernel left it (no host
    staging); `all == NULL` on the root is the free size query -- the records are exchanged once and read afterwards,
    whole or window by window, with scn_gather_fetch."""
    import ctypes as C

    from scanner_amd import Plan, synth

    L = capi.lib()
    dev = torch_cuda.device("cuda", 0)
    n, nb, fs, thr = 4096, 64, 8000000, 9.5
    x = synth.cfloat_batch(n, nb, seed=41)
    fc = 3e6 + 6e6 * np.arange(nb)
    with Plan(n, fs, thr, max_batch=nb, max_hits=1 << 16) as plan, sweep.HitGather(dev) as g:
        plan.submit_device(0, torch_cuda.from_numpy(x.view(np.uint8).reshape(-1)).cuda(), nb, fc)
        _, h, _ = plan.collect(0, want_power=False, hit_cap=1 << 16)
        assert len(h) > 100
        got, per_rank = g.gather_device(plan, 0)                 # counts-only collective call + local fetch inside
        assert per_rank.tolist() == [len(h)] and got.tobytes() == h.tobytes()
        # windows of the gathered list, and past its end
        win = np.zeros(50, capi.HIT_DTYPE)
        k = C.c_uint64()
        capi.check(L.scn_gather_fetch(g._comm, 37, win.ctypes.data_as(C.c_void_p), 50, C.byref(k)), "scn_gather_fetch")
        assert k.value == 50 and win.tobytes() == h[37:87].tobytes()
        capi.check(L.scn_gather_fetch(g._comm, len(h) - 5, win.ctypes.data_as(C.c_void_p), 50, C.byref(k)), "scn_gather_fetch")
        assert k.value == 5 and win[:5].tobytes() == h[-5:].tobytes()
        capi.check(L.scn_gather_fetch(g._comm, len(h), win.ctypes.data_as(C.c_void_p), 50, C.byref(k)), "scn_gather_fetch")
        assert k.value == 0
        # a caller buffer that is too small: truncated, nothing lost
        small = np.zeros(10, capi.HIT_DTYPE)
        tot = C.c_uint64()
        st = L.scn_gather_hits_device(g._comm, plan.handle, 0, 0, small.ctypes.data_as(C.c_void_p), 10, C.byref(tot), None)
        assert st == capi.E_TRUNCATED and tot.value == len(h) and small.tobytes() == h[:10].tobytes()
        # a slot that was never collected: the rank ANNOUNCES the failure inside the exchange (every rank of a larger
        # world would return an error instead of waiting for its records) and reports its own status
        st = L.scn_gather_hits_device(g._comm, plan.handle, 1, 0, None, 0, C.byref(tot), None)
        assert st == capi.E_STATE and tot.value == 0
        st = L.scn_gather_hits(g._comm, None, 5, 0, None, 0, C.byref(tot), None)   # a count without a list
        assert st == capi.E_INVALID
        st = L.scn_gather_fetch(g._comm, 0, win.ctypes.data_as(C.c_void_p), 50, C.byref(k))
        assert st == capi.E_STATE                                # the failed gather left no list behind
        got, _ = g.gather_device(plan, 0)                        # and the communicator is still usable
        assert got.tobytes() == h.tobytes()


def test_bench_c4_line(torch_cuda):
    """bench.py --config c4 end to end (reduced table so the suite stays short; the full table is the test above)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4", "--centres", "4096", "--steps", "4",
                          "--warmup", "1", "--settle", "0.05", "--no-cpu-baseline", "--no-overlap-leg"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip())
    assert d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 4096 and d["config"]["n"] == 4096
    c = d["c4_check"]
    assert c["match"] is True and c["expected_hits"] == c["gathered_hits"] == 7 * 1024
    assert d["roofline"]["kernel"].startswith("scn_fft_kernel<16, SCN_K_FLOAT_COMPLEX")
    assert d["with_hit_records"]["hits_per_step"] == 7 * 1024


def test_gather_post_wait_one_rank(torch_cuda):
    """The steady-state form (scn_gather_post / scn_gather_wait) with a one-rank communicator: a collected slot's device list
    goes through pack -> (no peers) -> compaction into the communicator's pinned list on its own stream; four posts in flight,
    tickets come back in order, the lists are the slots' own; a list longer than the message is cut and reported with the true
    count; a slot that was never collected travels as a marked message the root names."""
    import ctypes as C

    from scanner_amd import Plan, synth

    L = capi.lib()
    dev = torch_cuda.device("cuda", 0)
    n, nb, fs, thr = 4096, 64, 8000000, 9.5
    with Plan(n, fs, thr, max_batch=nb, max_hits=1 << 16) as plan, sweep.HitGather(dev) as g:
        lists = []
        for s in range(4):
            x = synth.cfloat_batch(n, nb, seed=50 + s)
            plan.submit_device(s, torch_cuda.from_numpy(x.view(np.uint8).reshape(-1)).cuda(), nb, 3e6 + 6e6 * np.arange(nb))
        for s in range(4):
            lists.append(plan.collect(s, want_power=False, hit_cap=1 << 16)[1])
            assert len(lists[-1]) > 100
        cap = max(len(h) for h in lists) + 7
        tickets = [g.post(plan, s, cap) for s in range(4)]                 # four in flight
        assert tickets == [0, 1, 2, 3]
        tk = C.c_uint32()
        assert L.scn_gather_post(g._comm, plan.handle, 0, 0, cap, C.byref(tk)) == capi.E_STATE   # the ring is full
        for s in (0, 1, 2, 3):
            got, per_rank = g.wait(tickets[s])
            assert per_rank.tolist() == [len(lists[s])] and got.tobytes() == lists[s].tobytes()
        with pytest.raises(capi.ScannerError):
            g.wait(0)                                                      # not in flight any more
        # the view form, and a ring that goes round
        for rep in range(6):
            s = rep % 4
            got, _ = g.wait(g.post(plan, s, cap), copy=False)
            assert got.tobytes() == lists[s].tobytes()
        # a list longer than the message: the first cap records, the true count, SCN_E_TRUNCATED on both calls
        small = len(lists[1]) - 10
        with pytest.raises(capi.ScannerError) as e:
            g.post(plan, 1, small)
        assert e.value.status == capi.E_TRUNCATED
        with pytest.raises(capi.ScannerError) as e:
            g.wait(g.last_ticket)
        assert e.value.status == capi.E_TRUNCATED and g.last_per_rank.tolist() == [len(lists[1])]
        assert g.last_list.tobytes() == lists[1][:small].tobytes()
        # a part that cannot be prepared (the slot has a submit pending, never collected): posted all the same, marked
        x = synth.cfloat_batch(n, nb, seed=99)
        plan.submit_device(2, torch_cuda.from_numpy(x.view(np.uint8).reshape(-1)).cuda(), nb, 3e6 + 6e6 * np.arange(nb))
        with pytest.raises(capi.ScannerError) as e:
            g.post(plan, 2, cap)
        assert e.value.status == capi.E_STATE
        with pytest.raises(capi.ScannerError) as e:
            g.wait(g.last_ticket)
        assert e.value.status == capi.E_COMM and "rank 0" in str(e.value) and len(g.last_list) == 0
        plan.collect(2, want_power=False)
        got, _ = g.wait(g.post(plan, 2, cap))                              # and the communicator is still usable
        assert len(got) > 100
        # the three-step form and the steady-state form share the communicator
        got2, _ = g.gather_device(plan, 2)
        assert got2.tobytes() == got.tobytes()
