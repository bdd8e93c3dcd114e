"""Multi-GPU sweep plumbing: the frequency table (frequencyTable.cpp:9-37) is range-sharded
over ranks -- rank r of R owns the contiguous centre-frequency indices
[floor(C*r/R), floor(C*(r+1)/R)) -- every buffer is independent, so the data path needs no
collective.  Only the final hit list is gathered, by the C-ABI's scn_gather_hits (RCCL over xGMI:
ncclAllGather of the counts + one group of ncclSend/ncclRecv to the root, scn_gather.hip).
torch.distributed is used for ONE thing on that path: handing rank 0's 128-byte RCCL rendezvous id to
the other ranks.  On CPU (the gloo tests) the same gather runs over torch.distributed with the layout
computed by the C-ABI's scn_gather_layout.  Because shards are contiguous, rank-major concatenation is
already the global (centre index, i) order.
"""
import ctypes as C

import numpy as np

from . import capi


def shard_range(count, rank, world):
    """[lo, hi) of `count` table entries owned by `rank` (same split as scn_frequency_table)."""
    return (count * rank) // world, (count * (rank + 1)) // world


class HitGather:
    """The sweep's communicator: scn_comm (RCCL) on a GPU, torch.distributed (gloo) on CPU."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist

        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = device
        self._comm = None
        if device is not None and torch.device(device).type == "cuda":
            L = capi.lib()
            uid = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8)
            if self.rank == 0:
                buf = (C.c_uint8 * capi.COMM_ID_BYTES)()
                capi.check(L.scn_comm_unique_id(buf), "scn_comm_unique_id")
                uid = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
            if self.world > 1:
                obj = [uid.numpy().tobytes()]
                dist.broadcast_object_list(obj, src=0, group=group)  # 128 bytes through the launcher's store
                uid = torch.frombuffer(bytearray(obj[0]), dtype=torch.uint8)
            raw = (C.c_uint8 * capi.COMM_ID_BYTES).from_buffer_copy(uid.numpy().tobytes())
            h = C.c_void_p()
            capi.check(L.scn_comm_create(raw, self.rank, self.world, torch.device(device).index or 0, C.byref(h)),
                       "scn_comm_create")
            self._comm = h

    def close(self):
        if self._comm:
            capi.lib().scn_comm_destroy(self._comm)
            self._comm = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def gather(self, hits, dst=0, local_status=capi.OK):
        """Every rank passes its ordered scn_hit array; returns (all hits on dst / empty elsewhere, per-rank counts).
        local_status (the torch.distributed form only): a rank that could not prepare its part says so -- it still takes
        part in the exchange, and every rank then raises ScannerError (scn_gather_protocol.h, step 1)."""
        hits = np.ascontiguousarray(hits, dtype=capi.HIT_DTYPE)
        if self._comm:
            return self._gather_rccl(hits, dst)
        return self._gather_torch(hits, dst, local_status)

    def _gather_rccl(self, hits, dst, plan=None, slot=0):
        """ONE collective (counts, then the records straight into the root's device list); the root then reads the list --
        whose length it only knows now -- into an exact-size array with scn_gather_fetch, a local call.  With `plan`,
        this rank's part is the slot's ordered list as the compaction kernel left it in device memory (no host staging)."""
        L = capi.lib()
        total = C.c_uint64()
        per_rank = np.zeros(self.world, np.uint32)
        vp = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        if plan is not None:
            st = L.scn_gather_hits_device(self._comm, plan.handle, slot, dst, None, 0, C.byref(total), per_rank.ctypes.data_as(C.c_void_p))
            capi.check(st, "scn_gather_hits_device")
        else:
            st = L.scn_gather_hits(self._comm, vp(hits), len(hits), dst, None, 0, C.byref(total), per_rank.ctypes.data_as(C.c_void_p))
            capi.check(st, "scn_gather_hits")
        if self.rank != dst:
            return np.zeros(0, capi.HIT_DTYPE), per_rank
        out = np.zeros(total.value, capi.HIT_DTYPE)
        got = C.c_uint64()
        capi.check(L.scn_gather_fetch(self._comm, 0, vp(out), out.size, C.byref(got)), "scn_gather_fetch")
        assert got.value == total.value
        return out, per_rank

    def gather_device(self, plan, slot, dst=0):
        """Every rank passes a plan whose `slot` has been collected; its ordered device list is sent as it is."""
        if not self._comm:
            raise RuntimeError("gather_device needs the RCCL communicator (a cuda device)")
        return self._gather_rccl(None, dst, plan=plan, slot=slot)

    # ---- the steady-state form: a list per sweep, posted asynchronously (scn_gather_post / scn_gather_wait) ----------------
    def post(self, plan_or_hits, slot=0, cap_per_rank=8192, dst=0, local_status=capi.OK):
        """Collective, asynchronous.  On a GPU: `plan_or_hits` is a Plan whose `slot` has been collected; its device list
        travels in one fixed-size message (a header + cap_per_rank records) on the communicator's stream, and the call
        returns a ticket at once (capi.GATHER_TICKETS in flight; the slot must not be submitted again before wait(ticket)).
        A rank whose part cannot go out whole raises AFTER having posted -- its ticket is in .last_ticket and must still
        be waited for.  On CPU (gloo): `plan_or_hits` is this rank's ordered scn_hit array, same message format over
        torch.distributed (the control-flow twin the world-8 dry run exercises)."""
        if self._comm:
            tk = C.c_uint32()
            st = self._L_post(self._comm, plan_or_hits.handle, slot, dst, cap_per_rank, C.byref(tk))
            self.last_ticket = tk.value
            if st:
                capi.check(st, "scn_gather_post")
            return tk.value
        return self._post_torch(np.ascontiguousarray(plan_or_hits, dtype=capi.HIT_DTYPE), cap_per_rank, dst, local_status)

    def wait(self, ticket, copy=True):
        """Blocks until post `ticket` has completed.  Root: (the rank-major list, per-rank true counts) -- the list is a
        view of the communicator's pinned memory with copy=False (valid for GATHER_TICKETS further posts); raises
        ScannerError(E_TRUNCATED) when a rank's list did not fit its message, ScannerError(E_COMM) naming the rank that
        could not prepare its part.  Other ranks: (empty, zeros)."""
        if not self._comm:
            return self._wait_torch(ticket)
        ptr, total = C.c_void_p(), C.c_uint64()
        per_rank = np.zeros(self.world, np.uint32)
        st = self._L_wait(self._comm, ticket, C.byref(ptr), C.byref(total), per_rank.ctypes.data_as(C.c_void_p))
        self.last_per_rank = per_rank
        if not ptr.value or not total.value:
            raw = np.zeros(0, capi.HIT_DTYPE)
        else:
            raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(total.value * capi.HIT_DTYPE.itemsize,)).view(capi.HIT_DTYPE)
        self.last_list = raw.copy() if copy else raw   # (what did arrive stays readable after a truncated / marked post)
        if st:
            capi.check(st, "scn_gather_wait")
        return self.last_list, per_rank

    @property
    def _L_post(self):
        return capi.lib().scn_gather_post

    @property
    def _L_wait(self):
        return capi.lib().scn_gather_wait

    _STREAM_MAGIC = 0x53434E47
    _HEAD = np.dtype([("count", "<u4"), ("status", "<u4"), ("sent", "<u4"), ("magic", "<u4"), ("seq", "<u8")])

    def _post_torch(self, hits, cap, dst, local_status):
        import torch
        import torch.distributed as dist

        st = getattr(self, "_tickets", None)
        if st is None:
            st = self._tickets = {"next": 0, "slots": [None] * capi.GATHER_TICKETS}
        tk = st["next"] % capi.GATHER_TICKETS
        if st["slots"][tk] is not None:
            raise capi.ScannerError(capi.E_STATE, "gather post", f"{capi.GATHER_TICKETS} posts in flight")
        n = len(hits) if local_status == capi.OK else 0
        sent = min(n, cap)
        msg = np.zeros(cap + 1, capi.HIT_DTYPE)
        head = np.zeros(1, self._HEAD)
        head[0] = (n, local_status if local_status != capi.OK else (capi.E_TRUNCATED if sent < n else capi.OK), sent, self._STREAM_MAGIC, st["next"])
        msg[:1] = head.view(capi.HIT_DTYPE)
        msg[1:1 + sent] = hits[:sent]
        buf = torch.from_numpy(msg.view(np.uint8).reshape(-1).copy())
        out = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == dst else None
        work = dist.gather(buf, out, dst=dst, group=self.group, async_op=True) if self.world > 1 else None
        if self.world == 1:
            out = [buf]
        st["slots"][tk] = (work, out, cap, st["next"], dst, buf)
        st["next"] += 1
        self.last_ticket = tk
        if local_status != capi.OK:
            raise capi.ScannerError(local_status, "gather post", "this rank could not prepare its part")
        if sent < n:
            raise capi.ScannerError(capi.E_TRUNCATED, "gather post", f"{n} hits, a message holds {cap}")
        return tk

    def _wait_torch(self, ticket):
        st = getattr(self, "_tickets", None)
        if st is None or ticket >= capi.GATHER_TICKETS or st["slots"][ticket] is None:
            raise capi.ScannerError(capi.E_STATE, "gather wait", f"ticket {ticket} is not in flight")
        work, out, cap, seq, dst, _ = st["slots"][ticket]
        st["slots"][ticket] = None
        if work is not None:
            work.wait()
        per_rank = np.zeros(self.world, np.uint32)
        if self.rank != dst:
            return np.zeros(0, capi.HIT_DTYPE), per_rank
        # the root's reading of the headers: scn_stream_outcome (scn_gather_protocol.h), restated
        parts, bad, trunc = [], None, None
        for r in range(self.world):
            m = out[r].numpy().view(capi.HIT_DTYPE)
            h = m[:1].view(self._HEAD)[0]
            intact = h["magic"] == self._STREAM_MAGIC and h["seq"] == seq and h["sent"] <= cap and h["sent"] <= h["count"]
            if not intact:
                bad = bad or (r, capi.E_COMM, "does not belong to this post")
                continue
            per_rank[r] = h["count"]
            parts.append(m[1:1 + int(h["sent"])])
            if h["status"] != capi.OK and not (h["status"] == capi.E_TRUNCATED and h["sent"] < h["count"]):
                bad = bad or (r, int(h["status"]), "could not prepare its part of the gather")
            elif h["sent"] < h["count"]:
                trunc = trunc if trunc is not None else r
        self.last_per_rank = per_rank
        self.last_list = np.concatenate(parts) if parts else np.zeros(0, capi.HIT_DTYPE)
        if bad:
            raise capi.ScannerError(capi.E_COMM, "gather wait", f"rank {bad[0]} {bad[2]} (status {bad[1]})")
        if trunc is not None:
            raise capi.ScannerError(capi.E_TRUNCATED, "gather wait", f"rank {trunc} holds {per_rank[trunc]} hits, its message {cap}")
        return self.last_list, per_rank

    def _gather_torch(self, hits, dst, local_status=capi.OK):
        """The steps of scn_gather_protocol.h over torch.distributed: all-gather {count, status}; if anybody announced a
        failure every rank raises and nothing is transferred; else the records go to dst."""
        import torch
        import torch.distributed as dist

        if self.world == 1:
            if local_status != capi.OK:
                raise capi.ScannerError(local_status, "gather", "this rank could not prepare its part")
            return hits, np.array([len(hits)], np.uint32)
        dev = self.device or "cpu"
        cnt = torch.tensor([len(hits) if local_status == capi.OK else 0, int(local_status)], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(counts, cnt, group=self.group)
        bad = [r for r, c in enumerate(counts) if int(c[1].item()) != capi.OK]
        if bad:
            if self.rank == bad[0]:
                raise capi.ScannerError(local_status, "gather", "this rank could not prepare its part")
            raise capi.ScannerError(capi.E_COMM, "gather", f"rank {bad[0]} could not prepare its part of the gather "
                                                           f"(status {int(counts[bad[0]][1].item())}): nothing was exchanged")
        per_rank = np.array([int(c[0].item()) for c in counts], np.uint32)
        off = capi.gather_layout(per_rank)
        width = max(int(per_rank.max()), 1) * capi.HIT_DTYPE.itemsize
        buf = torch.zeros(width, dtype=torch.uint8, device=dev)
        if len(hits):
            buf[: hits.nbytes] = torch.from_numpy(hits.view(np.uint8).reshape(-1).copy()).to(dev)
        out = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == dst else None
        dist.gather(buf, out, dst=dst, group=self.group)
        if self.rank != dst:
            return np.zeros(0, capi.HIT_DTYPE), per_rank
        all_hits = np.zeros(int(off[-1]), capi.HIT_DTYPE)
        for r in range(self.world):
            all_hits[int(off[r]):int(off[r + 1])] = out[r][: int(per_rank[r]) * capi.HIT_DTYPE.itemsize].cpu().numpy().view(capi.HIT_DTYPE)
        return all_hits, per_rank


def gather_hits(hits, device, group=None, dst=0):
    """One-shot form: gather every rank's (already ordered) scn_hit array to `dst`."""
    with HitGather(device, group) as g:
        return g.gather(hits, dst)[0]
