"""Multi-GPU sweep plumbing: the frequency table (frequencyTable.cpp:9-37) is range-sharded
over ranks -- rank r of R owns the contiguous centre-frequency indices
[floor(C*r/R), floor(C*(r+1)/R)) -- every buffer is independent, so the data path needs no
collective.  Only the final hit list is gathered (RCCL over xGMI when the process group's
backend is "nccl"; gloo on CPU for tests): one all_gather of counts, one padded gather of
24-byte scn_hit records to the destination rank.  Because shards are contiguous,
rank-major concatenation is already the global (centre index, i) order.
"""
import numpy as np

from . import capi


def shard_range(count, rank, world):
    """[lo, hi) of `count` table entries owned by `rank` (same split as scn_frequency_table)."""
    return (count * rank) // world, (count * (rank + 1)) // world


def gather_hits(hits, device, group=None, dst=0):
    """Gather every rank's (already ordered) scn_hit array to `dst`.
    Returns the concatenated array on dst, an empty array elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    hits = np.ascontiguousarray(hits, dtype=capi.HIT_DTYPE)
    cnt = torch.tensor([len(hits)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1) * capi.HIT_DTYPE.itemsize
    buf = torch.zeros(width, dtype=torch.uint8, device=device)
    if len(hits):
        buf[: hits.nbytes] = torch.from_numpy(hits.view(np.uint8).reshape(-1).copy()).to(device)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return np.zeros(0, capi.HIT_DTYPE)
    parts = [out[r][: counts[r] * capi.HIT_DTYPE.itemsize].cpu().numpy().view(capi.HIT_DTYPE) for r in range(world)]
    return np.concatenate(parts) if parts else np.zeros(0, capi.HIT_DTYPE)
