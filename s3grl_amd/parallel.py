"""Multi-GPU: link pairs are independent, so the link list is sharded and nothing else.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Graph and X are
replicated on every rank; each rank runs the engine on a contiguous, cost-balanced range of the
link list.  A collective is only needed when the consumer wants the whole result on every rank:
`sharded_precompute(..., gather=True)` then does one small all-gather of the per-rank row counts
and ONE padded `all_gather_into_tensor` of the rows (equal-sized shards; each shard crosses each
xGMI link once).  The compute callable is injected so the sharding/collective logic can be tested
on CPU with gloo.
"""
from __future__ import annotations

import numpy as np
import torch


def shard_bounds(num_links, world_size, cost=None):
    """Contiguous ranges [b[r], b[r+1]) balanced by `cost` (e.g. deg(src)+deg(dst)); contiguous
    so that the concatenation of the shards is the original order (pos then neg, like the
    reference's call order)."""
    if cost is None:
        return [(num_links * r) // world_size for r in range(world_size + 1)]
    c = np.cumsum(np.asarray(cost, dtype=np.float64))
    total = c[-1] if len(c) else 0.0
    b = [0]
    for r in range(1, world_size):
        b.append(int(np.searchsorted(c, total * r / world_size, side="left")))
    b.append(num_links)
    return [min(max(x, 0), num_links) for x in np.maximum.accumulate(b)]


def link_cost(A, link_index):
    """Cheap proxy of a link's subgraph size: deg(src) + deg(dst) + 1."""
    deg = np.diff(A.indptr)
    li = np.asarray(link_index)
    return deg[li[0]] + deg[li[1]] + 1


def sharded_precompute(compute, link_index, *, rank, world_size, cost=None, group=None, gather=True):
    """`compute(link_index_shard) -> (rows [R_r, ...], row_ptr [L_r + 1])` on this rank's device.

    Returns (rows, row_ptr) of the WHOLE list on every rank when `gather`, else the local shard
    plus its (begin, end) range.
    """
    import torch.distributed as dist

    li = torch.as_tensor(link_index)
    L = li.shape[1]
    b = shard_bounds(L, world_size, cost)
    lo, hi = b[rank], b[rank + 1]
    rows, row_ptr = compute(li[:, lo:hi])
    if not gather or world_size == 1:
        return rows, row_ptr, (lo, hi)
    dev = rows.device
    # 1) sizes
    mine = torch.tensor([rows.shape[0], hi - lo], dtype=torch.int64, device=dev)
    sizes = torch.empty(world_size * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.cpu().view(world_size, 2)
    rmax, lmax = int(sizes[:, 0].max()), int(sizes[:, 1].max())
    # 2) one padded all-gather of the rows, one of the per-link row counts
    tail = rows.shape[1:]
    pad = torch.zeros((rmax,) + tuple(tail), dtype=rows.dtype, device=dev)
    pad[:rows.shape[0]] = rows
    allrows = torch.empty((world_size * rmax,) + tuple(tail), dtype=rows.dtype, device=dev)
    dist.all_gather_into_tensor(allrows, pad, group=group)
    cnt = torch.zeros(lmax, dtype=torch.int64, device=dev)
    cnt[:hi - lo] = (row_ptr[1:] - row_ptr[:-1]).to(dev)
    allcnt = torch.empty(world_size * lmax, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allcnt, cnt, group=group)
    out_rows = torch.cat([allrows[r * rmax: r * rmax + int(sizes[r, 0])] for r in range(world_size)])
    counts = torch.cat([allcnt[r * lmax: r * lmax + int(sizes[r, 1])] for r in range(world_size)])
    out_ptr = torch.zeros(L + 1, dtype=torch.int64, device=dev)
    out_ptr[1:] = torch.cumsum(counts, 0)
    return out_rows, out_ptr, (lo, hi)
