"""Multi-GPU: link pairs are independent, so the link list is sharded and nothing else.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Graph and X are
replicated on every rank; each rank runs the engine on its share of the link list — cost-balanced,
pair-aware (`ShardPlan`: both directions of a pair on one rank) or, on request, contiguous ranges
(reference tuned_SIGN.py:147-187: the loop bodies share no state).  The one exchange
step is the reassembly of the result on every rank: padded `all_gather_into_tensor` calls (equal
sized contributions: a single RCCL all-gather, each rank's slice crossing each xGMI link once)
followed by a compaction of the padded slices into the reference's order.

Two flavours:

* fixed rows per link (PoS, SoP, hybrid: 2 rows) — every size is known from the shard bounds, so
  nothing is exchanged but the rows themselves, and a rank's range can be cut into `chunks`
  pieces whose all-gathers run on RCCL's stream while the next piece is being computed;
* ragged (PoS Plus) — one small all-gather of the per-rank row counts first.

The compute callable is injected, so the sharding / collective logic is tested on CPU with gloo
and the same code runs the engine on the GPU (`engine_compute`).
"""
from __future__ import annotations

import time

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
    return [int(min(max(x, 0), num_links)) for x in np.maximum.accumulate(b)]


def shard_assignment(link_index, world_size, cost=None, pair_aware=True):
    """Which rank computes which link: (order int64 [L], bounds [world_size + 1]) — rank r owns the
    links order[bounds[r]:bounds[r+1]].

    pair_aware: both directions of a pair, (s,d) and (d,s), go to the SAME rank, so that the engine
    still folds the reversed duplicate into its primary there (the reference's train positives hold
    both directions of every train edge: 23 % of PubMed's list is served for free on one GPU, and
    contiguous ranges of the permuted list split most of those pairs across ranks).  Pairs are
    taken in order of first appearance and cut into `world_size` runs of equal total cost — the pairs
    present in both directions and the others separately, a rank taking one run of each.  Without pair_aware: contiguous
    ranges (`shard_bounds`), order = identity."""
    li = np.asarray(torch.as_tensor(link_index).cpu())
    L = int(li.shape[1])
    if not pair_aware or L == 0:
        return np.arange(L, dtype=np.int64), shard_bounds(L, world_size, cost)
    lo, hi = np.minimum(li[0], li[1]).astype(np.int64), np.maximum(li[0], li[1]).astype(np.int64)
    key = lo * (int(hi.max()) + 1) + hi
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    c = np.ones(L) if cost is None else np.asarray(cost, dtype=np.float64)
    gcost = np.bincount(inv, weights=c, minlength=len(first))
    # Pairs present in BOTH directions and the others are cut apart, each kind into `world_size` runs of
    # equal cost, and rank r takes run r of either kind: every rank then holds the same share of reversed
    # duplicates — of what the mirror-free exchange (`mirror_rows`) does not put on the wire.  A padded
    # all-gather moves world_size x the LARGEST contribution: with the pairs simply in list order the ranks
    # that hold the negatives (no reversed duplicates) set that size for everybody.
    fwd = np.bincount(inv, weights=(li[0] <= li[1]).astype(np.float64), minlength=len(first))
    both_dirs = (fwd > 0) & (fwd < np.bincount(inv, minlength=len(first)))
    rank_of_group = np.empty(len(first), dtype=np.int64)
    for kind in (True, False):
        members = np.flatnonzero(both_dirs == kind)
        gorder = members[np.argsort(first[members], kind="stable")]   # pairs by first appearance
        gb = shard_bounds(len(gorder), world_size, gcost[gorder])      # runs of pairs of equal total cost
        for r in range(world_size):
            rank_of_group[gorder[gb[r]:gb[r + 1]]] = r
    rank_of_link = rank_of_group[inv]
    # inside a rank: pair by pair (first appearance), the two directions next to each other — a rank's
    # list is cut into pieces for the pipelined all-gather, and a cut must not separate partners
    order = np.lexsort((np.arange(L), first[inv], rank_of_link)).astype(np.int64)
    counts = np.bincount(rank_of_link, minlength=world_size)
    bounds = [0] + [int(x) for x in np.cumsum(counts)]
    return order, bounds


class ShardPlan:
    """An assignment made once (set-up, like the uploads) and reused by every step: the list grouped
    by rank on the device, where every column belongs in the caller's list, and the bounds."""

    def __init__(self, link_index, world_size, cost=None, pair_aware=True, device=None, replicate=None):
        """`replicate` (bool mask over the list, pair-aware plans only): links that EVERY rank computes itself
        instead of receiving them — the exchange, not the compute, bounds a sharded step (1.5 GB over xGMI
        against 12 ms of engine for the whole headline list), and the cheapest links of a list cost a few
        per cent of its compute but their full share of its bytes (`replicate_cheapest`).  They sit behind
        the ranks' own links in `links` / `order`; `bounds` covers the sharded part only."""
        li = torch.as_tensor(link_index)
        self.num_links = int(li.shape[1])
        rep = None if replicate is None else np.asarray(replicate, dtype=bool)
        if rep is not None and (not pair_aware or not rep.any()):
            rep = None
        if rep is None:
            order, self.bounds = shard_assignment(li, world_size, cost, pair_aware)
            self.rep_start = self.num_links
        else:
            assert rep.shape == (self.num_links,)
            li_np = np.asarray(li.cpu())
            keep, dup = np.flatnonzero(~rep), np.flatnonzero(rep)
            c = None if cost is None else np.asarray(cost, dtype=np.float64)
            o_s, self.bounds = shard_assignment(li_np[:, keep], world_size, None if c is None else c[keep], True)
            o_r, _ = shard_assignment(li_np[:, dup], 1, None, True)      # partners next to each other
            order = np.concatenate([keep[o_s], dup[o_r]]).astype(np.int64)
            self.rep_start = len(keep)
        self.identity = not pair_aware
        self.order = torch.from_numpy(order)
        self.cost = None if cost is None else np.asarray(cost, dtype=np.float64)[order]
        dev = li.device if device is None else torch.device(device)
        self.links = li.to(dev)[:, self.order.to(dev)].contiguous()
        self.order_dev = self.order.to(dev)
        # a link that is the reverse of the one right before it in a rank's list (pair-aware shards put the
        # two directions of a pair next to each other, the first of the list first): its two rows are the
        # other's in swapped order, so they need not travel (see `_fixed(..., mirror_rows=True)`)
        sl = np.asarray(self.links.cpu())
        rev = np.zeros(self.num_links, dtype=bool)
        if pair_aware and self.num_links > 1:
            rev[1:] = (sl[0, 1:] == sl[1, :-1]) & (sl[1, 1:] == sl[0, :-1]) & (sl[0, 1:] != sl[1, 1:])
            for r in range(world_size + 1):                 # never across a rank boundary (or into the replicated part)
                if 0 < self.bounds[r] < self.num_links:
                    rev[self.bounds[r]] = False
            # (a, b), (b, a), (a, b): the third is the reverse of a mirror, not of a primary — it travels
            run = rev.copy()
            run[1:] &= rev[:-1]
            while run.any():
                rev[np.flatnonzero(run & ~np.roll(run, 1))] = False
                run = rev.copy()
                run[1:] &= rev[:-1]
        self.reverse_of_previous = rev
        self._transport = {}

    def transport(self, chunks, device):
        """Per piece c (of `chunks` per rank) and rank r, for the mirror-free exchange of `_fixed`:
        `pb[r]` piece bounds, `prim[r][c]` positions inside the piece of the links that travel,
        `mir[r][c]` = (positions inside the piece of the links that do not, index of their primary among
        the piece's travelling links) — device tensors, made once."""
        key = (int(chunks), str(device))
        t = self._transport.get(key)
        if t is not None:
            return t
        world = len(self.bounds) - 1
        pb = [chunk_bounds(self.bounds[r], self.bounds[r + 1], chunks, self.cost) for r in range(world)]
        prim, mir, nprim = [], [], []
        for r in range(world):
            pr, mr, nr = [], [], []
            for c in range(chunks):
                p0, p1 = pb[r][c], pb[r][c + 1]
                m = self.reverse_of_previous[p0:p1].copy()
                if len(m):
                    m[0] = False                           # its primary sits in the piece before: it travels
                keep = np.flatnonzero(~m)
                slot_of = np.cumsum(~m) - 1                # travelling links before-or-at every position
                mpos = np.flatnonzero(m)
                pr.append(torch.from_numpy(keep).to(device))
                mr.append((torch.from_numpy(mpos).to(device), torch.from_numpy(slot_of[mpos]).to(device)))
                nr.append(len(keep))
            prim.append(pr)
            mir.append(mr)
            nprim.append(nr)
        t = (pb, prim, mir, nprim)
        self._transport[key] = t
        return t


def replicate_cheapest(link_index, cost, fraction):
    """Mask of the links every rank should compute itself: the cheapest pairs (both directions of a pair
    together, priced by their summed cost per link) up to `fraction` of the list."""
    li = np.asarray(torch.as_tensor(link_index).cpu())
    L = int(li.shape[1])
    mask = np.zeros(L, dtype=bool)
    want = int(L * float(fraction))
    if want <= 0 or L == 0:
        return mask
    lo, hi = np.minimum(li[0], li[1]).astype(np.int64), np.maximum(li[0], li[1]).astype(np.int64)
    _, inv, cnt = np.unique(lo * (int(hi.max()) + 1) + hi, return_inverse=True, return_counts=True)
    per_link = np.bincount(inv, weights=np.asarray(cost, dtype=np.float64), minlength=len(cnt)) / cnt
    groups = np.argsort(per_link, kind="stable")
    take = groups[:int(np.searchsorted(np.cumsum(cnt[groups]), want, side="right"))]
    pick = np.zeros(len(cnt), dtype=bool)
    pick[take] = True
    return pick[inv]


def link_cost(A, link_index):
    """Cheap proxy of a link's subgraph size: deg(src) + deg(dst) + 1."""
    deg = np.diff(A.indptr)
    li = np.asarray(link_index)
    return deg[li[0]] + deg[li[1]] + 1


def khop_cost(A, link_index, num_hops):
    """Better proxy for num_hops >= 2: the number of walks of length <= num_hops leaving src and
    dst (powers of the degree vector, O(num_hops * nnz) once) — an upper bound of the subgraph's
    node count that tracks it far better than the endpoint degrees on graphs with hubs."""
    B = (A != 0).astype(np.float64)
    w = np.ones(A.shape[0])
    tot = np.ones(A.shape[0])
    for _ in range(max(int(num_hops), 1)):
        w = B @ w
        tot += w
    li = np.asarray(link_index)
    return tot[li[0]] + tot[li[1]]


def measured_cost(engine, graph, link_index, num_hops, mode="pos"):
    """The engine's own per-link cost (`Engine.link_costs`: the sizing pass alone + the cost model
    of s3grl_plan_link_cost) — what SURVEY §8(e) calls balancing by "the plan pass's exact vol(S)".
    One cheap pass over the whole list at set-up time, identical on every rank.  A reversed
    duplicate that the engine folds into its primary is priced as what it costs (two more output
    rows), the primary in full: with pair-aware shards the pair's cost lands on one rank."""
    return engine.link_costs(graph, engine.links(link_index), num_hops=num_hops, mode=mode).cpu().numpy().astype(np.float64)


def sop_cost(engine, graph, A, link_index):
    """Per-link cost of the SoP flow: the row kernel reads the same 2(K+1) table rows for every link,
    the scalar phase grows with the 1-hop ball (deg src + deg dst), and a reversed duplicate that the
    engine folds into its primary costs its output rows only (it is recognised by the sizing pass of a
    count-only plan, like for PoS)."""
    folded = engine.folded_mask(graph, engine.links(link_index)).cpu().numpy()
    return np.where(folded, 8.0, link_cost(A, link_index) + 32.0)


def chunk_bounds(lo, hi, chunks, cost=None):
    """Cut [lo, hi) into `chunks` contiguous cost-balanced pieces (empty pieces allowed)."""
    c = None if cost is None else np.asarray(cost)[lo:hi]
    return [lo + x for x in shard_bounds(hi - lo, chunks, c)]


class _Done:
    def wait(self):
        return True


def _all_gather(out, inp, group, async_op=False):
    import torch.distributed as dist

    if out.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (more ranks than GPUs on one box, bench.py S3GRL_BENCH_BACKEND=gloo): gloo
        # moves host memory, so the contribution is staged through the host, synchronously
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, inp.cpu(), group=group)
        out.copy_(h_out)
        return _Done()
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)


def sharded_precompute(compute, link_index, *, rank, world_size, cost=None, group=None, gather=True,
                       rows_per_link=None, chunks=1, row_shape=None, dtype=torch.float32,
                       device=None, timers=None, collective_at_world1=False, reuse_buffers=False,
                       pair_aware=False, shards=None, local_operator0=None, mirror_rows=False):
    """Shard `link_index` ([2, L]) over the ranks and (when `gather`) reassemble the whole result
    on every rank.

    Ragged flavour (`rows_per_link is None`):
        `compute(link_index_shard) -> (rows [R_r, ...], row_ptr [L_r + 1])` on this rank's device.
    Fixed flavour (`rows_per_link = 2` for PoS / SoP / hybrid; needs `row_shape`, `device`):
        `compute(link_index_piece, out)` fills `out` ([piece links * rows_per_link, *row_shape]).

    Returns (rows, row_ptr, where): the WHOLE list (in the caller's order) on every rank when
    `gather`, else the local shard; `where` = (lo, hi) for contiguous shards, or — `pair_aware`, see
    `shard_assignment` — the int64 tensor of the list positions this rank computed (the rows of a
    local shard are in that order).  `shards`: a `ShardPlan` made once for this list (a step must not
    redo the assignment); `cost` / `pair_aware` are then taken from it.  `local_operator0` (fixed
    flavour): a callable `fill(final [L, rows_per_link, *row_shape])` that writes operator 0 of every
    link (`[:, :, 0, :]` = [z | X[node]], which every rank can form from the replicated X and the
    link list) — the ranks then exchange operators 1..K only: 1 / (K+1) fewer bytes on the wire, and
    at 8 ranks the step is bound by the wire.  `mirror_rows` (fixed flavour, pair-aware `shards`,
    rows_per_link = 2): the caller vouches that the rows of a link (d, s) are the rows of (s, d) in
    swapped order, bit for bit (PoS and SoP: the engine itself serves a reversed duplicate that way) —
    a link that follows its reverse in a rank's piece then does not travel: every rank rebuilds it from
    the primary it received (PubMed's list: 23 % of the links).  `timers` (dict, optional) receives
    host-side timestamps.
    `collective_at_world1`: run the pieces / in-place all-gather / compaction path even on a
    one-rank group (a test hook: it is how the collective code meets real RCCL on a one-GPU box).
    `reuse_buffers` (fixed flavour): the returned `rows` live in a process-wide buffer that the NEXT
    call with the same shape overwrites — for a benchmark loop that must not allocate GBs per step.
    By default the caller owns what it gets back (the reference calls pos then neg per split with
    equal counts, sgrl_link_pred.py:195-204: both results must stay valid); only the padded
    all-gather slots are ever shared between calls.
    """
    li = torch.as_tensor(link_index)
    L = int(li.shape[1])
    if shards is None and pair_aware:
        shards = ShardPlan(li, world_size, cost, True)
    if shards is not None and not shards.identity:
        assert shards.num_links == L and len(shards.bounds) == world_size + 1
        assert shards.rep_start == L or (gather and rows_per_link is not None), \
            "replicated links belong to the gathered fixed-rows flavour"
        order, order_t, b = shards.order, shards.order_dev, shards.bounds
        li, cost = shards.links, shards.cost      # the list, grouped by rank; positions map back through `order`
    elif shards is not None:
        order, order_t, b, cost = None, None, shards.bounds, shards.cost
    else:
        order, order_t = None, None
        b = shard_bounds(L, world_size, cost)
    lo, hi = b[rank], b[rank + 1]
    where = (lo, hi) if order is None else order_t[lo:hi]
    if order is not None and shards.rep_start < L:      # + the links every rank computes itself
        where = torch.cat([where, order_t[shards.rep_start:]])
    if rows_per_link is not None:
        rows, row_ptr, _ = _fixed(compute, li, b, rank, world_size, cost, group, gather, int(rows_per_link),
                                  max(int(chunks), 1), tuple(row_shape), dtype, device, timers,
                                  collective_at_world1, bool(reuse_buffers), order_t, local_operator0,
                                  shards if (mirror_rows and shards is not None and not shards.identity
                                             and int(rows_per_link) == 2) else None,
                                  L if shards is None else shards.rep_start)
        return rows, row_ptr, where
    rows, row_ptr = compute(li[:, lo:hi])
    if not gather or world_size == 1:
        if order is not None and gather:      # one rank, whole list: back into the caller's order
            rows, row_ptr = _reorder_ragged(rows, row_ptr, order_t.to(rows.device))
        return rows, row_ptr, where
    dev = rows.device
    # 1) sizes
    mine = torch.tensor([rows.shape[0], hi - lo], dtype=torch.int64, device=dev)
    sizes = torch.empty(world_size * 2, dtype=torch.int64, device=dev)
    _all_gather(sizes, mine, group)
    sizes = sizes.cpu().view(world_size, 2)
    rmax, lmax = int(sizes[:, 0].max()), int(sizes[:, 1].max())
    # 2) one padded all-gather of the rows, one of the per-link row counts
    tail = rows.shape[1:]
    pad = torch.zeros((rmax,) + tuple(tail), dtype=rows.dtype, device=dev)
    pad[:rows.shape[0]] = rows
    allrows = torch.empty((world_size * rmax,) + tuple(tail), dtype=rows.dtype, device=dev)
    _all_gather(allrows, pad, group)
    cnt = torch.zeros(lmax, dtype=torch.int64, device=dev)
    cnt[:hi - lo] = (row_ptr[1:] - row_ptr[:-1]).to(dev)
    allcnt = torch.empty(world_size * lmax, dtype=torch.int64, device=dev)
    _all_gather(allcnt, cnt, group)
    out_rows = torch.cat([allrows[r * rmax: r * rmax + int(sizes[r, 0])] for r in range(world_size)])
    counts = torch.cat([allcnt[r * lmax: r * lmax + int(sizes[r, 1])] for r in range(world_size)])
    out_ptr = torch.zeros(L + 1, dtype=torch.int64, device=dev)
    out_ptr[1:] = torch.cumsum(counts, 0)
    if order is not None:
        out_rows, out_ptr = _reorder_ragged(out_rows, out_ptr, order_t.to(dev))
    return out_rows, out_ptr, where


def _reorder_ragged(rows, row_ptr, order):
    """rows / row_ptr of the links in shard order (position i = list position order[i]) -> the
    caller's order."""
    L = order.numel()
    cnt = row_ptr[1:] - row_ptr[:-1]
    out_cnt = torch.empty_like(cnt)
    out_cnt[order] = cnt
    out_ptr = torch.zeros(L + 1, dtype=torch.int64, device=rows.device)
    out_ptr[1:] = torch.cumsum(out_cnt, 0)
    # destination row of every source row: start of its link in the caller's order + offset in the link
    dst_start = out_ptr[:-1][order]
    total = int(row_ptr[-1])
    dst = torch.repeat_interleave(dst_start - row_ptr[:-1], cnt, output_size=total) + \
        torch.arange(total, device=rows.device)
    out = torch.empty_like(rows)
    out[dst] = rows
    return out, out_ptr


class _Buffers:
    """Reused across calls: the padded all-gather slots (two, so that piece c+1 is computed while
    piece c is in flight) and — only with `reuse_buffers=True` — the rows handed back."""
    cache = {}

    @classmethod
    def get(cls, key, shape, dtype, device):
        t = cls.cache.get(key)
        if t is None or t.shape != tuple(shape) or t.dtype != dtype or t.device != torch.device(device):
            t = torch.empty(shape, dtype=dtype, device=device)
            cls.cache[key] = t
        return t

    @classmethod
    def clear(cls):
        cls.cache.clear()


def _result(key, shape, dtype, device, reuse):
    """Memory of a returned tensor: the caller's own unless it asked for the shared buffer."""
    if reuse:
        return _Buffers.get(key, shape, dtype, device)
    return torch.empty(shape, dtype=dtype, device=device)


def _fixed(compute, li, b, rank, world, cost, group, gather, rpl, chunks, row_shape, dtype, device,
           timers, collective_at_world1=False, reuse=False, order=None, local_op0=None, mirrors=None,
           rep_start=None):
    """`li` is the list grouped by rank (rank r owns columns b[r]:b[r+1]); `order` (device-resident
    positions in the caller's list, or None = identity) says where every column belongs;
    `local_op0`: see `sharded_precompute(local_operator0=…)`; `mirrors`: the ShardPlan when reversed
    duplicates are rebuilt from their primaries instead of exchanged (`mirror_rows=True`); `rep_start`:
    the columns from there on are computed by every rank itself (ShardPlan(replicate=…)) while the pieces
    of the others are on the wire."""
    L = int(li.shape[1])
    lo, hi = b[rank], b[rank + 1]
    row_ptr = torch.arange(0, rpl * L + 1, rpl, dtype=torch.int64, device=device)
    if order is not None:
        order = order.to(device)
    if not gather or (world == 1 and not collective_at_world1):
        if gather and order is not None:
            # one rank, whole list, but grouped by a pair-aware plan (the replicated links at its end):
            # computed in that order — the folds of the plan survive — and put back into the caller's
            final = _result(("final", rank), (rpl * L,) + row_shape, dtype, device, reuse)
            tmp = _Buffers.get(("grouped", rank), (rpl * L,) + row_shape, dtype, device)
            if L:
                compute(li, tmp)
                final.view((L, rpl) + row_shape).index_copy_(0, order, tmp.view((L, rpl) + row_shape))
            return final, row_ptr, (lo, hi)
        rows = _result(("local", rank), (rpl * (hi - lo),) + row_shape, dtype, device, reuse)
        compute(li[:, lo:hi], rows)
        return rows, row_ptr[lo:hi + 1] - rpl * lo, (lo, hi)
    # piece c of rank r = links [pb[r][c], pb[r][c+1]); every rank derives every rank's bounds
    if mirrors is not None:
        pb, prim, mir, nprim = mirrors.transport(chunks, device)
    else:
        pb = [chunk_bounds(b[r], b[r + 1], chunks, cost) for r in range(world)]
        nprim = [[pb[r][c + 1] - pb[r][c] for c in range(chunks)] for r in range(world)]
    final = _result(("final", rank), (rpl * L,) + row_shape, dtype, device, reuse)
    works = [None] * chunks
    slots = [None] * chunks
    pmaxes = [rpl * max(nprim[r][c] for r in range(world)) for c in range(chunks)]
    cap = max(max(pmaxes), 1)
    # what travels: whole rows, or operators 1..K when every rank fills operator 0 itself
    xshape = row_shape if local_op0 is None else (row_shape[0] - 1,) + row_shape[1:]

    final_links = final.view((L, rpl) + row_shape)
    final_x = final if local_op0 is None else final[:, 1:]
    final_links_x = final_links if local_op0 is None else final_links[:, :, 1:]

    def compact(c):
        slot, pmax = slots[c]
        for r in range(world):
            p0, p1 = pb[r][c], pb[r][c + 1]
            n = rpl * (p1 - p0)
            if not n:
                continue
            if mirrors is not None:      # what travelled goes to its list position; the reversed duplicates
                npr = nprim[r][c]        # are their primaries' rows in swapped order
                got = slot[r, :rpl * npr].view((npr, rpl) + xshape)
                pos = order[p0:p1]
                final_links_x.index_copy_(0, pos[prim[r][c]], got)
                mpos, mslot = mir[r][c]
                if mpos.numel():
                    final_links_x.index_copy_(0, pos[mpos], got.index_select(0, mslot).flip(1))
                continue
            if order is None:
                final_x[rpl * p0: rpl * p0 + n].copy_(slot[r, :n], non_blocking=True)
            else:      # scatter by list position: 2 rows x K(+1) x (1+F) floats per link
                final_links_x.index_copy_(0, order[p0:p1], slot[r, :n].view((p1 - p0, rpl) + xshape))

    rep_start = L if rep_start is None else int(rep_start)

    def replicated():
        """the links nobody sends: computed here, into their list positions (pieces of at most the size of
        the exchange's slots, through the same staging buffer)"""
        if rep_start >= L:
            return
        step = max(cap // rpl, 1)
        buf = _Buffers.get(("rep", rank), (rpl * step,) + row_shape, dtype, device)
        for a in range(rep_start, L, step):
            z = min(a + step, L)
            compute(li[:, a:z], buf[:rpl * (z - a)])
            final_links.index_copy_(0, order[a:z], buf[:rpl * (z - a)].view((z - a, rpl) + row_shape))

    t_comm = 0.0
    op0_stream = None
    for c in range(chunks):
        pmax = max(pmaxes[c], 1)
        buf = _Buffers.get(("slot", rank, c % 2, local_op0 is None), (world * cap,) + xshape, dtype, device)
        slot = buf[:world * pmax].view((world, pmax) + xshape)
        slots[c] = (slot, pmax)
        p0, p1 = pb[rank][c], pb[rank][c + 1]
        if p1 > p0:
            n = rpl * (p1 - p0)
            if mirrors is not None:      # whole piece computed (the engine folds the duplicates), primaries packed
                full = _Buffers.get(("fullm", rank), (rpl * max(pb[rank][k + 1] - pb[rank][k] for k in range(chunks)),) +
                                    row_shape, dtype, device)
                compute(li[:, p0:p1], full[:n])
                sent = full[:n].view((p1 - p0, rpl) + row_shape).index_select(0, prim[rank][c])
                npr = nprim[rank][c]
                slot[rank, :rpl * npr].view((npr, rpl) + xshape).copy_(
                    sent if local_op0 is None else sent[:, :, 1:], non_blocking=True)
            elif local_op0 is None:
                compute(li[:, p0:p1], slot[rank, :n])
            else:      # the engine writes whole rows: pack operators 1..K into the slot
                full = _Buffers.get(("full", rank), (cap,) + row_shape, dtype, device)
                compute(li[:, p0:p1], full[:n])
                slot[rank, :n].copy_(full[:n, 1:], non_blocking=True)
        t0 = time.perf_counter()
        # in place: this rank's contribution already sits in its slice of the output
        works[c] = _all_gather(slot.view((world * slot.shape[1],) + xshape), slot[rank], group,
                               async_op=True)
        if c == chunks - 1:
            replicated()                    # while the pieces travel
        if c == 0 and local_op0 is not None:
            # operator 0 of the whole list: three indexed copies of X rows (0.7 GB on the headline) — on a side
            # stream, so that the next piece's kernels do not queue behind them
            if torch.device(device).type == "cuda":
                side = _Buffers.cache.get(("side_stream", rank))
                if side is None:
                    side = torch.cuda.Stream(device=device)
                    _Buffers.cache[("side_stream", rank)] = side
                main = torch.cuda.current_stream(device)
                side.wait_stream(main)          # `final` is ready (allocated / last read on the main stream)
                with torch.cuda.stream(side):
                    local_op0(final_links)
                final.record_stream(side)
                op0_stream = side
            else:
                local_op0(final_links)      # while the first pieces travel
        if c >= 1:
            works[c - 1].wait()
            compact(c - 1)
        t_comm += time.perf_counter() - t0
    t0 = time.perf_counter()
    works[chunks - 1].wait()
    compact(chunks - 1)
    if op0_stream is not None:
        torch.cuda.current_stream(device).wait_stream(op0_stream)
    t_comm += time.perf_counter() - t0
    if timers is not None:
        timers["comm_host_s"] = timers.get("comm_host_s", 0.0) + t_comm
    return final, row_ptr, (lo, hi)


def engine_compute(engine, graph, x, *, mode="pos", num_hops=1, sign_k=3, stats=None, **kw):
    """The fixed-flavour compute callable running the HIP engine on this rank's device:
    `compute(link_index_piece [2, n] (host or device), out)`.  `stats` (dict, optional) accumulates
    the plans' link and folded-link counts."""

    def compute(piece, out):
        links = engine.links(piece)
        if mode == "sop":
            engine.precompute(graph, x, links, mode="sop", sign_k=sign_k, out=out)
            return
        plan = engine.plan(graph, links, mode=mode, num_hops=num_hops, sign_k=sign_k, **kw)
        try:
            plan.run(x, out)
            if stats is not None:      # counts the host already holds: no wait for the plan's kernels
                stats["links"] = stats.get("links", 0) + plan.num_links
                stats["folded_links"] = stats.get("folded_links", 0) + plan.folded_links
        finally:
            plan.close()

    return compute
