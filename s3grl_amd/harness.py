"""Thin end-to-end harness for the downstream step: a SIGNNet twin that
consumes the engine's collated output directly on the device — no per-link Python objects, no
host sync in the pooling — and a train / evaluate loop shaped like the reference's
(`train_bce` sgrl_link_pred.py:440-472, `test` :538-587, AUC :704-770).

The model mirrors reference models.py:301-383: `operator_diff` = Linear -> ELU -> BatchNorm ->
dropout over the concatenated operators (PyG MLP with act_first, plain_last=False), centre /
common-neighbour pooling (`s3grl_amd.pool.centre_pool`, HIP kernels), `link_pred_mlp` = Linear
-> ReLU -> BatchNorm -> dropout -> Linear.  It exists to show that what the engine emits is what
the reference's MLP consumes; it is plain PyTorch apart from the pooling.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .pool import centre_pool


class SIGNNetTwin(nn.Module):
    def __init__(self, in_width, hidden=256, k_heuristic=0, k_pool_strategy="", dropout=0.5):
        """in_width = (sign_k + 1) * (1 + F): one collated row (reference models.py:316-320)."""
        super().__init__()
        self.k_heuristic, self.k_pool_strategy = k_heuristic, k_pool_strategy
        self.operator_diff = nn.Sequential(nn.Linear(in_width, hidden), nn.ELU(),
                                           nn.BatchNorm1d(hidden), nn.Dropout(dropout))
        # reference models.py:327-337: hidden x 2 for mean / sum pooling of the common-neighbour rows,
        # hidden x (1 + k_heuristic) when they are concatenated, hidden alone without the heuristic
        if not k_heuristic:
            ch = 1
        elif k_pool_strategy == "concat":
            ch = 1 + int(k_heuristic)
        else:
            ch = 2
        self.link_pred_mlp = nn.Sequential(nn.Linear(hidden * ch, hidden), nn.ReLU(),
                                           nn.BatchNorm1d(hidden), nn.Dropout(dropout),
                                           nn.Linear(hidden, 1))

    def forward(self, rows, row_ptr):
        """rows [ΣR_b, K+1, 1+F] of the links of one mini-batch, row_ptr [B+1] local to it."""
        h = self.operator_diff(rows.reshape(rows.shape[0], -1))
        z = centre_pool(h, row_ptr, self.k_heuristic, self.k_pool_strategy)
        return self.link_pred_mlp(z).view(-1)


def batch_slices(row_ptr, link_ids, total=None):
    """Device-side gather of the rows of a set of links: (row index [ΣR_b], local row_ptr [B+1]).
    `total` = ΣR_b when the caller knows it on the host (`RowCounts`): then nothing here waits for
    the device; without it the size is read back (one sync)."""
    start, end = row_ptr[link_ids], row_ptr[link_ids + 1]
    cnt = end - start
    local = torch.zeros(link_ids.numel() + 1, dtype=torch.int64, device=row_ptr.device)
    local[1:] = torch.cumsum(cnt, 0)
    if total is None:
        total = int(local[-1])
    idx = torch.repeat_interleave(start - local[:-1], cnt, output_size=total) + torch.arange(
        total, device=row_ptr.device)
    return idx, local


class RowCounts:
    """Rows per link on the HOST, fetched once per dataset: the loader draws its mini-batches on the
    host (like the reference's DataLoader) and knows every batch's row total without asking the
    device — no host sync per mini-batch (the reference has one per batch, models.py:341)."""

    def __init__(self, row_ptr):
        self.counts = (row_ptr[1:] - row_ptr[:-1]).cpu()

    def total(self, link_ids_host):
        return int(self.counts[link_ids_host].sum())


def auc_score(scores, labels):
    """Rank-based AUC (ties averaged), on the device."""
    s = scores.double()
    order = torch.argsort(s)
    ranks = torch.empty_like(s)
    ranks[order] = torch.arange(1, s.numel() + 1, dtype=torch.float64, device=s.device)
    # average ranks of ties
    uniq, inv, cnt = torch.unique(s, return_inverse=True, return_counts=True)
    sums = torch.zeros_like(uniq).scatter_add_(0, inv, ranks)
    ranks = (sums / cnt.double())[inv]
    pos = labels > 0
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2) / max(n_pos * n_neg, 1))


def train_and_evaluate(train, test, *, k_heuristic=0, k_pool_strategy="", hidden=256, epochs=10,
                       batch_size=32, lr=1e-3, seed=0, dropout=0.5):
    """train / test: (rows, row_ptr, y) device tensors as `Engine.precompute` returns them."""
    torch.manual_seed(seed)
    rows, row_ptr, y = train
    dev = rows.device
    model = SIGNNetTwin(rows.shape[1] * rows.shape[2], hidden, k_heuristic, k_pool_strategy,
                        dropout).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    L = y.numel()
    yf = y.float()
    counts = RowCounts(row_ptr)
    for _ in range(epochs):
        model.train()
        perm = torch.randperm(L)
        for b in range(0, L - 1, batch_size):
            ids_h = perm[b:b + batch_size]
            if ids_h.numel() < 2:
                continue
            ids = ids_h.to(dev, non_blocking=True)
            idx, local = batch_slices(row_ptr, ids, counts.total(ids_h))
            loss = nn.functional.binary_cross_entropy_with_logits(model(rows[idx], local), yf[ids])
            opt.zero_grad()
            loss.backward()
            opt.step()
    model.eval()
    rows_t, ptr_t, y_t = test
    out = []
    counts_t = RowCounts(ptr_t)
    with torch.no_grad():
        for b in range(0, y_t.numel(), 1024):
            ids_h = torch.arange(b, min(b + 1024, y_t.numel()))
            ids = ids_h.to(dev, non_blocking=True)
            idx, local = batch_slices(ptr_t, ids, counts_t.total(ids_h))
            out.append(model(rows_t[idx], local))
    return auc_score(torch.cat(out), y_t), model
