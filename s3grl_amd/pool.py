"""Consumer-side centre / common-neighbour pooling on the engine's layout — the device-resident
replacement of reference `SIGNNet._centre_pool_helper` (models.py:339-369), differentiable.

    h = operator_diff(rows.reshape(total_rows, -1))          # [ΣR, hidden]
    z = centre_pool(h, row_ptr, k_heuristic=1, k_pool_strategy="mean")   # [B, 2*hidden]

Forward and backward are HIP kernels behind the C ABI (s3grl_centre_pool_forward/_backward);
there is no host sync (the reference calls np.unique on the CPU for every batch).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N
from .engine import default_engine

_MODES = {"": 0, None: 0, "mean": 1, "sum": 2}


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class _CentrePool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, row_ptr, mode):
        eng = default_engine(h.device)
        h = h.contiguous()
        B, H = row_ptr.numel() - 1, h.shape[1]
        out = torch.empty((B, H if mode == 0 else 2 * H), dtype=torch.float32, device=h.device)
        N.check(N.lib().s3grl_centre_pool_forward(eng._ctx, _ptr(h), _ptr(row_ptr), B, H, mode,
                                                  _ptr(out)), "s3grl_centre_pool_forward")
        ctx.save_for_backward(h, row_ptr)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, row_ptr = ctx.saved_tensors
        eng = default_engine(h.device)
        grad_out = grad_out.contiguous()
        gh = torch.empty_like(h)
        B, H = row_ptr.numel() - 1, h.shape[1]
        N.check(N.lib().s3grl_centre_pool_backward(eng._ctx, _ptr(h), _ptr(row_ptr), B, H, ctx.mode,
                                                   _ptr(grad_out), _ptr(gh)),
                "s3grl_centre_pool_backward")
        return gh, None, None


def centre_pool(h, row_ptr, k_heuristic=0, k_pool_strategy=""):
    """h fp32 [ΣR, H] on the GPU, row_ptr int64 [B+1] on the GPU."""
    if not h.is_cuda:
        raise RuntimeError("centre_pool runs on the MI355X only; there is no CPU fallback")
    if h.dtype != torch.float32 or row_ptr.dtype != torch.int64:
        raise ValueError("h must be float32 and row_ptr int64")
    if k_heuristic and k_pool_strategy not in ("mean", "sum", "concat"):
        raise NotImplementedError(f"Check pool strat: {k_pool_strategy}")   # models.py:335
    if k_heuristic and k_pool_strategy == "concat":
        # models.py:363-367: every link must carry exactly k_heuristic rows after its two centre
        # rows (the reference's reshape fails otherwise); they are appended side by side
        B, H, R = row_ptr.numel() - 1, h.shape[1], 2 + int(k_heuristic)
        if h.shape[0] != B * R:
            raise RuntimeError(f"shape '[{B}, {H * k_heuristic}]' is invalid: 'concat' pooling needs "
                               f"exactly {k_heuristic} common-neighbour rows per link")
        h_a = _CentrePool.apply(h, row_ptr, 0)
        h_k = h.view(B, R, H)[:, 2:, :].reshape(B, H * int(k_heuristic))
        return torch.cat([h_a, h_k], dim=-1)
    mode = _MODES[k_pool_strategy] if k_heuristic else 0
    return _CentrePool.apply(h, row_ptr, mode)


def row_ptr_from_batch(batch):
    """PyG-style graph-id vector (sorted, as `follow_batch` emits it) -> row_ptr, on the device,
    without the host round trip of np.unique (reference models.py:341)."""
    _, counts = torch.unique_consecutive(batch, return_counts=True)
    rp = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=batch.device)
    rp[1:] = torch.cumsum(counts, 0)
    return rp
