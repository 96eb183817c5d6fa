#!/usr/bin/env python3
"""How should a COLD call hand 1-3 GB of rows to the host?  (GPU box.)  Fresh page-locked memory costs its
page-locking (0.1-0.15 s per GB) before the 57 GB/s copy; pageable memory costs its page faults.  Times, for one
tensor of --gb gigabytes on the device: (a) a fresh pinned block + D2H, (b) `.cpu()`, (c) D2H through a small
pinned ring + multi-threaded CPU copy into fresh pageable memory, for a few ring sizes and thread counts."""
import argparse
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=1.2)
    a = ap.parse_args()
    n = int(a.gb * (1 << 30) / 4)
    dev = torch.device("cuda:0")
    src = torch.rand(n, device=dev)
    torch.cuda.synchronize()

    def t(fn, name):
        torch.cuda.synchronize()
        c = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - c
        print(f"{name:60s} {dt * 1e3:8.1f} ms  {a.gb / dt:6.2f} GB/s", flush=True)
        return r

    def pinned():
        h = torch.empty(n, dtype=torch.float32, pin_memory=True)
        h.copy_(src, non_blocking=True)
        return h

    r = t(pinned, "fresh pinned block + D2H")
    del r
    r = t(lambda: src.cpu(), "src.cpu() (fresh pageable)")
    del r
    print("torch threads", torch.get_num_threads())
    for chunk_mb in (16, 64, 256):
        for threads in (1, 4, 8, 16):
            torch.set_num_threads(threads)
            ce = chunk_mb * (1 << 20) // 4
            ring = [torch.empty(ce, dtype=torch.float32, pin_memory=True) for _ in range(2)]
            evs = [torch.cuda.Event() for _ in range(2)]
            copy_stream = torch.cuda.Stream()

            def ringcopy():
                out = torch.empty(n, dtype=torch.float32)
                k = 0
                pend = []
                for off in range(0, n, ce):
                    m = min(ce, n - off)
                    slot = k % 2
                    if len(pend) == 2:          # the slot's previous content must have been consumed
                        o2, m2, s2 = pend.pop(0)
                        evs[s2].synchronize()
                        out[o2:o2 + m2].copy_(ring[s2][:m2])
                    with torch.cuda.stream(copy_stream):
                        ring[slot][:m].copy_(src[off:off + m], non_blocking=True)
                        evs[slot].record(copy_stream)
                    pend.append((off, m, slot))
                    k += 1
                for o2, m2, s2 in pend:
                    evs[s2].synchronize()
                    out[o2:o2 + m2].copy_(ring[s2][:m2])
                return out

            r = t(ringcopy, f"ring 2 x {chunk_mb} MB, {threads} CPU threads -> fresh pageable")
            assert torch.equal(r[:1000], src[:1000].cpu()) and torch.equal(r[-1000:], src[-1000:].cpu())
            del r, ring
    # the same into memory that asked for transparent huge pages (2 MiB faults instead of 4 KiB ones)
    import ctypes

    libc = ctypes.CDLL("libc.so.6", use_errno=True)

    def huge_empty(count):
        out = torch.empty(count + (1 << 19), dtype=torch.float32)      # room to start on a 2 MiB boundary
        p = out.data_ptr()
        lo = (p + (1 << 21) - 1) & ~((1 << 21) - 1)
        ln = (p + out.numel() * 4 - lo) & ~((1 << 21) - 1)
        rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(ln), 14)   # MADV_HUGEPAGE
        off = (lo - p) // 4
        return out[off:off + count], rc

    for threads in (1, 4, 16):
        torch.set_num_threads(threads)
        ce = 64 * (1 << 20) // 4
        ring = [torch.empty(ce, dtype=torch.float32, pin_memory=True) for _ in range(2)]
        evs = [torch.cuda.Event() for _ in range(2)]
        copy_stream = torch.cuda.Stream()

        def ringcopy_huge():
            out, rc = huge_empty(n)
            pend, k = [], 0
            for off in range(0, n, ce):
                m = min(ce, n - off)
                slot = k % 2
                if len(pend) == 2:
                    o2, m2, s2 = pend.pop(0)
                    evs[s2].synchronize()
                    out[o2:o2 + m2].copy_(ring[s2][:m2])
                with torch.cuda.stream(copy_stream):
                    ring[slot][:m].copy_(src[off:off + m], non_blocking=True)
                    evs[slot].record(copy_stream)
                pend.append((off, m, slot))
                k += 1
            for o2, m2, s2 in pend:
                evs[s2].synchronize()
                out[o2:o2 + m2].copy_(ring[s2][:m2])
            return out

        r = t(ringcopy_huge, f"ring 2 x 64 MB, {threads} threads -> pageable + MADV_HUGEPAGE")
        assert torch.equal(r[-1000:], src[-1000:].cpu())
        del r

    def registered():
        out, rc = huge_empty(n)
        out.zero_()                                   # fault it in (huge pages)
        rt = torch.cuda.cudart()
        e = rt.cudaHostRegister(out.data_ptr(), out.numel() * 4, 0)
        out.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        rt.cudaHostUnregister(out.data_ptr())
        return out

    torch.set_num_threads(16)
    r = t(registered, "pageable + MADV_HUGEPAGE, touched, cudaHostRegister, direct D2H")
    assert torch.equal(r[-1000:], src[-1000:].cpu())
    try:
        print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
    except OSError as e:
        print("THP: ?", e)


if __name__ == "__main__":
    main()
