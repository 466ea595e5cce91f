#!/usr/bin/env python3
"""Does a buffer that was JUST written come back out of the Infinity Cache (256 MiB, memory-side) faster than out of HBM?
Write N MiB (a fill kernel), read it back (a sum) — directly behind the write, and with 1 GiB of other traffic in between.
Decides whether chunking producer -> consumer pairs of the s >= 3 query side (qt: 537 MB written by one kernel, read by the next)
could take their round trip off the HBM interface."""
import torch
dev = torch.device("cuda", 0)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GiB of other traffic
def t(fn, n=5):
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for mib in (32, 64, 128, 192, 256, 512):
    n = mib * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    res = {}
    for label, between in (("read right behind the write", False), ("1 GiB of other writes in between", True)):
        times = []
        for _ in range(5):
            a.fill_(1.0)
            if between:
                big.fill_(2.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s = a.sum(); e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        res[label] = min(times)
    w = t(lambda: a.fill_(1.0))
    print(f"{mib:4d} MiB: write {w*1e3:7.1f} us ({mib/1024/(w*1e-3):5.2f} TB/s) | " + " | ".join(f"{k}: {v*1e3:7.1f} us ({mib*1.048576e-3/(v):5.2f} TB/s)" for k, v in res.items()), flush=True)
