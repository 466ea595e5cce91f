#!/usr/bin/env python3
"""Time of one weight pack (inference image / training image), HIP events, median of 5 — what bench.py reports as pack_ms."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

device = torch.device("cuda:0")
for dtype in (torch.bfloat16,):
    model = bench.build_model(4096, 2, dtype, device)
    stream_ptr = torch.cuda.current_stream(device).cuda_stream
    for train in (False, True):
        ts = []
        for _ in range(6):
            model._packed_key = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); model._ensure_packed(dtype, device, stream_ptr, force=train); e1.record()
            torch.cuda.synchronize(device)
            ts.append(e0.elapsed_time(e1))
        print(f"pack_ms {'train_pack' if train else 'inference'}: median {sorted(ts[1:])[2]:.3f} ms  all {[round(t, 2) for t in ts]}", flush=True)
