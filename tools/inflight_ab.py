#!/usr/bin/env python3
"""Independent forwards in flight (round 5): a stream of B-image batches issued on ONE stream against the same stream of batches issued
round-robin on 2 / 3 streams — the half-empty tail rounds of one forward's persistent GEMMs (a 32-image shard: 1.125 / 1.5 / 2.25 CU rounds
per launch) can take another forward's workgroups when that forward is not ordered behind it.  Throughput of the stream, not latency of a batch.

    python tools/inflight_ab.py [--batches 32 64 128 256] [--out gpurun_out/inflight_ab.json]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[32, 64, 128, 256])
    ap.add_argument("--out", default="gpurun_out/inflight_ab.json")
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    dtype = torch.bfloat16
    model = bench.build_model(4096, 2, dtype, dev)
    res = {}
    for B in a.batches:
        x, xm = bench.make_device_inputs(B, dtype, "tower", dev, seed=1234)
        streams = [torch.cuda.Stream() for _ in range(3)]
        outs = {}

        def run(n_streams, steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                for i in range(steps):
                    if n_streams == 1:
                        y = model((x, xm))
                    else:
                        with torch.cuda.stream(streams[i % n_streams]):
                            y = model((x, xm))
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3, y

        for n in (1, 2, 3):
            _, outs[n] = run(n, 12)
        same = all(torch.equal(outs[1], outs[n]) for n in (2, 3))
        times = {1: [], 2: [], 3: []}
        for _ in range(a.rounds):
            for n in (1, 2, 3):
                times[n].append(run(n, a.steps)[0])
        r = {n: round(statistics.median(v), 4) for n, v in times.items()}
        res[f"B{B}"] = {"ms_per_forward": r, "images_per_s": {n: round(B / (t * 1e-3), 1) for n, t in r.items()},
                        "two_over_one": round(r[2] / r[1], 4), "three_over_one": round(r[3] / r[1], 4), "bit_identical": bool(same)}
        print(f"B={B}", res[f"B{B}"], flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
