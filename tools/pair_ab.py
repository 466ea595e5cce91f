#!/usr/bin/env python3
"""Same-process A/B of the pair GEMM kernel (tp_gemm_pair.hip) against the ping-pong kernel on the whole forward.

Arms = settings of the tuning table (TP_TUNE_PAIR_GEMM / TP_TUNE_PAIR_STAGGER / anything given with --arm), interleaved
round-robin inside ONE process on the bench's workload (guide §5.4 rule 24); per arm: median / min ms per forward over the rounds
and the per-stage breakdown from tp_forward_staged's events.

    python tools/pair_ab.py [--batches 256 32] [--rounds 7] [--iters 10] [--arm name:KEY=V,KEY=V ...]
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tokenpacker_amd import _capi  # noqa: E402

DEFAULT_ARMS = ["off:PAIR_GEMM=1", "default:PAIR_GEMM=0", "pair_forced:PAIR_GEMM=2", "pair_forced_nostagger:PAIR_GEMM=2,PAIR_STAGGER=0"]


def parse_arm(spec):
    name, _, kvs = spec.partition(":")
    sets = []
    for kv in filter(None, kvs.split(",")):
        k, v = kv.split("=")
        sets.append((getattr(_capi, "TP_TUNE_" + k.upper()), int(v)))
    return name, sets


def apply(sets):
    for k, v in _capi._TUNING_DEFAULTS.items():
        _capi.set_tuning(k, v)
    for k, v in sets:
        _capi.set_tuning(k, v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[256, 32])
    ap.add_argument("--scale-factor", type=int, default=2)
    ap.add_argument("--hidden-size", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--arm", action="append", default=[])
    ap.add_argument("--out", default="gpurun_out/pair_ab.json")
    args = ap.parse_args()
    arms = [parse_arm(a) for a in (args.arm or DEFAULT_ARMS)]
    device = torch.device("cuda:0")
    dtype = torch.bfloat16
    model = bench.build_model(args.hidden_size, args.scale_factor, dtype, device)
    results = []
    for B in args.batches:
        x, xm = bench.make_device_inputs(B, dtype, "tower", device, 1234)
        times = {n: [] for n, _ in arms}
        stages = {n: [] for n, _ in arms}
        outs = {}
        with torch.no_grad():
            for n, sets in arms:                           # warm-up (packs once, allocates the workspace) + bit-identity
                apply(sets)
                for _ in range(3):
                    y = model((x, xm))
                torch.cuda.synchronize()
                outs[n] = y.clone()
            ref = outs[arms[0][0]]
            same = {n: bool(torch.equal(outs[n], ref)) for n, _ in arms}
            for _ in range(args.rounds):
                for n, sets in arms:
                    apply(sets)
                    model((x, xm))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(args.iters):
                        model((x, xm))
                    e1.record()
                    torch.cuda.synchronize()
                    times[n].append(e0.elapsed_time(e1) / args.iters)
                    _, evs = model.forward_staged((x, xm))
                    torch.cuda.synchronize()
                    stages[n].append([evs[i].elapsed_time(evs[i + 1]) for i in range(_capi.TP_NUM_STAGES)])
        apply([])
        for n, _ in arms:
            st = [round(statistics.median(s[i] for s in stages[n]), 4) for i in range(_capi.TP_NUM_STAGES)]
            rec = dict(batch=B, arm=n, ms=round(statistics.median(times[n]), 4), ms_min=round(min(times[n]), 4),
                       img_per_s=round(B / statistics.median(times[n]) * 1e3, 1), bit_identical_to_first_arm=same[n],
                       stages_ms=dict(zip(_capi.STAGE_NAMES, st)), staged_sum=round(sum(st), 4))
            results.append(rec)
            print(json.dumps(rec), flush=True)
        del x, xm
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
