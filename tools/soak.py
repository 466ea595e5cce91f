#!/usr/bin/env python3
"""Race soak: the same forward (and training step) over and over on fixed inputs — every output must equal the first one bit for bit.  The
kernels' hand-placed waits / barriers are proven on paper and screened in the suite for a few hundred launches; this runs them for minutes.

    python tools/soak.py [--seconds 60] [--out gpurun_out/soak.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenpacker_amd import TokenPacker, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0, help="per configuration")
    ap.add_argument("--out", default="gpurun_out/soak.json")
    a = ap.parse_args()
    res = []
    for (s, B, train) in [(2, 256, False), (2, 32, False), (2, 10, False), (2, 1, False), (3, 256, False), (4, 64, False), (2, 32, True)]:
        D, dtype = 4096, torch.bfloat16
        params = synth.make_params(900 + s, D)
        m = TokenPacker(hidden_size=D, scale_factor=s)
        m.load_state_dict(params)
        m = m.to(device="cuda", dtype=dtype)
        x, xm = synth.make_inputs(901, min(B, 8), dtype)
        reps = (B + x.shape[0] - 1) // x.shape[0]
        xg, xmg = x.repeat(reps, 1, 1)[:B].cuda(), xm.repeat(reps, 1, 1)[:B].cuda()
        if train:
            m.train().requires_grad_(True)

            def step():
                m.zero_grad(set_to_none=True)
                y = m((xg, xmg))
                y.float().square().mean().backward()
                return torch.cat([y.flatten().float()[:65536]] + [p.grad.flatten().float()[:4096] for p in m.parameters()])
        else:
            m.eval().requires_grad_(False)

            def step():
                with torch.no_grad():
                    return m((xg, xmg))
        ref = step().clone()
        torch.cuda.synchronize()
        t0, n, bad = time.time(), 0, 0
        while time.time() - t0 < a.seconds:
            outs = [step().clone() for _ in range(8)]
            torch.cuda.synchronize()
            for o in outs:
                n += 1
                if not torch.equal(o, ref):
                    bad += 1
        r = {"scale_factor": s, "B": B, "training_step": train, "launches": n, "mismatches": bad, "seconds": round(time.time() - t0, 1)}
        res.append(r)
        print(r, flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    sys.exit(1 if any(r["mismatches"] for r in res) else 0)


if __name__ == "__main__":
    main()
