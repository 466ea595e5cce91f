#!/usr/bin/env python3
"""What consuming the tower's four hidden-state slices directly is worth: the reference's feature_select
(torch.cat of 4 layers + [:,1:], clip_encoder.py:28-44) followed by the projector, vs tower.select_features +
the projector's parts path (tp_forward_parts).  B = 256, bf16, one MI355X."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import TokenPacker, tower  # noqa: E402


def timed(fn, iters=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    B, dtype = 256, torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    hs = [None] * 25
    for l in (12, 16, 22, 23):
        hs[l] = torch.randn(B, 577, 1024, generator=g, device="cuda").to(dtype)
    m = TokenPacker(hidden_size=4096, scale_factor=2).to(device="cuda", dtype=dtype).eval().requires_grad_(False)
    with torch.no_grad():
        def ref_path():
            x, xm = tower.concat_reference(hs)
            return m((x, xm))

        def parts_path():
            x, parts = tower.select_features(hs)
            return m((x, parts))

        def cat_only():
            return tower.concat_reference(hs)

        assert torch.equal(ref_path(), parts_path())
        rec = {"B": B, "cat_plus_projector_ms": round(timed(ref_path), 3), "parts_projector_ms": round(timed(parts_path), 3),
               "torch_cat_alone_ms": round(timed(cat_only), 3)}
    rec["saved_ms"] = round(rec["cat_plus_projector_ms"] - rec["parts_projector_ms"], 3)
    print(json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rec, open("gpurun_out/parts_bench.json", "w"))


if __name__ == "__main__":
    main()
