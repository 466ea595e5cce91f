#!/usr/bin/env python3
"""Parity sweep of a tuning knob over every golden case (dtype x fp32-out): max / mean of the §8c metric per setting.
    python tools/fold_parity.py FOLD_OUT_PROJ 0 1"""
import glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tokenpacker_oracle as orc
from tokenpacker_amd import TokenPacker, _capi, synth

key = getattr(_capi, "TP_TUNE_" + sys.argv[1])
vals = [int(v) for v in sys.argv[2:]]
cases = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "s[0-9]_D*.npz")))
rows = []
for path in cases:
    z = np.load(path)
    s, D, B = int(z["scale_factor"]), int(z["hidden_size"]), int(z["batch"])
    params = synth.make_params(int(z["param_seed"]), D)
    x, xm = synth.make_inputs(int(z["input_seed"]), B)
    for dtype in (torch.bfloat16, torch.float16):
        p_lp = {k: v.to(dtype) for k, v in params.items()}
        y_exact = orc.forward(p_lp, x.to(dtype), xm.to(dtype), scale_factor=s, compute_dtype=torch.float64, io_dtype=dtype)
        errs = []
        for v in vals:
            _capi.set_tuning(key, v)
            m = TokenPacker(hidden_size=D, scale_factor=s)
            m.load_state_dict(params)
            m = m.to(device="cuda", dtype=dtype).eval().requires_grad_(False)
            m.output_fp32 = True
            with torch.no_grad():
                y = m((x.to(dtype).cuda(), xm.to(dtype).cuda()))
            errs.append((orc.rel_err(y, y_exact), orc.rel_l2(y, y_exact)))
        rows.append((os.path.basename(path)[:-4], str(dtype)[6:], errs))
        print(rows[-1][0], rows[-1][1], "  ".join(f"{sys.argv[1]}={v}: {e:.3e} (l2 {l:.3e})" for v, (e, l) in zip(vals, errs)), flush=True)
for i, v in enumerate(vals):
    print(f"{sys.argv[1]}={v}: worst rel_err {max(r[2][i][0] for r in rows):.3e}, mean {np.mean([r[2][i][0] for r in rows]):.3e}, worst l2 {max(r[2][i][1] for r in rows):.3e}")
