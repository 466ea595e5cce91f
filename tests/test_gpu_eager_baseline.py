"""The reference's op sequence on PyTorch-ROCm eager (oracle/reference_ops.py): sanity vs the HIP path
and the timing that BASELINE.json's ">= 5x the reference PyTorch-ROCm projector" is measured
against.  Writes gpurun_out/eager_rocm.json."""
import json
import os
import time

import pytest
import torch

from oracle import tokenpacker_oracle as orc
from oracle.reference_ops import eager_forward
from tokenpacker_amd import TokenPacker, synth

pytestmark = pytest.mark.gpu


def test_eager_port_matches_oracle_cpu_semantics_on_gpu():
    dtype, s, D, B = torch.float32, 3, 256, 2
    params = synth.make_params(21, D)
    m = TokenPacker(hidden_size=D, scale_factor=s)
    m.load_state_dict(params)
    m = m.cuda().eval().requires_grad_(False)
    x, xm = synth.make_inputs(22, B)
    with torch.no_grad():
        y = eager_forward(m, x.cuda(), xm.cuda())
    y_ref = orc.forward(params, x, xm, scale_factor=s, compute_dtype=torch.float64)
    assert orc.rel_err(y, y_ref) < 1e-4


@pytest.mark.parametrize("s", [2, 3, 4])
def test_hip_vs_eager_rocm_speed_and_agreement(s):
    dtype, D, B = torch.bfloat16, 4096, 256
    torch.manual_seed(0)
    m = TokenPacker(hidden_size=D, scale_factor=s).to(device="cuda", dtype=dtype).eval().requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 576, 1024, generator=g, device="cuda").to(dtype)
    xm = torch.randn(B, 576, 4096, generator=g, device="cuda").to(dtype)

    def timed(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    with torch.no_grad():
        y_hip = m((x, xm))
        y_eager = eager_forward(m, x, xm)
        agree = orc.rel_err(y_hip[:8], y_eager[:8].float())
        ms_eager = timed(lambda: eager_forward(m, x, xm), 10)
        ms_hip = timed(lambda: m((x, xm)), 20)
    rec = {"scale_factor": s, "B": B, "D": D, "dtype": "bf16", "eager_rocm_ms": round(ms_eager, 3),
           "hip_ms": round(ms_hip, 3), "speedup": round(ms_eager / ms_hip, 3),
           "eager_images_per_s": round(B / ms_eager * 1e3, 1), "hip_images_per_s": round(B / ms_hip * 1e3, 1),
           "rel_err_hip_vs_eager_bf16": agree}
    print("\n[eager-baseline]", json.dumps(rec))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/eager_rocm_s{s}.json", "w") as f:
        json.dump(rec, f)
    assert agree < 2e-2       # two bf16 pipelines with different rounding points
