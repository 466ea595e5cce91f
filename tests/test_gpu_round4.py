"""Round-4 additions, on a real MI355X: the parity claim as a distribution over many seeds (gated on the WORST seed), tuning
contexts (tp_desc.tuning: a forward's knobs are its own, whatever other threads do to the process-wide table)."""
import json
import os
import sys
import threading

import pytest
import torch

from tokenpacker_amd import _capi, synth
from tests.test_gpu_forward import _module

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Stated claim (README / DESIGN §3), metric max|y - y_ref| / max|y_ref| against the fp64 oracle on the SAME rounded operands
# (SURVEY.md §8c), as a DISTRIBUTION over seeds (tools/parity_sweep.py, 128 seeds: profiles/r04e_parity_seed_sweep.json and
# r04f_parity_seed_sweep_s34_fp16.json):
#   scale_factor 2 (the north_star's gated configuration): every one of 128 seeds <= 8.9e-4 (median 6.0e-4 / 6.4e-4, rel-L2 <= 6.3e-4);
#   scale_factor 3, 4 (absorbed schedule): medians 6.5e-4 .. 6.8e-4, p90 <= 7.9e-4, rel-L2 <= 6.6e-4 — but the max-norm's tail is heavier
#   there (two fp16 roundings sit on the logit path, Q and qt, against one at s = 2): 127 of 128 seeds <= 1e-3, worst 1.09e-3.
# Round 4 removed three fp16 roundings that sat in series on the value path (the LayerNorm fold inside the pre-multiplied chain
# weights; on the absorbed schedule `u` and the pre-multiplied V weight, both carried as hi + lo): medians -10 .. -12 %.
# This test runs TP_PARITY_SEEDS seeds and gates the WORST seed, the p90 and the median.  Default 12 (round 5): every seed is six fp64
# CPU-oracle forwards, and the round-4 default of 48 (288 SERIAL oracle forwards in one test, ~1.4 s each) pushed the driver's
# `pytest -m gpu` past its 1200 s limit (GPUTEST_r04: rc 124).  The sweep now prepares its CPU side on 8 threads ahead of the GPU loop
# (4 seeds: 4.5 s on the r05q box; the whole 128-seed, 768-forward distribution: 130 s, profiles/r05q_parity_seed_sweep.json) — the
# 128-seed distribution stays an artefact of tools/parity_sweep.py, not something the default collection recomputes.
# (the absorbed schedule's query side: qt stays fp32 between the per-head query GEMM and the attention kernel — the tail of the
# s = 3, 4 distributions was on the logit side: worst of the first 64 seeds 9.7e-4 / 1.05e-3 -> 9.0e-4 / 9.1e-4)
# Round 5: at s = 3, 4 `u` is ONE fp16 value again (its hi | lo residual cost 0.12 ms per B = 256 forward in two HBM-bound kernels and
# bought nothing on the worst seeds): 128 seeds, profiles/r05y_parity_seed_sweep.json — worst 8.4e-4 .. 9.2e-4 (every seed <= 9.3e-4 as
# before), medians 6.2e-4 / 6.7e-4 (bf16 + fp32 output / fp16; +5 %), p90 7.2e-4 .. 8.1e-4.  The in-suite sample is 12 seeds: its median and
# its second-largest value scatter around those, so the s = 3, 4 gates sit one sample-sigma above them; the worst-seed gate is 1e-3 everywhere.
GATES = {2: dict(worst=1.0e-3, p90=8.2e-4, median=7.0e-4), 3: dict(worst=1.0e-3, p90=8.8e-4, median=7.5e-4),
         4: dict(worst=1.0e-3, p90=8.8e-4, median=7.5e-4)}


def test_parity_seed_sweep_gated_on_the_worst_seed():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import parity_sweep
    seeds = int(os.environ.get("TP_PARITY_SEEDS", "12"))
    summary = parity_sweep.sweep(seeds, log=lambda m: print("\n" + m), workers=8)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_seed_sweep_test.json", "w") as f:
        json.dump(summary, f, indent=1)
    for key, r in summary.items():
        g = GATES[int(key[1])]
        assert r["max"] <= g["worst"], (key, r["max"], r["rel_max_per_seed"].index(max(r["rel_max_per_seed"])))
        assert r["p90"] <= g["p90"] and r["median"] <= g["median"], (key, r["p90"], r["median"])


# The worst seed of each recorded 128-seed distribution (profiles/r05y_parity_seed_sweep.json; tools/parity_sweep.py's seed numbering), as
# NAMED regression cases: the 12-seed sample above no longer reaches the tail, these six forwards do.
# A change of schedule moves low bits and with them WHICH seed is worst — re-mint the list from a fresh 128-seed artefact when the
# s = 2 default or the absorbed schedule changes (round 6: the decoupled K launch; profiles/r06*_parity_seed_sweep.json).
WORST_SEEDS = {"s2_bf16_fp32out": [100], "s2_fp16": [41], "s3_bf16_fp32out": [32], "s3_fp16": [62], "s4_bf16_fp32out": [7], "s4_fp16": [75]}


def test_parity_worst_seeds_of_the_recorded_distributions():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import parity_sweep
    summary = parity_sweep.sweep(0, log=lambda m: print("\n" + m), workers=6, seed_lists=WORST_SEEDS)
    assert set(summary) == set(WORST_SEEDS)
    for key, r in summary.items():
        assert r["max"] <= 1.0e-3, (key, r["seed_list"], r["rel_max_per_seed"])


def _inputs(B, dtype, seed=77):
    x, xm = synth.make_inputs(seed, B, dtype)
    return x.cuda(), xm.cuda()


def test_a_module_with_its_own_tuning_context_ignores_the_process_wide_table():
    """The module's context selects the schedule (here: the separate attention kernel instead of attention in the in-projection
    epilogues — different low bits); flipping the process-wide table afterwards, or meanwhile, changes nothing for it; a module
    without a context follows the table as before."""
    dtype, D, s, B = torch.float16, 256, 2, 9
    params = synth.make_params(411, D)
    x, xm = _inputs(B, dtype)
    plain = _module(params, s, D, dtype)
    own = _module(params, s, D, dtype)
    own.tuning = _capi.TuningContext(fuse_attn=1)
    with torch.no_grad():
        y_default = plain((x, xm))
        _capi.set_tuning(_capi.TP_TUNE_FUSE_ATTN, 1)
        try:
            y_table1 = plain((x, xm))
        finally:
            _capi.set_tuning(_capi.TP_TUNE_FUSE_ATTN, 0)
        y_own = own((x, xm))
        assert not torch.equal(y_default, y_table1)            # the knob really changes the result's low bits
        assert torch.equal(y_own, y_table1)                    # the context selected that schedule ...
        _capi.set_tuning(_capi.TP_TUNE_FUSE_ATTN, 2)
        _capi.set_tuning(_capi.TP_TUNE_TRI_STATS, 1)
        try:
            assert torch.equal(own((x, xm)), y_own)            # ... and the table cannot reach it
            assert not torch.equal(plain((x, xm)), y_default)
        finally:
            _capi.set_tuning(_capi.TP_TUNE_FUSE_ATTN, 0)
            _capi.set_tuning(_capi.TP_TUNE_TRI_STATS, 0)
        assert torch.equal(plain((x, xm)), y_default)
        own.tuning.set(_capi.TP_TUNE_FUSE_ATTN, 0)             # the context is live: the next forward follows it
        assert torch.equal(own((x, xm)), y_default)


def test_concurrent_forwards_with_different_contexts_do_not_see_each_other():
    """Two host threads, one module each, different contexts, own streams (the serving situation: model_worker.py runs `generate` on
    a thread per request) while the main thread keeps flipping the process-wide table: every result equals the thread's serial
    reference bit for bit."""
    dtype, D, s, B = torch.float16, 256, 2, 5
    params = synth.make_params(412, D)
    x, xm = _inputs(B, dtype, 78)
    mods = [_module(params, s, D, dtype) for _ in range(2)]
    mods[0].tuning = _capi.TuningContext(fuse_attn=0)
    mods[1].tuning = _capi.TuningContext(fuse_attn=1, tri_stats=1)
    with torch.no_grad():
        refs = [m((x, xm)).clone() for m in mods]
    assert not torch.equal(refs[0], refs[1])
    errors, stop = [], threading.Event()

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.no_grad(), torch.cuda.stream(st):
                for _ in range(40):
                    y = mods[i]((x, xm))
                    st.synchronize()
                    if not torch.equal(y, refs[i]):
                        errors.append(f"thread {i}: result changed")
                        return
        except Exception as exc:     # noqa
            errors.append(f"thread {i}: {exc!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    flips = 0
    while any(t.is_alive() for t in threads):
        _capi.set_tuning(_capi.TP_TUNE_FUSE_ATTN, flips % 3)
        _capi.set_tuning(_capi.TP_TUNE_TRI_STATS, flips % 2)
        flips += 1
    for t in threads:
        t.join()
    for k, v in _capi._TUNING_DEFAULTS.items():
        _capi.set_tuning(k, v)
    assert not errors, errors
    assert flips > 10
