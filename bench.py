#!/usr/bin/env python3
"""bench.py — TokenPacker projector throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]          # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (``TokenPacker.forward``) over one synthetic batch of CLIP
features already resident in HBM: per GPU ``[256, 576, 1024]`` + ``[256, 576, 4096]`` bf16 ->
``[256, 144, 4096]`` (BASELINE.json configs[1]: scale_factor=2, B=256, CLIP-L 336 px, bf16).  With
N > 1 the batch is sharded (weak scaling: 256 images per GPU, weights replicated) and every step
ends with the ONE all-gather of projected tokens the north_star prescribes, so each rank holds
``[256*N, 144, 4096]``; ``--no-gather`` drops it (DDP-style, the LLM consumes the local shard).

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  roofline     — dominant kernel = the first K/V layer GEMM (x_multi·[Wk0;Wv0]^T + GELU, 45 % of the
                 path's FLOPs): algorithmic FLOPs per launch / its average duration measured with HIP
                 events recorded by the library on the launch stream inside the timed forward
                 (tp_forward_staged), against the dense bf16 MFMA peak.
  cpu_baseline — the CPU oracle (a torch-CPU port of the reference's arithmetic, fp32, all host
                 cores) timed on a bounded sample (BASELINE config 1: B=4) on rank 0 at N=1.
  stages_ms    — per-kernel breakdown of one forward.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per image, D = hidden size, s = scale factor
def flops_per_image(s: int, D: int, g: int = 24) -> float:
    N, M, C, Cm, E, H, d = g * g, (g // s) ** 2, 1024, 4096, 1024, 8, 128
    return 2.0 * (N * Cm * E * 2 + N * E * E * 4 + M * C * E + M * E * E * 2 + M * E * D + M * D * D) \
        + 4.0 * M * H * d * s * s


def bytes_per_image(s: int, D: int, g: int = 24) -> float:
    N, M = g * g, (g // s) ** 2
    return (N * 1024 + N * 4096 + M * D) * 2.0


MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--scale-factor", type=int, default=2)
    ap.add_argument("--hidden-size", type=int, default=4096)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--layout", default="tower", choices=["tower", "contiguous"],
                    help="tower = non-contiguous [:,1:] slices as the CLIP tower hands them over")
    ap.add_argument("--no-gather", action="store_true", help="skip the all-gather of projected tokens (N>1)")
    ap.add_argument("--overlap-chunks", type=int, default=1)
    ap.add_argument("--sync-gather", action="store_true",
                    help="N>1: finish each step's all-gather before the next forward (default: the gather of step i "
                         "overlaps the forward of step i+1, two rotating output buffers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--tile", type=int, default=0, help="force GEMM tile (0 auto, 128, 256)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1 (nccl = RCCL; gloo only to exercise the N>1 flow on one GPU)")
    ap.add_argument("--single-device", action="store_true",
                    help="test aid: every rank uses cuda:0 (with --backend gloo)")
    return ap.parse_args()


def make_device_inputs(B, dtype, layout, device, seed):
    """Synthetic unit-normal CLIP features generated on the device (no PCIe in the timed region)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rows = 577 if layout == "tower" else 576
    xb = torch.randn(B, rows, 1024, generator=g, device=device, dtype=torch.float32).to(dtype)
    xmb = torch.randn(B, rows, 4096, generator=g, device=device, dtype=torch.float32).to(dtype)
    if layout == "tower":
        return xb[:, 1:], xmb[:, 1:]
    return xb, xmb


def cpu_baseline(seconds: float, s: int, D: int):
    """Timed CPU leg: the oracle (torch-CPU port of the reference arithmetic) in fp32 on all host
    cores, BASELINE config 1 (B=4).  This is the ONLY place bench.py touches oracle/."""
    from oracle import tokenpacker_oracle as orc          # noqa: the cpu_baseline leg
    from tokenpacker_amd import synth
    B = 4
    params = synth.make_params(0, D)
    x, xm = synth.make_inputs(1234, B)
    cores = torch.get_num_threads()
    with torch.no_grad():
        for _ in range(2):
            orc.forward(params, x, xm, scale_factor=s, compute_dtype=torch.float32)
        n, t0 = 0, time.perf_counter()
        while True:
            orc.forward(params, x, xm, scale_factor=s, compute_dtype=torch.float32)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds or n >= 200:
                break
    return {"value": round(B * n / el, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "ms_per_image": round(1e3 * el / (B * n), 3),
            "sample": f"{n} forwards of B={B}, s={s}, D={D}, fp32 torch-CPU oracle, {el:.1f} s "
                      f"(BASELINE config 1 shape)"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a torch.distributed.run launch (WORLD_SIZE={world})")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU (no CPU fallback for the product path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from tokenpacker_amd import TokenPacker, _capi, shard

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    B, s, D = args.batch, args.scale_factor, args.hidden_size
    M = (24 // s) ** 2
    if args.tile:
        _capi.set_tuning(_capi.TP_TUNE_GEMM_TILE, args.tile)

    torch.manual_seed(0)
    model = TokenPacker(hidden_size=D, scale_factor=s)
    # default init has zero biases / unit LN affine; randomise them so those code paths do real work
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
            elif name.startswith("ln_"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
    model = model.to(device=device, dtype=dtype).eval().requires_grad_(False)

    x, xm = make_device_inputs(B, dtype, args.layout, device, seed=1234 + rank)
    total = B * world
    gather = world > 1 and not args.no_gather

    pipe = shard.TokenGatherPipeline(total, depth=2) if (gather and not args.sync_gather) else None

    def step():
        if pipe is not None:                 # forward of this step overlaps the gather of the previous one
            slot = pipe.submit(model((x, xm)))
            return pipe._bufs[slot]
        if gather:
            return shard.project_sharded(model, x, xm, total, overlap_chunks=args.overlap_chunks)
        return model((x, xm))

    def fence():
        if pipe is not None:
            pipe.drain()                     # every gather issued so far is complete inside the timed region
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            y = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step()
        fence()
        elapsed = time.perf_counter() - t0

        # per-kernel timing inside the real forward: HIP events recorded by the library on the
        # launch stream (tp_forward_staged).  A few extra forwards after the timed region.
        n_prof = min(max(args.steps, 3), 10)
        stage_ms = [0.0] * _capi.TP_NUM_STAGES
        for _ in range(n_prof):
            _, evs = model.forward_staged((x, xm))
            torch.cuda.synchronize(device)
            for i in range(_capi.TP_NUM_STAGES):
                stage_ms[i] += evs[i].elapsed_time(evs[i + 1]) / n_prof

        # N>1 diagnostics, outside the timed region: the same shard without the gather (what DDP training sees) and
        # the gather alone on a fixed shard, so the scaling numbers can be split into compute and xGMI time
        extra = {}
        if gather:
            n_x = min(max(args.steps, 3), 10)
            y_loc = model((x, xm))
            fence()
            t1 = time.perf_counter()
            for _ in range(n_x):
                y_loc = model((x, xm))
            fence()
            extra["forward_only_ms"] = 1e3 * (time.perf_counter() - t1) / n_x
            gp = pipe if pipe is not None else shard.TokenGatherPipeline(total, depth=2)
            gp.submit(y_loc)
            fence()
            t1 = time.perf_counter()
            for _ in range(n_x):
                gp.submit(y_loc)
            gp.drain()
            fence()
            extra["gather_only_ms"] = 1e3 * (time.perf_counter() - t1) / n_x

    assert y.shape == ((total if gather else B), M, D) and torch.isfinite(y[:2].float()).all()

    t = torch.tensor([elapsed, extra.get("forward_only_ms", 0.0), extra.get("gather_only_ms", 0.0)],
                     dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    ms_per_step = 1e3 * elapsed / args.steps
    images_per_s = total * args.steps / elapsed

    if rank == 0:
        fl_img = flops_per_image(s, D)
        kv0_flops = 2.0 * B * 576 * 4096 * 2048                 # algorithmic FLOPs of the dominant launch
        kv0_ms = stage_ms[1]
        achieved = kv0_flops / (kv0_ms * 1e-3) / 1e12
        traffic, traffic_detail = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tfile):
            try:
                traffic_detail = json.load(open(tfile)).get(f"kv_layer0_B{B}_{args.dtype}")
                traffic = traffic_detail["total"] if (traffic_detail and args.layout == "tower") else None
            except Exception:
                traffic = None
        out = {
            "metric": "projector images/sec",
            "value": round(images_per_s, 1),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_image": round(ms_per_step / total, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"TokenPacker projector forward, scale_factor={s} (576->{M} tokens), "
                                   f"B={B} images/GPU, CLIP-L/14 336px grid 24x24, C=1024, Cmulti=4096, D={D}",
                       "global_batch": total, "per_gpu_batch": B, "scale_factor": s, "hidden_size": D,
                       "input_layout": args.layout,
                       "parallelism": f"batch-shard x{world}" + ((" + all_gather(tokens)" + (
                           ", gather of step i overlapped with forward of step i+1" if pipe is not None else "")) if gather else ""),
                       "weights": "random init (reference distribution), synthetic unit-normal CLIP features"},
            "whole_path": {"achieved_tflops": round(fl_img * B / (ms_per_step * 1e-3) / 1e12, 1),
                           "frac_of_mfma_peak": round(fl_img * B / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                           "algorithmic_gflop_per_image": round(fl_img / 1e9, 3),
                           "algorithmic_io_mb_per_image": round(bytes_per_image(s, D) / 1e6, 3),
                           "io_gbps": round(bytes_per_image(s, D) * B / (ms_per_step * 1e-3) / 1e9, 1)},
            "roofline": {"kernel": "tp::gemm8_kernel<T, f16, STRIDED_A> 256x256x64 ping-pong (kv_layer0: x_multi·[Wk0;Wv0]^T + bias + GELU)",
                         "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "flops_per_launch": kv0_flops, "avg_launch_ms": round(kv0_ms, 4),
                         "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/traffic.json)",
                         "algorithmic_bytes": float(B * 576 * 4096 * 2 + 2048 * 4096 * 2 + B * 576 * 2048 * 2)},
            "stages_ms": {n: round(v, 4) for n, v in zip(_capi.STAGE_NAMES, stage_ms)},
        }
        if gather:
            # max over ranks; gather_only moves (N-1)/N of [total, M, D] into every rank per step
            out["multi_gpu"] = {"forward_only_ms": round(float(t[1].item()), 4),
                                "forward_only_images_per_s": round(total / (float(t[1].item()) * 1e-3), 1),
                                "gather_only_ms": round(float(t[2].item()), 4),
                                "gather_bytes_received_per_rank": int((world - 1) * B * M * D * y.element_size())}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, s, D)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
