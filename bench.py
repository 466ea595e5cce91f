#!/usr/bin/env python3
"""bench.py — TokenPacker projector throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]          # N=1; N>1 with no launcher around it: bench.py starts its own
                                                             # N ranks (torch.distributed.run on 127.0.0.1) and rank 0 prints
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (``TokenPacker.forward``) over one synthetic batch of CLIP features already
resident in HBM — BASELINE.json's metric, "projector images/sec, B=256 576->144 tokens, 1/2/4/8 x MI355X": a GLOBAL
batch of 256 images ``[256, 576, 1024]`` + ``[256, 576, 4096]`` bf16 -> ``[256, 144, 4096]`` (configs[1]:
scale_factor=2, CLIP-L 336 px, bf16).  With N > 1 the batch is SHARDED (``--scaling strong``, the default: 256/N
images per GPU, weights replicated — SURVEY.md §8e) and every step ends with the ONE all-gather of projected tokens
the north_star prescribes, so each rank holds ``[256, 144, 4096]``; the gather of step i overlaps the forward of
step i+1 (rotating output buffers, all gathers drained inside the timed region).  ``--gather rccl`` is
``all_gather_into_tensor`` (RCCL kernels over xGMI); ``--gather sdma`` is ``shard.DirectGather``: the shard is written
straight into its rows of the receive buffer and travels as one copy-engine transfer per peer + a sequence flag — no
compute unit, self-tested on the node before it is timed; ``--gather auto`` runs the rccl configuration, PRINTS
its line, and only then runs the sdma configuration in fresh processes (bounded in time), printing its outcome as a second line
``{"sdma_leg": ...}`` — nothing that transport does can cost the run the rccl line or its exit code.  The default is ``rccl``: the
one collective the north_star names, and exactly one line.  ``--scaling weak`` keeps 256 images per GPU instead; ``--no-gather`` drops the gather
(DDP-style: the LLM consumes the local shard).

Other workloads (each prints the same one-line JSON):
  --hd    BASELINE configs[3], TokenPacker-HD: 32 images x 9 crops = 288 crops sharded over the ranks (ragged when
          288 % N != 0), ONE all-gather of b_max-row slots, then the HD token assembly (tp_hd_assemble) reading the
          gathered buffer in place.
  --e2e   BASELINE configs[4], encode_images() end to end (N > 1: `python bench.py --e2e --gpus N` starts its own ranks like the
          projector bench; the line also carries `reference_leg` / `vs_reference`: the same run with the reference's projector): random-init CLIP-ViT-L/14-336 forward (HF transformers on
          PyTorch-ROCm — the producer, not our code) -> HIP projector on the four hidden-state slices (no torch.cat)
          -> Vicuna-7B-shaped prefill (32 Llama layers of GEMMs + SDPA on PyTorch-ROCm — the consumer, not our code),
          B=64 split over the ranks DDP-style; tokens/s plus the split into tower / projector / prefill time.

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  roofline     — dominant kernel = the first K/V layer GEMM (x_multi·[Wk0;Wv0]^T + GELU, 45 % of the path's FLOPs):
                 algorithmic FLOPs per launch / its average duration measured with HIP events recorded by the library
                 on the launch stream inside a real forward (tp_forward_staged), against the dense bf16 MFMA peak;
                 ``traffic`` = HBM bytes per launch from the rocprofv3 PMC passes condensed in profiles/traffic.json
                 (stamped with the git tree of the kernel sources they were measured on; a stale file is refused).
                 ``at_sustained_clock`` (information BESIDE ``frac``, which stays achieved / 2.5 PFLOP/s): the same rate against
                 the MFMA peak at the shader clock the power limit held over this launch in that PMC session (~1.8 of 2.4 GHz).
  cpu_baseline — the reference's op sequence (nn.Linear / F.interpolate / nn.MultiheadAttention ..., fp32) on the host
                 cores, BASELINE config 1 (B=4), a bounded ~12 s sample, rank 0 at N=1 only.
  eager_rocm_baseline — the same op sequence under PyTorch-ROCm eager on THIS GPU, same inputs / dtype / batch: the
                 denominator of the north_star's ">= 5x" (``hip_over_eager``).  ``vs_baseline`` stays null: BASELINE.md
                 publishes no number for the metric.  (Both baseline legs run AFTER the timed region; they are the only
                 code that reaches into oracle/, through one import site.)
  sweep        — N=1: scale_factor 3 and 4 at the same batch; the 2 / 4 / 8-GPU shards (B/2, B/4, B/8), B=10 and B=1 at this scale factor.
  timing       — ``long_run``: the same step loop continued in fenced blocks until >= --min-seconds have been measured
                 (mean / p10 / median / p90 ms per step); the K-step region stays the metric.
  multi_gpu    — N>1: ranks and backend as torch.distributed reports them, every rank's device (index, uuid, PCI id,
                 pid), the NCCL_* / HSA_* variables that are set, the gather mode, forward-only / gather-only times,
                 and with --probe-other-gather the transport not selected.
  stages_ms    — per-kernel breakdown of one forward.
  clocks       — the device's current sclk / mclk / fclk levels (sysfs pp_dpm_*) and average socket power right before and right after
                 the timed region, so that a slow run can be tied to an actual clock / power state.
  pack_ms      — N=1: the one-time weight packing the timed region excludes, inference image and training image.
  memory_side  — mlp0_gelu / kv_layer2_stats: a launch that stores 128 KiB per 23-us tile over one that stores nothing; tells a
                 run in the node's slow memory-side power state (mid-sized batches, profiles/r03u_mid_batch_anomaly.txt) from a
                 normal one.
"""
from __future__ import annotations

import argparse
import hashlib
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
    ap.add_argument("--batch", type=int, default=256,
                    help="images: the GLOBAL batch under --scaling strong (default), per GPU under --scaling weak")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
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
    ap.add_argument("--hd", action="store_true", help="TokenPacker-HD workload (BASELINE configs[3]); see module docstring")
    ap.add_argument("--hd-images", type=int, default=32)
    ap.add_argument("--e2e", action="store_true", help="encode_images() + 7B-shaped prefill (BASELINE configs[4])")
    ap.add_argument("--e2e-batch", type=int, default=64, help="global batch of the --e2e workload")
    ap.add_argument("--e2e-text-tokens", type=int, default=64)
    ap.add_argument("--e2e-layers", type=int, default=32, help="Llama layers of the prefill (32 = 7B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: min(cores, 32))")
    ap.add_argument("--tile", type=int, default=0, help="force GEMM tile (0 auto, 128, 256)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="tp_set_tuning, e.g. --tune ABSORB_KV=2 --tune RESERVE_CUS=1 (keys: _capi.TP_TUNE_*)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1 (nccl = RCCL; gloo only to exercise the N>1 flow on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N=1: initialise torch.distributed (world size 1, --backend) and send the step through the same collective "
                         "code as N>1 (all-gather pipeline, barrier, all-reduce of the clock) — the N=1 line of a scaling run and the "
                         "plain N=1 line can then be compared through the collective path")
    ap.add_argument("--single-device", action="store_true",
                    help="test aid: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "sdma", "auto"],
                    help="N>1: how the projected tokens travel — rccl (default): ONE all_gather_into_tensor per step (RCCL kernels "
                         "over xGMI), what the north_star names | sdma: shard.DirectGather, one hipMemcpyAsync per peer on the copy "
                         "engines, no compute unit (self-tested on the node first: every rank checks every peer's rows) | auto: the "
                         "rccl run first and its line PRINTED; only then the sdma leg, in fresh processes, so that nothing it does "
                         "(a device fault, a hang: it is bounded by --sdma-leg-timeout) can take the printed result with it; its "
                         "outcome follows as a second line {\"sdma_leg\": ...}")
    ap.add_argument("--sdma-leg-timeout", type=float, default=240.0, help="--gather auto: seconds the sdma leg's processes may take")
    ap.add_argument("--gather-depth", type=int, default=3, help="--gather sdma: rotating receive buffers")
    ap.add_argument("--probe-other-gather", action="store_true",
                    help="N>1: after the timed region also time the transport NOT selected by --gather (multi_gpu.other_gather)")
    ap.add_argument("--min-seconds", type=float, default=0.6,
                    help="after the K timed steps, keep stepping until this much time has been measured and report the "
                         "distribution over blocks of steps (timing.long_run); 0: off")
    ap.add_argument("--no-extras", action="store_true",
                    help="N=1: skip the eager PyTorch-ROCm baseline and the s=3 / s=4 / B=32 sweep that follow the timed region")
    return ap.parse_args()


def self_spawn(args) -> None:
    """``python bench.py --gpus N`` with no launcher around it: start the N ranks ourselves (torch.distributed.run on
    127.0.0.1, one process per GPU) and hand their output through — rank 0 prints the one JSON line."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL and tp_gather_* both need it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def make_device_inputs(B, dtype, layout, device, seed):
    """Synthetic unit-normal CLIP features generated on the device (no PCIe in the timed region)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rows = 577 if layout == "tower" else 576
    xb = torch.randn(B, rows, 1024, generator=g, device=device, dtype=torch.float32).to(dtype)
    xmb = torch.randn(B, rows, 4096, generator=g, device=device, dtype=torch.float32).to(dtype)
    if layout == "tower":
        return xb[:, 1:], xmb[:, 1:]
    return xb, xmb


def _reference_op_sequence():
    """The ONE place bench.py touches oracle/: the reference's own torch op sequence (oracle/reference_ops.py, a restatement
    of builder.py:107-137 checked against the reference module), used as the BASELINE that is timed next to the product —
    on the host cores (cpu_baseline) and under PyTorch-ROCm eager on the GPU (eager_rocm_baseline).  Never in the timed
    region, never part of what is shipped."""
    from oracle.reference_ops import eager_forward        # noqa: baseline legs only
    return eager_forward


def cpu_baseline(seconds: float, s: int, D: int, threads: int):
    """Timed CPU leg: the reference's own op sequence — nn.Linear x9, nn.GELU, nn.LayerNorm, F.interpolate and
    nn.MultiheadAttention (L=1, S=s*s) in the reference's token-major layout (oracle/reference_ops.py, a restatement of
    builder.py:107-137 checked against the reference module) — in fp32 on the host cores, BASELINE config 1 (B=4).
    (bench.py touches oracle/ only through _reference_op_sequence().)"""
    eager_forward = _reference_op_sequence()
    from tokenpacker_amd import TokenPacker, synth
    B = 4
    cores = threads if threads > 0 else min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    m = TokenPacker(hidden_size=D, scale_factor=s)
    m.load_state_dict(synth.make_params(0, D))
    m = m.eval().requires_grad_(False)
    x, xm = synth.make_inputs(1234, B)
    with torch.no_grad():
        for _ in range(2):
            eager_forward(m, x, xm)
        n, t0 = 0, time.perf_counter()
        while True:
            eager_forward(m, x, xm)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds or n >= 400:
                break
    return {"value": round(B * n / el, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "what": "oracle/reference_ops.py: the reference's own torch op sequence restated (pinned on the reference, tests/golden/)",
            "ms_per_image": round(1e3 * el / (B * n), 3),
            "sample": f"{n} forwards of B={B}, s={s}, D={D}, fp32, the reference's torch op sequence incl. "
                      f"nn.MultiheadAttention on {cores} host threads of {os.cpu_count()} logical cores, {el:.1f} s "
                      f"(BASELINE config 1 shape)"}


def read_clocks(device) -> dict:
    """Current sclk / mclk / fclk (the level sysfs marks with '*') and average power of the GPU behind `device`, from the amdgpu
    sysfs files; {} where the box exposes none (never an error: this is a diagnostic)."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(device)
        bus = getattr(pr, "pci_bus_id", None)
        if isinstance(bus, int):             # torch reports the three numbers; sysfs paths end in "dddd:bb:dd.f"
            bus = f"{getattr(pr, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(pr, 'pci_device_id', 0):02x}.0"
        bus = (bus or "").lower()
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        pick = [c for c in cards if bus and os.path.realpath(c).lower().endswith(bus)] or \
               [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
        if not pick:
            return {}
        idx = device.index if (not bus and device.index is not None and device.index < len(pick)) else 0
        dev, out = pick[idx], {}
        for name in ("sclk", "mclk", "fclk", "socclk"):
            f = os.path.join(dev, "pp_dpm_" + name)
            if os.path.exists(f):
                cur = [l.split(":")[1].strip().rstrip("*").strip() for l in open(f).read().splitlines() if l.strip().endswith("*")]
                if cur:
                    out[name] = cur[0]
        for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_input")):
            out["power_w"] = round(int(open(f).read().strip()) / 1e6, 1)
            break
        return out
    except Exception as exc:             # noqa
        return {"error": repr(exc)[:120]}


def _time_forward(fn, device, warm: int, iters: int) -> float:
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(device)
    return 1e3 * (time.perf_counter() - t0) / iters


def gpu_extras(args, model, x, xm, dtype, device, images_per_s):
    """Rank 0 at N=1, AFTER the timed region, driver-verifiable companions of the headline:
      eager_rocm_baseline — the reference's own op sequence (oracle/reference_ops.eager_forward: nn.Linear / nn.GELU /
          nn.LayerNorm / F.interpolate / nn.MultiheadAttention, builder.py:107-137) run by PyTorch-ROCm eager on THIS GPU on
          the same inputs, dtype and batch — the denominator of the north_star's ">= 5x".  A baseline leg like cpu_baseline:
          it is timed, never shipped.  (``vs_baseline`` stays null: BASELINE.md publishes no number for this metric.)
      sweep — the other BASELINE configs on one GPU: scale_factor 3 and 4 at the same batch, and the 8-GPU shard (B/8)."""
    eager_forward = _reference_op_sequence()
    out = {}
    B = x.shape[0]
    with torch.no_grad():
        sweep = {}
        # the 2 / 4 / 8-GPU shards of this batch (what a strong-scaling run gives each rank), a typical HD crop count, one image
        for b2 in sorted({max(B // 2, 1), max(B // 4, 1), max(B // 8, 1), 10, 1}, reverse=True):
            if b2 < B:
                ms = _time_forward(lambda: model((x[:b2], xm[:b2])), device, 20, 100)
                sweep[f"s{args.scale_factor}_B{b2}"] = {"ms_per_step": round(ms, 4), "images_per_s": round(b2 / ms * 1e3, 1)}
        for s2 in (3, 4):
            m2 = build_model(args.hidden_size, s2, dtype, device)
            ms = _time_forward(lambda: m2((x, xm)), device, 5, 30)
            sweep[f"s{s2}_B{B}"] = {"ms_per_step": round(ms, 4), "images_per_s": round(B / ms * 1e3, 1)}
            del m2
        out["sweep"] = sweep
        # Weight packing is excluded from the timed region (one-time per weight set) but REPORTED: the inference
        # image (every folded / pre-multiplied weight + the pack-time QR of the triangular statistics) is rebuilt after every
        # parameter update that an eval forward follows; the training image (TP_DESC_TRAIN_PACK) is rebuilt every training step.
        def pack_ms(train):
            stream_ptr = torch.cuda.current_stream(device).cuda_stream
            ts = []
            for _ in range(4):
                model._packed_key = None                     # forces a re-pack (what a parameter update does)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                model._ensure_packed(dtype, device, stream_ptr, force=train)
                e1.record()
                torch.cuda.synchronize(device)
                ts.append(e0.elapsed_time(e1))
            return round(sorted(ts[1:])[1], 3)
        try:
            out["pack_ms"] = {"inference": pack_ms(False), "train_pack": pack_ms(True),
                              "note": "one-time per weight set, outside the timed region (median of 3 re-packs, HIP events)"}
            model._packed_key = None                         # (leave an inference image behind for what follows)
            model((x[:1], xm[:1]))
        except Exception as exc:         # noqa
            out["pack_ms"] = {"error": repr(exc)[:200]}
        try:
            ms = _time_forward(lambda: eager_forward(model, x, xm), device, 10, 30)
            out["eager_rocm_baseline"] = {"ms_per_step": round(ms, 3), "value": round(B / ms * 1e3, 1), "unit": "images/s",
                                          "hip_over_eager": round(images_per_s / (B / ms * 1e3), 3),
                                          "what": f"the reference's torch op sequence (builder.py:107-137 incl. nn.MultiheadAttention) under "
                                                  f"PyTorch-ROCm {torch.__version__} eager, {args.dtype}, B={B}, same inputs and weights, "
                                                  f"10 warm-up + 30 timed forwards"}
        except Exception as exc:         # noqa: an OOM of the eager path must not cost the line
            out["eager_rocm_baseline"] = {"error": repr(exc)[:200]}
    return out


def kernel_source_digest() -> str:
    """sha256 over the kernel sources: what profiles/traffic.json must have been measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tokenpacker_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(B: int, dtype: str, layout: str):
    """PMC-derived HBM bytes per kv_layer0 launch — only if the file was stamped on THIS kernel source."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile) or layout != "tower":
        return None, None
    try:
        doc = json.load(open(tfile))
        entry = doc.get(f"kv_layer0_B{B}_{dtype}")
        stamp = doc.get("_stamp", {})
        if not entry:
            return None, None
        if stamp.get("kernel_source_sha16") != kernel_source_digest():
            return None, f"profiles/traffic.json is stale (measured on kernel sources {stamp.get('kernel_source_sha16')}, tag {stamp.get('tag')})"
        return entry["total"], f"{stamp.get('tag')} @ {stamp.get('head')}"
    except Exception as exc:        # noqa: a broken file must not break the bench line
        return None, f"profiles/traffic.json unreadable: {exc}"


def load_sustained_clock(B: int, dtype: str, layout: str):
    """The shader clock the chip sustained over the dominant launch in the PMC session traffic.json was stamped in (GHz), or None.
    The MFMA peak is priced at the 2.4 GHz boost clock; under this kernel the power limit holds the chip at ~1.8 GHz, and the loop's
    time per K-tile follows the number of active CUs (DESIGN.md section 5.9 b).  Reported BESIDE `frac`, never instead of it."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        entry = doc.get(f"kv_layer0_B{B}_{dtype}")
        if layout != "tower" or not entry or doc.get("_stamp", {}).get("kernel_source_sha16") != kernel_source_digest():
            return None
        return float(entry["shader_clock_ghz"])
    except Exception:               # noqa
        return None


def build_model(D, s, dtype, device):
    from tokenpacker_amd import TokenPacker
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
    return model.to(device=device, dtype=dtype).eval().requires_grad_(False)


# ---------------------------------------------------------------------------------------------------------------------
def run_e2e(args, world, rank, device, dtype, dist):
    """BASELINE configs[4]: tower -> HIP projector -> 7B-shaped prefill, DDP-style (no collective: each rank's LLM
    consumes its own shard)."""
    import torch.nn.functional as F
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from tokenpacker_amd import shard, tower
    s, D = args.scale_factor, args.hidden_size
    M = (24 // s) ** 2
    lo, hi = shard.shard_bounds(args.e2e_batch, world, rank)
    b = hi - lo
    torch.manual_seed(7)
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                            num_attention_heads=16, image_size=336, patch_size=14))
    clip = clip.to(device=device, dtype=dtype).eval().requires_grad_(False)
    model = build_model(D, s, dtype, device)
    L, H, Dh, F_ = args.e2e_layers, 32, D // 32, 11008 if D == 4096 else int(D * 2.6875)
    g = torch.Generator(device=device).manual_seed(11 + rank)

    def w(*shape):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * shape[-1] ** -0.5).to(dtype)
    layers = [dict(qkv=w(3 * D, D), o=w(D, D), gate_up=w(2 * F_, D), down=w(D, F_),
                   n1=torch.ones(D, device=device, dtype=dtype), n2=torch.ones(D, device=device, dtype=dtype)) for _ in range(L)]
    images = torch.randn(b, 3, 336, 336, generator=g, device=device, dtype=torch.float32).to(dtype)
    text = (0.02 * torch.randn(b, args.e2e_text_tokens, D, generator=g, device=device, dtype=torch.float32)).to(dtype)
    T = M + args.e2e_text_tokens

    def rms(t, wgt):
        return (t.float() * torch.rsqrt(t.float().square().mean(-1, keepdim=True) + 1e-6)).to(t.dtype) * wgt

    def prefill(h):
        for ly in layers:
            q, k, v = F.linear(rms(h, ly["n1"]), ly["qkv"]).view(b, T, 3, H, Dh).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(b, T, D)
            h = h + F.linear(a, ly["o"])
            gu = F.linear(rms(h, ly["n2"]), ly["gate_up"])
            h = h + F.linear(F.silu(gu[..., :F_]) * gu[..., F_:], ly["down"])
        return h

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    use_reference = {"on": False}          # the reference leg (after the timed region): the reference's own projector + feature_select
    eager_forward = _reference_op_sequence() if not args.no_extras else None

    def step(timed=False):
        if timed:
            ev[0].record()
        hs = clip(images, output_hidden_states=True).hidden_states          # the tower (clip_encoder.py:46-62)
        if timed:
            ev[1].record()
        if use_reference["on"]:
            # what the reference does here: feature_select concatenates hidden states 12, 16, 22, 23 (clip_encoder.py:28-44),
            # then its projector's torch op sequence (builder.py:107-137) under PyTorch-ROCm eager
            x_ref, xm_ref = tower.concat_reference(hs)
            tok = eager_forward(model, x_ref, xm_ref)
        else:
            x, parts = tower.select_features(hs)                             # [:, 1:] views, no torch.cat
            tok = model((x, parts))                                          # the hot path (llava_arch.py:97)
        if timed:
            ev[2].record()
        out = prefill(torch.cat([tok, text], dim=1))                         # visual tokens ahead of the text
        if timed:
            ev[3].record()
        return out

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    clocks = {}

    def timed_region():
        """W warm-up steps, then EXACTLY K timed steps between two fences; max over the ranks."""
        for _ in range(max(args.warmup, 1)):
            out = step()
        fence()
        clocks["before"] = read_clocks(device) if rank == 0 else {}
        t_start = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        if rank == 0:                        # while the last steps are still executing: the clocks of the loaded chip
            clocks["during"] = read_clocks(device)
        fence()
        el = torch.tensor([time.perf_counter() - t_start], dtype=torch.float64, device=device)
        clocks["after"] = read_clocks(device) if rank == 0 else {}
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return out, float(el.item())

    with torch.no_grad():
        y, elapsed = timed_region()               # (no collective on this path: every rank prefills its own samples)
        step(timed=True)
        torch.cuda.synchronize(device)
        split = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
        # The reference leg — BASELINE configs[4] asks for tokens/s "vs reference": the SAME tower, prefill, inputs and protocol with
        # the reference's projector (its torch op sequence on PyTorch-ROCm eager, fed by its torch.cat feature_select) in place of
        # the HIP one.  After the timed region, never inside it; skipped with --no-extras.
        ref_elapsed, ref_split, ref_err = 0.0, None, None
        if eager_forward is not None:
            try:
                use_reference["on"] = True
                y_ref, ref_elapsed = timed_region()
                step(timed=True)
                torch.cuda.synchronize(device)
                ref_split = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
                del y_ref
            except Exception as exc:     # noqa: an OOM of the eager leg must not cost the line
                ref_err, ref_elapsed = repr(exc)[:200], 0.0
            finally:
                use_reference["on"] = False
    assert y.shape == (b, T, D) and torch.isfinite(y[:1].float()).all()
    t = torch.tensor([elapsed, ref_elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, ref_elapsed = float(t[0].item()), float(t[1].item())
    if rank != 0:
        return
    tokens = args.e2e_batch * T
    out = {"metric": "encode_images + prefill tokens/sec", "value": round(tokens * args.steps / elapsed, 1), "unit": "tokens/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"BASELINE configs[4]: CLIP-ViT-L/14-336 forward (random init, HF transformers on PyTorch-ROCm) -> "
                                  f"HIP TokenPacker (scale_factor={s}, four hidden-state slices, no torch.cat) -> Llama-7B-shaped "
                                  f"prefill ({L} layers, D={D}, PyTorch-ROCm GEMMs + SDPA), {M} visual + {args.e2e_text_tokens} text "
                                  f"tokens per sample",
                      "global_batch": args.e2e_batch, "per_gpu_batch": b, "parallelism": f"DDP-style batch shard x{world}, no collective",
                      "tokens_per_step": tokens},
           "split_ms": {"tower": round(split[0], 3), "projector_hip": round(split[1], 3), "prefill": round(split[2], 3)},
           "clocks": clocks,
           "note": "tower and prefill are PyTorch-ROCm library code around the path (producer / consumer), timed to place the "
                   "projector inside encode_images(); only projector_hip is this repository's kernels"}
    if ref_elapsed > 0:
        ref_value = tokens * args.steps / ref_elapsed
        out["reference_leg"] = {"value": round(ref_value, 1), "unit": "tokens/s", "ms_per_step": round(1e3 * ref_elapsed / args.steps, 3),
                                "split_ms": {"tower": round(ref_split[0], 3), "feature_select_cat_plus_projector_eager": round(ref_split[1], 3),
                                             "prefill": round(ref_split[2], 3)},
                                "what": "same tower, prefill, inputs, W + K protocol; the projector is the reference's torch op sequence "
                                        f"(builder.py:107-137) under PyTorch-ROCm {torch.__version__} eager behind its torch.cat "
                                        "feature_select (clip_encoder.py:28-44); measured AFTER the timed region"}
        out["vs_reference"] = round(out["value"] / ref_value, 4)
        out["projector_speedup_in_place"] = round(ref_split[1] / split[1], 2) if split[1] > 0 else None
    elif ref_err:
        out["reference_leg"] = {"error": ref_err}
    if world > 1:
        out["multi_gpu"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "per_rank_batches": shard.shard_sizes(args.e2e_batch, world),
                            "collective_self_check": dict(COLLECTIVE_CHECK),
                            "note": "DDP-style: the only collectives of this line are the start-up self-check and the MAX-reduction of the clock"}
    flush_c_stdio()
    print(json.dumps(out), flush=True)


def sdma_self_test(total, M, D, dtype, device, depth, world, rank, dist, shard):
    """Build the copy-engine gather and prove it on this node before anything is timed with it: two steps, every rank fills
    its rows with a value only it would write and checks EVERY rank's rows after the gather.  The verdict is reduced over the
    ranks (MIN), so every rank takes the same branch afterwards; waits are bounded (5 s), set-up errors are raised on every
    rank together (shard.DirectGather).  Returns (gather object or None, reason)."""
    try:
        g = shard.DirectGather(total, (M, D), dtype, device, depth=depth, timeout_ms=5000)
    except Exception as exc:             # noqa: raised on every rank alike
        return None, repr(exc)[:300]
    ok, err = True, None
    try:
        for it in range(2):
            g.begin().fill_(float(rank + 1 + 16 * it))
            buf = g.result(g.submit())
            torch.cuda.synchronize(device)
            for r in range(world):
                lo, hi = shard.shard_bounds(total, world, r)
                if hi > lo and not bool((buf[lo:hi] == float(r + 1 + 16 * it)).all()):
                    ok, err = False, f"rank {rank}: rows of rank {r} wrong after step {it}"
        if int(g.status.item()) != 0:
            ok, err = False, f"rank {rank}: a peer's shard did not arrive within 5 s"
    except Exception as exc:             # noqa
        ok, err = False, f"rank {rank}: {exc!r}"[:300]
    verdict = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    if int(verdict.item()) == 0:
        try:
            g.status.zero_()
            g.close()
        except Exception:                # noqa
            pass
        return None, err or "a peer failed the self-test"
    g.timeout_ms = 30000
    return g, None


def run_sdma_leg(args, world) -> None:
    """--gather auto, AFTER the rccl line has been printed and the process group is gone: the same workload with the
    copy-engine gather in FRESH processes (``python bench.py --gpus N --gather sdma``, its own rendezvous port), bounded by
    --sdma-leg-timeout.  Whatever happens in there — a failed self-test, a device fault, a hang — this process only reads
    the child's exit code; the outcome is printed as a second line and the exit code of the run stays 0."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--gather", "sdma", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--batch", str(args.batch), "--scaling", args.scaling, "--scale-factor", str(args.scale_factor),
           "--hidden-size", str(args.hidden_size), "--dtype", args.dtype, "--layout", args.layout, "--backend", args.backend,
           "--gather-depth", str(args.gather_depth), "--no-cpu-baseline", "--no-extras", "--min-seconds", "0"]
    if args.single_device:
        cmd.append("--single-device")
    if args.sync_gather:
        cmd.append("--sync-gather")
    for kv in args.tune:
        cmd += ["--tune", kv]
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                        "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS")}
    leg = {"command": " ".join(cmd[1:])}
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.sdma_leg_timeout,
                             start_new_session=True)
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode == 0 and lines:
            d = json.loads(lines[-1])
            leg.update(status="ok", value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], multi_gpu=d.get("multi_gpu"))
        else:
            leg.update(status=f"failed (exit code {res.returncode})", stderr_tail=res.stderr[-600:])
    except subprocess.TimeoutExpired:
        leg.update(status=f"timed out after {args.sdma_leg_timeout:.0f} s (its processes were killed)")
    except Exception as exc:             # noqa: nothing in this leg may cost the run its exit code
        leg.update(status=f"not run: {exc!r}"[:300])
    print(json.dumps({"sdma_leg": leg}), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
COLLECTIVE_CHECK = {}


def collective_self_check(dist, world: int, rank: int, device, backend: str, timeout_s: float = 60.0) -> dict:
    """N > 1, BEFORE anything is warmed up or timed: one tiny ``all_gather_into_tensor`` (the very collective of the data path,
    SURVEY.md §8e) whose result every rank checks element for element, then a MIN-reduced verdict — so that a mis-configured
    collective library (RCCL transport selection, IPC mode, a rank on the wrong device) ends the run HERE with a message that says
    so, on every rank together, instead of as a hang or garbage inside the timed region.  Bounded: the wait polls the work handle's
    completion for ``timeout_s`` and aborts the process group's run with a diagnostic instead of blocking forever."""
    t0 = time.perf_counter()
    n = 1024
    src = torch.arange(n, device=device, dtype=torch.float32) + 1000.0 * (rank + 1)
    if os.environ.get("TP_BENCH_INJECT_COLLECTIVE_FAULT") and rank == world - 1:
        src += 1.0                                            # (tests: what a mis-routed / corrupted gather looks like to every rank)
    dst = torch.zeros(world * n, device=device, dtype=torch.float32)
    try:
        work = dist.all_gather_into_tensor(dst, src, async_op=True)
        deadline = time.perf_counter() + timeout_s
        while not work.is_completed():
            if time.perf_counter() > deadline:
                raise TimeoutError(f"all_gather_into_tensor of {n * 4} bytes per rank did not complete within {timeout_s:.0f} s")
            time.sleep(0.002)
        work.wait()
        torch.cuda.synchronize(device)
        want = torch.cat([torch.arange(n, device=device, dtype=torch.float32) + 1000.0 * (r + 1) for r in range(world)])
        ok = bool(torch.equal(dst, want))
        note = None if ok else f"rank {rank}: gathered values differ from what the ranks sent (first bad index {int((dst != want).nonzero()[0])})"
    except Exception as exc:     # noqa: every failure mode gets the same message below
        ok, note = False, f"rank {rank}: {exc!r}"
    if not ok:
        # no second collective on a library that just failed one: say what to look at and leave with a distinct exit code
        env = {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_DEBUG", "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY",
                                              "HSA_ENABLE_SDMA", "MASTER_ADDR", "MASTER_PORT") if os.environ.get(k) is not None}
        sys.stderr.write(f"bench.py: the collective self-check FAILED before the warm-up ({backend}, world {world}): {note}\n"
                         f"  device {device} = {torch.cuda.get_device_name(device)}; env {env}\n"
                         f"  (HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this pool; rerun with NCCL_DEBUG=INFO for the transport RCCL chose; "
                         f"--backend gloo --single-device runs the same code path without RCCL)\n")
        sys.stderr.flush()
        os._exit(17)
    verdict = torch.ones(1, device=device, dtype=torch.float32)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)           # (a second, different collective: reductions work as well)
    torch.cuda.synchronize(device)
    return {"all_gather_into_tensor": "ok", "bytes_per_rank": n * 4, "all_reduce_min": float(verdict.item()),
            "seconds": round(time.perf_counter() - t0, 3), "when": "before the warm-up"}


def flush_c_stdio() -> None:
    """RCCL announces itself with a printf to C stdout ("Librccl path : ..."), which is block-buffered when stdout is a pipe and
    would surface at process exit — BEHIND the one JSON line a driver that reads the last line of stdout is looking for."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                        # noqa: cosmetic
        pass


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            self_spawn(args)                                     # never returns
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU (no CPU fallback for the product path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist                 # (world == 1 with --force-dist: a one-rank group through the same code)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        COLLECTIVE_CHECK.update(collective_self_check(dist, world, rank, device, args.backend))
        flush_c_stdio()                      # (RCCL's "Librccl path : ..." banner: out NOW, on every rank, not behind the JSON line at exit)

    from tokenpacker_amd import _capi, hd, shard

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    s, D = args.scale_factor, args.hidden_size
    M = (24 // s) ** 2
    if args.tile:
        _capi.set_tuning(_capi.TP_TUNE_GEMM_TILE, args.tile)
    for kv in args.tune:
        key, val = kv.split("=")
        _capi.set_tuning(getattr(_capi, "TP_TUNE_" + key.upper()), int(val))

    if args.e2e:
        run_e2e(args, world, rank, device, dtype, dist)
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- the workload: what each rank projects, and how many units the whole job processes per step -------------
    if args.hd:
        # 32 images x (2 x 4 grid + global view) = 288 crops (BASELINE configs[3]); crops are the projector's batch
        hb, wb = [2] * args.hd_images, [4] * args.hd_images
        total = sum(hd.hd_crop_count(a, c) for a, c in zip(hb, wb))
        lo, hi = shard.shard_bounds(total, world, rank)
        B, scaling = hi - lo, "strong"
    elif args.scaling == "strong":
        total = args.batch
        lo, hi = shard.shard_bounds(total, world, rank)
        B, scaling = hi - lo, "strong"
    else:
        B, total, scaling = args.batch, args.batch * world, "weak"
    ragged = len(set(shard.shard_sizes(total, world))) > 1 if scaling == "strong" else False

    model = build_model(D, s, dtype, device)
    x, xm = make_device_inputs(max(B, 1), dtype, args.layout, device, seed=1234 + rank)
    x, xm = x[:B], xm[:B]
    gather = use_dist and not args.no_gather
    # ---- the gather's transport (N > 1): rccl | sdma | auto = both, after the sdma self-test ------------------------
    pipe = dgather = None
    transports, sdma_note = [], None
    if gather and not args.hd:
        if args.gather == "sdma":
            # proven on this node before anything is timed with it (set-up errors and a failed self-test are raised on every rank)
            dgather, sdma_note = sdma_self_test(total, M, D, dtype, device, args.gather_depth, world, rank, dist, shard)
            if dgather is None:
                raise RuntimeError(f"--gather sdma: the copy-engine transport failed its self-test on this node: {sdma_note}")
            if os.environ.get("TP_BENCH_INJECT_SDMA_FAULT") and rank == world - 1:
                os.abort()                   # (tests: a device fault inside the sdma leg must not cost the rccl line)
            transports = ["sdma"]
        else:                                # rccl, and the first leg of auto
            transports = ["rccl"]
        if "rccl" in transports and not args.sync_gather and not ragged:
            pipe = shard.TokenGatherPipeline(total, depth=2)
    cur = {"t": transports[0] if transports else None}      # the transport step() uses
    if args.hd:
        gsep = torch.Generator(device=device).manual_seed(5)
        sep = torch.randn(D, generator=gsep, device=device, dtype=torch.float32).to(dtype)
        ret = torch.randn(D, generator=gsep, device=device, dtype=torch.float32).to(dtype)

    def step():
        if args.hd:
            # project the local crops -> ONE all-gather (b_max-row slots, read in place) -> HD token assembly of all images
            if gather:
                g_tok = shard.project_sharded(model, x, xm, total, dense=False, force_collective=args.force_dist)
                if isinstance(g_tok, shard.GatheredTokens):
                    return hd.assemble_hd_tokens(g_tok.buf, hb, wb, sep, ret, crop_map=g_tok.crop_map())
                return hd.assemble_hd_tokens(g_tok, hb, wb, sep, ret)
            y_loc = model((x, xm))
            return hd.assemble_hd_tokens(y_loc, hb, wb, sep, ret) if world == 1 else y_loc
        if cur["t"] == "sdma":               # the shard is written straight into its rows of the receive buffer, then pushed
            view = dgather.begin()
            if B:
                model((x, xm), _out=view)
            ticket = dgather.submit()
            return dgather.result(ticket) if args.sync_gather else dgather.bufs[ticket[1]]
        if cur["t"] == "rccl" and pipe is not None:       # forward of this step overlaps the gather of the previous one
            slot = pipe.submit(model((x, xm)))
            return pipe._bufs[slot]
        if gather:
            return shard.project_sharded(model, x, xm, total, overlap_chunks=args.overlap_chunks, force_collective=args.force_dist)
        return model((x, xm))

    def fence():
        if pipe is not None:
            pipe.drain()                     # every gather issued so far is complete inside the timed region
        if dgather is not None:
            dgather.drain()
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    clocks = {}

    def timed_region():
        """W warm-up steps, then EXACTLY K timed steps between two fences; max over the ranks."""
        for _ in range(max(args.warmup, 1)):
            out = step()
        fence()
        clocks["before"] = read_clocks(device) if rank == 0 else {}
        t_start = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        if rank == 0:                        # while the last steps are still executing: the clocks of the loaded chip
            clocks["during"] = read_clocks(device)
        fence()
        el = torch.tensor([time.perf_counter() - t_start], dtype=torch.float64, device=device)
        clocks["after"] = read_clocks(device) if rank == 0 else {}
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return out, float(el.item())

    with torch.no_grad():
        per_transport = {}
        y, elapsed = timed_region()
        if transports:
            per_transport[cur["t"]] = elapsed
        chosen = cur["t"]

        # The K steps above are the driver's; at 4 ms per step they are < 0.1 s of clock.  Keep stepping (same loop, same
        # fences) until --min-seconds have been measured, in blocks, so that the line also carries a distribution.
        long_run = None
        if args.min_seconds > 0 and args.steps > 0 and not args.hd:
            per = max(elapsed / args.steps, 1e-6)
            blk = max(1, min(args.steps, int(0.05 / per) + 1))
            n_blk = int(min(200, max(5, args.min_seconds / (per * blk))))
            if use_dist:                    # every rank must run the same number of steps
                nb = torch.tensor([blk, n_blk], dtype=torch.int64, device=device)
                dist.broadcast(nb, src=0)
                blk, n_blk = int(nb[0].item()), int(nb[1].item())
            blocks = []
            for _ in range(n_blk):
                tb = time.perf_counter()
                for _ in range(blk):
                    y = step()
                fence()
                blocks.append((time.perf_counter() - tb) / blk * 1e3)
            tb_all = torch.tensor(blocks, dtype=torch.float64, device=device)
            if use_dist:
                dist.all_reduce(tb_all, op=dist.ReduceOp.MAX)
            bl = sorted(tb_all.tolist())
            long_run = {"blocks": n_blk, "steps_per_block": blk, "seconds": round(sum(bl) * blk * 1e-3, 3),
                        "ms_per_step_mean": round(sum(bl) / len(bl), 4), "ms_per_step_p10": round(bl[len(bl) // 10], 4),
                        "ms_per_step_median": round(bl[len(bl) // 2], 4), "ms_per_step_p90": round(bl[(len(bl) * 9) // 10], 4),
                        "note": "each block is fenced (synchronize [+ barrier]) like the timed region, so short blocks carry "
                                "the fence's own cost; the K-step region above stays the metric"}

        # per-kernel timing inside the real forward: HIP events recorded by the library on the
        # launch stream (tp_forward_staged).  A few extra forwards after the timed region.
        n_prof = min(max(args.steps, 3), 10)
        stage_ms = [0.0] * _capi.TP_NUM_STAGES
        for _ in range(n_prof if B > 0 else 0):
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
            def probe_rccl():
                gp = pipe if pipe is not None else shard.TokenGatherPipeline(total, depth=2)
                gp.submit(y_loc)
                fence()
                t1 = time.perf_counter()
                for _ in range(n_x):
                    gp.submit(y_loc)
                gp.drain()
                fence()
                return 1e3 * (time.perf_counter() - t1) / n_x

            def probe_sdma(g):
                for _ in range(g.depth):                 # the shard sits in every buffer's own rows: submit() moves nothing locally
                    g.submit(y_loc)
                g.drain()
                fence()
                t1 = time.perf_counter()
                for _ in range(n_x):
                    g.submit()
                g.drain()
                fence()
                return 1e3 * (time.perf_counter() - t1) / n_x

            if args.hd or (ragged and chosen != "sdma"):
                fence()
                t1 = time.perf_counter()
                for _ in range(n_x):
                    shard.all_gather_tokens(y_loc, total, dense=False)
                fence()
                extra["gather_only_ms"] = 1e3 * (time.perf_counter() - t1) / n_x
            elif chosen == "sdma":
                extra["gather_only_ms"] = probe_sdma(dgather)
            else:
                extra["gather_only_ms"] = probe_rccl()
            # the transport NOT chosen, on the same shard, on request only (--probe-other-gather: it is set up IN this process)
            if not args.hd and not ragged and args.probe_other_gather:
                if chosen == "sdma":
                    extra["other_gather"] = {"mode": "rccl", "gather_only_ms": probe_rccl()}
                elif chosen == "rccl":
                    g2 = dgather if dgather is not None else shard.DirectGather(total, (M, D), dtype, device, depth=args.gather_depth,
                                                                                 timeout_ms=5000)
                    extra["other_gather"] = {"mode": "sdma", "gather_only_ms": probe_sdma(g2)}
                    if g2 is not dgather:
                        g2.close()

    if args.hd:
        rows_img = hd.hd_token_rows(2, 4, M)
        if gather or world == 1:
            assert len(y) == args.hd_images and y[0].shape == (rows_img, D) and torch.isfinite(y[0].float()).all()
    else:
        assert y.shape == ((total if gather else B), M, D) and torch.isfinite(y[:2].float()).all(), y.shape

    t = torch.tensor([elapsed, extra.get("forward_only_ms", 0.0), extra.get("gather_only_ms", 0.0),
                      extra.get("other_gather", {}).get("gather_only_ms", 0.0),
                      extra.get("other_gather", {}).get("pipelined_ms_per_step", 0.0)], dtype=torch.float64, device=device)
    rank_info = None
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # what the collective library itself saw: ranks, and the device each one drives
        props = torch.cuda.get_device_properties(device)
        mine = {"rank": dist.get_rank(), "local_rank": local_rank, "device_index": device.index, "name": props.name,
                "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)
    elapsed = float(t[0].item())
    ms_per_step = 1e3 * elapsed / args.steps
    images_per_s = total * args.steps / elapsed

    if rank == 0:
        fl_img = flops_per_image(s, D)
        kv0_flops = 2.0 * B * 576 * 4096 * 2048                 # algorithmic FLOPs of the dominant launch ON THIS RANK
        kv0_ms = stage_ms[1]
        achieved = kv0_flops / (kv0_ms * 1e-3) / 1e12 if kv0_ms > 0 else 0.0
        traffic, traffic_src = load_traffic(B, args.dtype, args.layout)
        unit_name = "HD crops (projector batch elements)" if args.hd else "images"
        out = {
            "metric": "projector images/sec",
            "value": round(images_per_s, 1),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_image": round(ms_per_step / total, 6),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": (f"TokenPacker-HD: {args.hd_images} images x 9 crops (2x4 grid + global view) = {total} crops "
                                    f"sharded over {world} rank(s), projector + all-gather + HD token assembly, " if args.hd else
                                    "TokenPacker projector forward, ") +
                                   f"scale_factor={s} (576->{M} tokens), global batch {total} {unit_name}, {B} on rank 0, "
                                   f"CLIP-L/14 336px grid 24x24, C=1024, Cmulti=4096, D={D}",
                       "global_batch": total, "per_gpu_batch": B, "scale_factor": s, "hidden_size": D,
                       "input_layout": args.layout,
                       "parallelism": f"batch-shard x{world}" + ((" + all_gather(tokens)" + (
                           ", gather of step i overlapped with forward of step i+1" if ((chosen == "rccl" and pipe is not None) or (chosen == "sdma" and not args.sync_gather)) else "")) if gather else ""),
                       "weights": "random init (reference distribution), synthetic unit-normal CLIP features",
                       "tuning": {k: _capi.get_tuning(getattr(_capi, k)) for k in dir(_capi) if k.startswith("TP_TUNE_") and k != "TP_TUNE_COUNT"}},
            "whole_path": {"achieved_tflops": round(fl_img * total / (ms_per_step * 1e-3) / 1e12, 1),
                           "frac_of_mfma_peak": round(fl_img * total / (ms_per_step * 1e-3) / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
                           "algorithmic_gflop_per_image": round(fl_img / 1e9, 3),
                           "algorithmic_io_mb_per_image": round(bytes_per_image(s, D) / 1e6, 3),
                           "io_gbps": round(bytes_per_image(s, D) * total / (ms_per_step * 1e-3) / 1e9, 1)},
            "roofline": {"kernel": "tp::gemm8_kernel<T, f16, STRIDED_A> 256x256x64 ping-pong (kv_layer0: x_multi·[Wk0;Wv0]^T + bias + GELU)",
                         "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "flops_per_launch": kv0_flops, "avg_launch_ms": round(kv0_ms, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/traffic.json)",
                         "algorithmic_bytes": float(B * 576 * 4096 * 2 + 2048 * 4096 * 2 + B * 576 * 2048 * 2)},
            "stages_ms": {n: round(v, 4) for n, v in zip(_capi.STAGE_NAMES, stage_ms)},
        }
        # the store-heavy short-K launch over the launch that stores nothing: 0.8 - 0.9 normally, ~1.3 in the slow memory-side
        # power state some boxes put a 32 ... 128-image forward in (profiles/r03u_mid_batch_anomaly.txt)
        out["clocks"] = clocks
        sclk = load_sustained_clock(B, args.dtype, args.layout)
        if sclk:
            # information beside `frac` (which stays achieved / 2.5 PFLOP/s): the same rate against the peak at the clock the power limit allows
            out["roofline"]["at_sustained_clock"] = {"shader_clock_ghz": round(sclk, 3), "boost_clock_ghz": 2.4,
                                                     "peak": round(MFMA_PEAK_TFLOPS * sclk / 2.4, 1),
                                                     "frac": round(achieved / (MFMA_PEAK_TFLOPS * sclk / 2.4), 4),
                                                     "source": "profiles/traffic.json (cycle counters of the PMC session over this launch); DESIGN.md 5.9 (b)"}
        if stage_ms[2] > 0 and s == 2:
            out["memory_side"] = {"mlp0_over_statistics": round(stage_ms[8] / stage_ms[2], 3),
                                  "note": "mlp0_gelu / kv_layer2_stats of this rank's forward; ~0.85 normal, >= 1.2 = the node's slow memory-side power state"}
        if gather:
            # max over ranks; gather_only moves (N-1)/N of [total, M, D] into every rank per step
            out["multi_gpu"] = {"forward_only_ms": round(float(t[1].item()), 4),
                                "forward_only_images_per_s": round(total / (float(t[1].item()) * 1e-3), 1),
                                "gather_only_ms": round(float(t[2].item()), 4),
                                "collective": (f"DirectGather: {world - 1} hipMemcpyAsync (copy engines, no CU) per rank and step + sequence flags, "
                                               f"depth {args.gather_depth}" if chosen == "sdma" else
                                               f"{args.backend} all_gather_into_tensor over {world} ranks") + (" (ragged: b_max-row slots)" if ragged and chosen != "sdma" else ""),
                                "gather": chosen if chosen else "rccl",
                                "gather_requested": args.gather,
                                "transports_timed_ms_per_step": {k: round(1e3 * v / args.steps, 4) for k, v in per_transport.items()},
                                "sdma_self_test": "passed" if (dgather is not None and chosen == "sdma") else None,
                                "sdma_flags_fine_grained": (dgather.flags_fine_grained if dgather is not None else None),
                                "pipelined": bool((chosen == "rccl" and pipe is not None) or (chosen == "sdma" and not args.sync_gather)),
                                "gather_bytes_received_per_rank": int((total - B) * M * D * 2),
                                "ranks": dist.get_world_size(), "backend": dist.get_backend(), "rank_devices": rank_info,
                                "collective_self_check": dict(COLLECTIVE_CHECK),
                                "env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                                        "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY", "HSA_ENABLE_SDMA")
                                        if os.environ.get(k) is not None}}
            if "other_gather" in extra:
                out["multi_gpu"]["other_gather"] = {"mode": extra["other_gather"]["mode"], "gather_only_ms": round(float(t[3].item()), 4)}
                if float(t[4].item()) > 0:
                    out["multi_gpu"]["other_gather"]["pipelined_ms_per_step"] = round(float(t[4].item()), 4)
        if long_run is not None:
            out["timing"] = {"timed_region_s": round(elapsed, 4), "long_run": long_run}
        if world == 1 and not args.no_extras and not args.hd:
            out.update(gpu_extras(args, model, x, xm, dtype, device, images_per_s))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, s, D, args.cpu_threads)
        flush_c_stdio()
        print(json.dumps(out), flush=True)

    if dgather is not None:
        dgather.close()

    if use_dist:
        dist.destroy_process_group()
    if world > 1 and rank == 0 and args.gather == "auto" and gather and not args.hd:
        run_sdma_leg(args, world)


if __name__ == "__main__":
    main()
