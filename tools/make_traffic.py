#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py): HBM bytes per launch of the dominant kernel
(kv_layer0: the strided-A bf16 -> fp16 persistent ping-pong GEMM), stamped with the tag, the git HEAD and the digest of
the kernel sources it was measured on — bench.py refuses the file when that digest is not the current tree's.

    python tools/make_traffic.py gpurun_out/<tag>/pmc_summary.json <tag> [B] [dtype]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    summary, tag = sys.argv[1], sys.argv[2]
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    import bench
    d = json.load(open(summary))
    elem = "DF16b" if dtype == "bf16" else "DF16_"
    # gemm8_kernel<T, f16, AMODE 1 (strided A), TRAIN_EPI false, HALF false, XMODE 0, PROBE 0>
    keys = [k for k in d if "gemm8_kernel" in k and f"I{elem}DF16_Li1ELb0ELb0ELi0ELi0E" in k]
    if len(keys) != 1:
        sys.exit(f"kv_layer0 kernel not found (or ambiguous) in {summary}: {keys}")
    r = d[keys[0]]
    rd, wr = r["hbm_read_bytes_corrected"], r["hbm_write_bytes_uncalibrated"]
    alg = B * 576 * 4096 * 2 + 2048 * 4096 * 2 + B * 576 * 2048 * 2
    try:
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        head = "unknown"
    out = {"_stamp": {"tag": tag, "head": head, "kernel_source_sha16": bench.kernel_source_digest()},
           f"kv_layer0_B{B}_{dtype}": {
               "hbm_read_bytes": rd, "hbm_write_bytes": wr, "total": rd + wr, "algorithmic_bytes": alg,
               "mfma_busy_frac": r.get("mfma_busy_frac"), "shader_clock_ghz": r.get("shader_clock_ghz"),
               "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE "
                         f"KiB x2 per MI355X_MICROARCH.md, WRITE_SIZE KiB uncalibrated)",
               "note": "FETCH_SIZE counts L2->fabric requests incl. Infinity-Cache hits: the 16.8 MB weight panel is re-streamed "
                       "from MALL once per 4 row-panels per XCD (2.4 GB), x_multi itself is read once (1.21 GB)"}}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, out["_stamp"], out[f"kv_layer0_B{B}_{dtype}"]["total"])


if __name__ == "__main__":
    main()
