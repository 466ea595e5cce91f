#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter-collection CSVs (one directory per pass, see tools/gpu_round.sh) into
a per-kernel JSON: average counter value per dispatch, plus HBM bytes corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950 (FETCH_SIZE is reported in KiB and counts a
wide coalesced streaming read at exactly half its bytes -> x2; WRITE_SIZE in KiB, uncalibrated).

    python tools/pmc_summary.py gpurun_out/<tag>  > pmc_summary.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def main(root: str) -> None:
    acc = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> [values per dispatch]
    meta = {}
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            per_dispatch = defaultdict(float)
            names = {}
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", "?")) + (f" grid={row.get('Grid_Size')}" if ALL else "")
                key = (row.get("Dispatch_Id"), k, row.get("Counter_Name"))
                per_dispatch[key] += float(row.get("Counter_Value", 0) or 0)   # rows may be split per XCD/instance
                per_dispatch[(row.get("Dispatch_Id"), k, "duration_ns")] = \
                    float(row.get("End_Timestamp", 0)) - float(row.get("Start_Timestamp", 0))
                names[k] = row
            for (disp, k, c), v in per_dispatch.items():
                acc[k][c].append(v)
            for k, row in names.items():
                meta[k] = {x: row.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                  "Workgroup_Size", "Grid_Size")}
    out = {}
    for k, counters in sorted(acc.items()):
        if not (k.startswith("tp::") or k.startswith("_ZN2tp") or "gemm" in k or (ALL and "Cijk" in k)):
            continue
        rec = {"dispatches": max(len(v) for v in counters.values()), **{m: meta[k][m] for m in meta.get(k, {})}}
        for c, vals in counters.items():
            rec[c] = sum(vals) / len(vals)
        if "FETCH_SIZE" in rec:
            rec["hbm_read_bytes_corrected"] = rec["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in rec:
            rec["hbm_write_bytes_uncalibrated"] = rec["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and "GRBM_GUI_ACTIVE" in rec and rec["GRBM_GUI_ACTIVE"]:
            # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (value / duration = 8 x shader clock); MFMA busy
            # cycles are summed over the 256 CUs x 4 SIMDs (16 cycles per v_mfma_f32_16x16x32).
            cyc = rec["GRBM_GUI_ACTIVE"] / 8
            rec["mfma_busy_frac"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
            if rec.get("duration_ns"):
                rec["shader_clock_ghz"] = cyc / rec["duration_ns"]
            if "SQ_LDS_IDX_ACTIVE" in rec:
                rec["lds_busy_frac"] = rec["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        out[k] = rec
    json.dump(out, sys.stdout, indent=1)
    print()


ALL = "--all" in sys.argv

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
