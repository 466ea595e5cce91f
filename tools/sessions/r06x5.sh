#!/bin/bash
# round 6, session x5: memory-side counters of the plain K = 4096 GEMM on 256 and on 64 CUs (TP_TUNE_RESERVE_CUS = 24): average L2 read latency seen by
# the CUs' vector caches (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ), average fabric read latency seen by the L2 (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ),
# L2 hit rate, tag stalls, the address unit stalled by the cache
TAG=${TAG:-r06x5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for C in "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL GRBM_GUI_ACTIVE"; do
  # (two more passes — "TCC_REQ TCC_HIT TCC_MISS TCC_TAG_STALL TCC_BUSY" and the TA_*_STALLED_BY_TC counters — did not finish inside 300 s under the profiler)
  i=$((i+1))
  for RSV in 0 24; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_p${i}_r$RSV -o pmc -- python $R/tools/probes/gemm_once.py --reserve $RSV > $R/$OUT/p${i}_r$RSV.log 2>&1 ); echo "pass $i reserve $RSV exit $?"
  done
done
python - <<'PY'
import csv, glob, os, json, collections
out = {}
root = os.environ.get("OUT", "gpurun_out/r06x5")
for d in sorted(glob.glob(os.path.join(root, "pmc_p*_r*"))):
    rsv = d.rsplit("_r", 1)[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if "gemm8_kernel" not in row.get("Kernel_Name", ""): continue
            acc[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
            dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    if not acc: continue
    ids = sorted(acc, key=int)[2:]                       # skip the first two launches
    rec = out.setdefault(f"reserve_{rsv}", {})
    for c in acc[ids[0]]:
        rec[c] = sum(acc[i][c] for i in ids) / len(ids)
    rec["duration_us"] = sum(dur[i] for i in ids) / len(ids) / 1e3
for k, r in out.items():
    if r.get("TCP_TCC_READ_REQ"): r["avg_L2_read_latency_cycles"] = r["TCP_TCC_READ_REQ_LATENCY"] / r["TCP_TCC_READ_REQ"]
    if r.get("TCC_EA0_RDREQ"): r["avg_fabric_read_latency_cycles"] = r["TCC_EA0_RDREQ_LEVEL"] / r["TCC_EA0_RDREQ"]
    if r.get("TCC_REQ"): r["L2_hit_rate"] = r["TCC_HIT"] / r["TCC_REQ"]
json.dump(out, open(os.path.join(root, "memory_side_counters.json"), "w"), indent=1)
for k, r in out.items():
    print(k, {c: (round(v, 4) if v < 1e4 else float(f"{v:.4g}")) for c, v in r.items()})
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
