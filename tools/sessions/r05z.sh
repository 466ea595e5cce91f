#!/bin/bash
# round-5 final session on the frozen sources: the whole GPU suite, the bench line, sweeps, rocprof + PMC (traffic.json), training, e2e, HD
TAG=${TAG:-r05z}
bash tools/gpu_round.sh $TAG smoke tests gemm bench sweep prof pmc small e2e train prof3
OUT=gpurun_out/$TAG; R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
echo "== FETCH_SIZE under the blocked tile order (XCD_SWIZZLE 2): mlp2 / mlp0 read traffic =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/swz2/pmc_FETCH_SIZE -o pmc -- python $R/bench.py --tune XCD_SWIZZLE=2 --steps 3 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$OUT/pmc_swz2.log 2>&1 ); echo "pmc swz2 exit $?"
python tools/pmc_summary.py $OUT/swz2 > $OUT/pmc_summary_swz2.json 2>> $OUT/pmc_summary.err
TAG=$TAG python - <<'PY'
import json
import os
T=os.environ.get("TAG","r05z")
a=json.load(open(f"gpurun_out/{T}/pmc_summary.json")); b=json.load(open(f"gpurun_out/{T}/pmc_summary_swz2.json"))
for k in a:
    if "gemm8_kernel" in k and k in b and isinstance(a[k], dict) and a[k].get("hbm_read_bytes_corrected"):
        print(k[:75], "read GB default %.3f  swizzle2 %.3f" % (a[k]["hbm_read_bytes_corrected"]/1e9, b[k]["hbm_read_bytes_corrected"]/1e9))
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete; du -sh $OUT
echo "== parity, 128 seeds, every configuration, on these sources =="
S=$(date +%s); timeout 900 python tools/parity_sweep.py --seeds 128 --workers 16 --out $OUT/parity_seed_sweep.json 2>&1 | grep parity-sweep; echo "sweep wall $(( $(date +%s)-S )) s"
