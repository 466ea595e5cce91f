#!/bin/bash
# round-5 final session on the frozen sources: the whole GPU suite, the bench line, sweeps, rocprof + PMC (traffic.json), training, e2e, HD
bash tools/gpu_round.sh r05z smoke tests gemm bench sweep prof pmc small e2e train prof3
OUT=gpurun_out/r05z; R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
echo "== FETCH_SIZE under the blocked tile order (XCD_SWIZZLE 2): mlp2 / mlp0 read traffic =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/swz2/pmc_FETCH_SIZE -o pmc -- python $R/bench.py --tune XCD_SWIZZLE=2 --steps 3 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$OUT/pmc_swz2.log 2>&1 ); echo "pmc swz2 exit $?"
python tools/pmc_summary.py $OUT/swz2 > $OUT/pmc_summary_swz2.json 2>> $OUT/pmc_summary.err
python - <<'PY'
import json
a=json.load(open("gpurun_out/r05z/pmc_summary.json")); b=json.load(open("gpurun_out/r05z/pmc_summary_swz2.json"))
for k in a:
    if "gemm8_kernel" in k and k in b and isinstance(a[k], dict) and a[k].get("hbm_read_bytes_corrected"):
        print(k[:75], "read GB default %.3f  swizzle2 %.3f" % (a[k]["hbm_read_bytes_corrected"]/1e9, b[k]["hbm_read_bytes_corrected"]/1e9))
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete; du -sh $OUT
