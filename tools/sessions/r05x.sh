#!/bin/bash
# round-5 last session: HEAD once more exactly as the driver runs it (pytest -m gpu, smoke, bench.py), plus the kernel stats of the 8-GPU shard (B = 32)
OUT=gpurun_out/r05x; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
T0=$(date +%s); timeout 1200 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? wall $(( $(date +%s) - T0 )) s"; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05x/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"]); print({k:v["ms_per_step"] for k,v in d["sweep"].items()})
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof_b32 -o bench -- python $R/bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$OUT/rocprof_b32.log 2>&1 ); F=$(find $OUT/rocprof_b32 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/bench_b32_kernel_stats.csv && head -14 "$F" | cut -c1-170
find $OUT -name "*kernel_trace.csv" -delete
