#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, micro-bench, rocprof kernel trace.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo ==" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
echo "== smoke ==" ; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
echo "== pytest gpu ==" ; timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -40 $OUT/pytest_gpu.log
echo "== bench ==" ; timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== gemm bench ==" ; timeout 600 python tools/gemm_bench.py > $OUT/gemm_bench.log 2>&1; echo "exit $?"; cat $OUT/gemm_bench.log | tail -40
echo "== rocprof kernel trace ==" 
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1 ); echo "rocprof exit $?"
find $OUT/rocprof -name "*stats*" | head; 
F=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -25 "$F"
# keep the merged-back payload small: drop the raw per-dispatch trace
find $OUT/rocprof -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
