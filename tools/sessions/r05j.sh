#!/bin/bash
OUT=gpurun_out/r05j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_tri_stats.py tests/test_gpu_e2e.py tests/test_gpu_pair.py -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "exit $?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "parity-sweep|absorbed\)" $OUT/pytest.log | tail -12
echo "== s = 3, 4 =="
for sf in 3 4; do for pg in 0 1; do timeout 300 python bench.py --scale-factor $sf --tune PAIR_GEMM=$pg --no-cpu-baseline --no-extras > $OUT/bench_s${sf}_pair$pg.json 2>> $OUT/bench.err; python - "$OUT/bench_s${sf}_pair$pg.json" "s=$sf PAIR_GEMM=$pg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], "median", d["timing"]["long_run"]["ms_per_step_median"], {k:v for k,v in d["stages_ms"].items() if v > 0.02})
PY
done; done
echo "== rocprof s=3 =="
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof_s3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --scale-factor 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-seconds 0 > $GRAFT_REPO_ROOT/$OUT/rocprof_s3.log 2>&1 ); F=$(find $OUT/rocprof_s3 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -12 "$F" | cut -c1-200 && cp "$F" $OUT/bench_s3_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT/rocprof_s3 -type f -size +1M -delete
