mkdir -p gpurun_out/r03l; O=gpurun_out/r03l
timeout 600 python -m pytest tests/test_gpu_tri_stats.py tests/test_gpu_forward.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; echo "tests exit $?"; tail -3 $O/tests.log
for b in 1 4 8 32 256; do for q in 1 0; do
  st=100; [ $b = 256 ] && st=30
  timeout 300 python bench.py --batch $b --tune Q_SIDE_STREAM=$q --no-cpu-baseline --no-extras --steps $st --warmup 20 > $O/bench_b${b}_side$q.json 2>> $O/bench.err; python - $O/bench_b${b}_side$q.json $q $b <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); lr=d.get("timing",{}).get("long_run",{}); print("B=%s Q_SIDE_STREAM=%s: %.4f ms  (long-run median %s)" % (sys.argv[3], sys.argv[2], d["ms_per_step"], lr.get("ms_per_step_median")))
PY
done; done
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/rocprof_b1 -o b1 -- python $R/bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$O/rocprof_b1.log 2>&1 ); echo "rocprof exit $?"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/rocprof_b32 -o b32 -- python $R/bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$O/rocprof_b32.log 2>&1 ); echo "rocprof exit $?"
tail -3 $O/bench.err
