mkdir -p gpurun_out/r03m; O=gpurun_out/r03m
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider -k "point_queries or side_stream" > $O/tests.log 2>&1; echo "tests exit $?"; tail -3 $O/tests.log
for rep in 1 2; do for q in 1 2; do
  timeout 300 python bench.py --tune Q_SIDE_STREAM=$q --no-cpu-baseline --no-extras > $O/bench_side${q}_$rep.json 2>> $O/bench.err; python - $O/bench_side${q}_$rep.json $q <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); lr=d.get("timing",{}).get("long_run",{}); print("B=256 s=2 Q_SIDE_STREAM=%s: %.4f ms  (long-run median %s)" % (sys.argv[2], d["ms_per_step"], lr.get("ms_per_step_median")))
PY
done; done
for q in 1 2; do
  timeout 300 python bench.py --scale-factor 3 --tune Q_SIDE_STREAM=$q --no-cpu-baseline --no-extras > $O/bench_s3_side$q.json 2>> $O/bench.err; python - $O/bench_s3_side$q.json $q <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("B=256 s=3 Q_SIDE_STREAM=%s: %.4f ms" % (sys.argv[2], d["ms_per_step"]))
PY
  for b in 1 32; do timeout 300 python bench.py --batch $b --tune Q_SIDE_STREAM=$q --no-cpu-baseline --no-extras --steps 100 --warmup 20 > $O/bench_b${b}_side$q.json 2>> $O/bench.err; python - $O/bench_b${b}_side$q.json $q $b <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("B=%s Q_SIDE_STREAM=%s: %.4f ms" % (sys.argv[3], sys.argv[2], d["ms_per_step"]))
PY
  done
done
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/rocprof -o lite -- python $R/bench.py --tune Q_SIDE_STREAM=2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$O/rocprof.log 2>&1 ); echo "rocprof exit $?"
tail -3 $O/bench.err
