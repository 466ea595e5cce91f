mkdir -p gpurun_out/r03t; O=gpurun_out/r03t
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_forward.py tests/test_gpu_tri_stats.py tests/test_gpu_parts.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; echo "tests exit $?"; tail -3 $O/tests.log
for b in 1 2 3 4 6 8 10; do for w in 9 0; do
  timeout 300 python bench.py --batch $b --tune SMALL_GEMM_WAVES=$w --no-cpu-baseline --no-extras --steps 200 --warmup 30 > $O/bench_b${b}_w$w.json 2>> $O/bench.err; python - $O/bench_b${b}_w$w.json $w $b <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); lr=d.get("timing",{}).get("long_run",{}); print("B=%s SMALL_GEMM_WAVES=%s: %.4f ms  (long-run median %s)" % (sys.argv[3], sys.argv[2], d["ms_per_step"], lr.get("ms_per_step_median")))
PY
done; done
tail -3 $O/bench.err
