mkdir -p gpurun_out/r03k; O=gpurun_out/r03k
timeout 900 python -m pytest tests/test_gpu_tri_stats.py -m gpu -q -s -p no:cacheprovider > $O/tri_tests.log 2>&1; echo "tri tests exit $?"; grep -E "^\[|passed|failed|Error|error" $O/tri_tests.log | head -40
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_forward.py -m gpu -q -s -p no:cacheprovider > $O/r3_tests.log 2>&1; echo "r3 tests exit $?"; grep -E "^\[parity-sweep|^\[schedules|passed|failed" $O/r3_tests.log | head -40; grep -E "^(FAILED|ERROR)" $O/r3_tests.log | head
for t in 0 1 0 1; do timeout 300 python bench.py --tune TRI_STATS=$t --no-cpu-baseline --no-extras > $O/bench_tri$t.json 2>> $O/bench.err; python - $O/bench_tri$t.json $t <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("TRI_STATS=%s: %.1f img/s %.4f ms  long_run %s frac %.3f" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("timing",{}).get("long_run"), d["roofline"]["frac"]))
PY
done
for sf in 3 4; do for t in 0 1; do timeout 300 python bench.py --scale-factor $sf --tune TRI_STATS=$t --no-cpu-baseline --no-extras > $O/bench_s${sf}_tri$t.json 2>> $O/bench.err; python - $O/bench_s${sf}_tri$t.json $t $sf <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("s=%s TRI_STATS=%s: %.1f img/s %.4f ms" % (sys.argv[3], sys.argv[2], d["value"], d["ms_per_step"]))
PY
done; done
for b in 1 32; do for t in 0 1; do timeout 300 python bench.py --batch $b --tune TRI_STATS=$t --no-cpu-baseline --no-extras --steps 100 --warmup 20 > $O/bench_b${b}_tri$t.json 2>> $O/bench.err; python - $O/bench_b${b}_tri$t.json $t $b <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("B=%s TRI_STATS=%s: %.4f ms" % (sys.argv[3], sys.argv[2], d["ms_per_step"]))
PY
done; done
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rocprof -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$O/rocprof_bench.log 2>&1 ); echo "rocprof exit $?"
F=$(find $O/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -14 "$F" | cut -c1-220
find $O/rocprof -name "*kernel_trace.csv" -size +20M -delete
tail -5 $O/bench.err
