mkdir -p gpurun_out/r03p; O=gpurun_out/r03p
timeout 900 python -m pytest tests/test_gpu_bench_multi.py tests/test_gpu_tri_stats.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests exit $?"; tail -3 $O/tests.log
echo "== e2e =="; timeout 600 python bench.py --e2e --steps 5 --warmup 2 > $O/bench_e2e.json 2>> $O/bench.err; cut -c1-300 $O/bench_e2e.json; python - $O/bench_e2e.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d["split_ms"])
PY
bash tools/gpu_round.sh r03p bench prof pmc > $O/round.log 2>&1; grep -E "bench exit|rocprof exit|pmc .* exit|wrote" $O/round.log; python - $O/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_source"))
PY
