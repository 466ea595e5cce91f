#!/bin/bash
# round 6, session q: kv_layer0 enqueued in front of the side stream's launches — forward tests, then the small-batch lines and the headline
TAG=${TAG:-r06q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_round3.py tests/test_gpu_graph.py tests/test_gpu_parts.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
for rep in 1 2; do
for b in 1 2 4 10 32 256; do
  timeout 300 python bench.py --batch $b --no-cpu-baseline --no-extras --steps 200 --warmup 30 --min-seconds 0.3 > $OUT/bench_b${b}_$rep.json 2>> $OUT/bench.err
  python - "$OUT/bench_b${b}_$rep.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); lr=d.get("timing",{}).get("long_run",{})
print("B=%d: %.4f ms/step (long-run median %s p10 %s)" % (d["config"]["global_batch"], d["ms_per_step"], lr.get("ms_per_step_median"), lr.get("ms_per_step_p10")))
PY
done
done
