#!/bin/bash
# round 6, session b: the decoupled K launch (TP_TUNE_DECOUPLE_K) — parity tests, then the placement A/B by batch size
TAG=${TAG:-r06b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_kernels.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_graph.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest.log
for rep in 1 2; do
for b in 32 64 128 256; do
  for k in 0 1 2; do
    timeout 300 python bench.py --batch $b --no-cpu-baseline --no-extras --steps 100 --warmup 20 --min-seconds 0.3 --tune DECOUPLE_K=$k > $OUT/bench_b${b}_k${k}_$rep.json 2>> $OUT/bench.err
    python - "$OUT/bench_b${b}_k${k}_$rep.json" $k <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); lr=d.get("timing",{}).get("long_run",{})
print("B=%d DECOUPLE_K=%s: %.4f ms/step (long-run median %s)  stages %s" % (d["config"]["global_batch"], sys.argv[2], d["ms_per_step"], lr.get("ms_per_step_median"), {k: v for k, v in d["stages_ms"].items() if k in ("kv_layer2_stats","kv_inproj_lnfold","region_attention")}))
PY
  done
done
done
tail -5 $OUT/bench.err
