#!/bin/bash
# round 6, session h: TP_TUNE_DECOUPLE_K at the latency-bound batches (B = 1 .. 10: every launch is a few workgroups, so a K launch
# beside the statistics launch really runs beside it).  0 = round-5 form | 1 = raw logits, side stream | 2 = raw logits, caller stream
TAG=${TAG:-r06h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do
for b in 1 2 4 10; do
  for k in 0 1 2; do
    timeout 300 python bench.py --batch $b --no-cpu-baseline --no-extras --steps 300 --warmup 50 --min-seconds 0.3 --tune DECOUPLE_K=$k > $OUT/bench_b${b}_k${k}_$rep.json 2>> $OUT/bench.err
    python - "$OUT/bench_b${b}_k${k}_$rep.json" $k <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); lr=d.get("timing",{}).get("long_run",{})
print("B=%d DECOUPLE_K=%s: %.4f ms/step (long-run median %s p10 %s)" % (d["config"]["global_batch"], sys.argv[2], d["ms_per_step"], lr.get("ms_per_step_median"), lr.get("ms_per_step_p10")))
PY
  done
done
done
tail -3 $OUT/bench.err
