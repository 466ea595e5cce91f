#!/bin/bash
OUT=gpurun_out/r05r; mkdir -p $OUT; export TMPDIR=/tmp
for m in 1 2; do
  echo "== TP_EXP_U_MODE=$m: parity, 128 seeds, s = 3, 4 =="
  TP_EXP_U_MODE=$m timeout 600 python tools/parity_sweep.py --seeds 128 --workers 16 --scale-factors 3 4 --out $OUT/parity_umode$m.json 2>&1 | grep parity-sweep
done
echo "== timing =="
for sf in 3 4; do for m in 0 1 2 0 1 2; do TP_EXP_U_MODE=$m timeout 300 python bench.py --scale-factor $sf --no-cpu-baseline --no-extras > $OUT/bench_s${sf}_u$m.json 2>> $OUT/bench.err; python - "$OUT/bench_s${sf}_u$m.json" "s=$sf u_mode=$m" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], "median", d["timing"]["long_run"]["ms_per_step_median"], "attention stage", d["stages_ms"]["region_attention"])
PY
done; done
