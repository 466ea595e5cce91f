#!/bin/bash
OUT=gpurun_out/r05k; mkdir -p $OUT; export TMPDIR=/tmp
echo "== absorbed attention: 2 waves / SIMD (product) vs 3 (variant aw3: 168 VGPRs + 29 spilled) =="
for sf in 3 4; do for v in "" aw3 "" aw3; do TP_LIB_VARIANT=$v timeout 300 python bench.py --scale-factor $sf --no-cpu-baseline --no-extras > $OUT/bench_s${sf}_${v:-prod}.json 2>> $OUT/bench.err; python - "$OUT/bench_s${sf}_${v:-prod}.json" "s=$sf ${v:-prod}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], "median", d["timing"]["long_run"]["ms_per_step_median"], "attention stage", d["stages_ms"]["region_attention"])
PY
done; done
