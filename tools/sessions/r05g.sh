#!/bin/bash
OUT=gpurun_out/r05g; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tile order for launches with > 8 N-tiles: XCD_SWIZZLE 1 (row-panel-major) vs 2 (4 panels x 8 column groups) =="
for v in 1 2 1 2 1 2; do timeout 300 python bench.py --tune XCD_SWIZZLE=$v --no-cpu-baseline --no-extras --steps 40 --warmup 10 > $OUT/bench_swz$v.json 2>> $OUT/bench.err; python - "$OUT/bench_swz$v.json" "swizzle=$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], "median", d["timing"]["long_run"]["ms_per_step_median"], "kv_layer0", d["stages_ms"]["kv_layer0_gelu"], "mlp0", d["stages_ms"]["mlp0_gelu"], "mlp2", d["stages_ms"]["mlp2"])
PY
done
