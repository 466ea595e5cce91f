#!/bin/bash
# round 6, session e: (1) the RCCL one-rank tests after the stdout fix; (2) does the absorbed attention kernel run faster per image when
# its qt / u / Hkv working set is smaller (Infinity Cache)?  bench.py --scale-factor 3 at B = 32 .. 256, stage times per image
TAG=${TAG:-r06e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rccl_one_rank.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
for sf in 3 4; do
for b in 32 64 128 256; do
  timeout 300 python bench.py --scale-factor $sf --batch $b --no-cpu-baseline --no-extras --steps 50 --warmup 10 --min-seconds 0 > $OUT/bench_s${sf}_b$b.json 2>> $OUT/bench.err
  python - "$OUT/bench_s${sf}_b$b.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); b=d["config"]["global_batch"]
print("s=%d B=%d: %.4f ms/step  per image us:" % (d["config"]["scale_factor"], b, d["ms_per_step"]), {k: round(v/b*1000,3) for k,v in d["stages_ms"].items()})
PY
done
done
