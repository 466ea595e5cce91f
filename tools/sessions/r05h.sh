#!/bin/bash
OUT=gpurun_out/r05h; mkdir -p $OUT; export TMPDIR=/tmp
echo "== lib A/B: builtin lgkmcnt wait (new) vs inline-asm wait (old = libtokenpacker_base.so) =="
timeout 300 python tools/lib_ab.py --old tokenpacker_amd/libtokenpacker_base.so --out $OUT/lib_ab_wait.json 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "== bench: base, new, base, new =="
for v in base "" base ""; do TP_LIB_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > $OUT/bench_${v:-new}.json 2>> $OUT/bench.err; python - "$OUT/bench_${v:-new}.json" "${v:-new}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], "median", d["timing"]["long_run"]["ms_per_step_median"], d["roofline"]["frac"], {k:v for k,v in d["stages_ms"].items() if v>0.05})
PY
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
