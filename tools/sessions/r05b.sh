#!/bin/bash
# round-5 session b: DMA schedule 2 against schedule 1 (and schedule 1 with DEPTH 4), the per-head V GEMM on the pair kernel
OUT=gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp
echo "== targeted tests =="
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pair.py tests/test_gpu_forward.py tests/test_gpu_tri_stats.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_a.log
timeout 600 python -m pytest tests/test_gpu_bench_multi.py -m gpu -x -q -p no:cacheprovider -k "e2e_line_two or self_check or two_rank_bench_line" > $OUT/pytest_b.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_b.log
echo "== lib A/B: schedule 2 (new) vs schedule 1 (old) =="
timeout 300 python tools/lib_ab.py --old tokenpacker_amd/libtokenpacker_s1.so --out $OUT/lib_ab_s2_vs_s1.json 2>&1 | tail -12
echo "== lib A/B: schedule 1 depth 4 (new) vs schedule 1 (old) =="
timeout 300 python tools/lib_ab.py --old tokenpacker_amd/libtokenpacker_s1.so --new tokenpacker_amd/libtokenpacker_s1d4.so --out $OUT/lib_ab_s1d4_vs_s1.json 2>&1 | tail -12
echo "== bench: s1, s2, s1, s2 =="
for v in s1 "" s1 ""; do TP_LIB_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > $OUT/bench_${v:-s2}.json 2>> $OUT/bench.err; python - "$OUT/bench_${v:-s2}.json" "${v:-s2}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], d["roofline"]["frac"], d["timing"]["long_run"] if "timing" in d else "", {k:v for k,v in d["stages_ms"].items()})
PY
done
echo "== s = 3, 4 =="
for sf in 3 4; do for pg in 0 1; do timeout 300 python bench.py --scale-factor $sf --tune PAIR_GEMM=$pg --no-cpu-baseline --no-extras > $OUT/bench_s${sf}_pair$pg.json 2>> $OUT/bench.err; python - "$OUT/bench_s${sf}_pair$pg.json" "s=$sf PAIR_GEMM=$pg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], {k:v for k,v in d["stages_ms"].items()})
PY
done; done
tail -5 $OUT/bench.err
