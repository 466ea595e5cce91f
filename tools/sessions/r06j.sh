#!/bin/bash
# round 6, session j: the backward's fp16 gradient chain with a dynamic scale (TP_TUNE_BWD_CHAIN 0 = new default, 1 = bf16 chain of rounds 1-5)
TAG=${TAG:-r06j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_wgrad.py tests/test_gpu_parts.py -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_bwd.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" $OUT/pytest_bwd.log | tail -3; grep -E "^\[grad\]" $OUT/pytest_bwd.log | tail -40; grep -E "^(FAILED|ERROR)|Error" $OUT/pytest_bwd.log | head -20
for k in 0 1 0 1; do
  timeout 600 python tools/train_bench.py --tune BWD_CHAIN=$k --out $OUT/train_bench_chain$k.json 2>&1 | grep "^{" | cut -c1-400
done
