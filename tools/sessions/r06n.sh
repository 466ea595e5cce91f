#!/bin/bash
# round 6, session n: backward + round-6 tests; rocprof kernel stats of the B = 32 training step (the reference's per-GPU pretraining batch)
TAG=${TAG:-r06n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round6.py -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -vE "^\[grad\] torch.*rel_l2=|^\[parity\]" $OUT/pytest.log | tail -12
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/rocprof_train32 -o train -- python $R/tools/train_bench.py --batches 32 --hip-only --out $R/$OUT/train_bench_prof32.json > $R/$OUT/rocprof_train32.log 2>&1 ); echo "rocprof exit $?"
F=$(find $OUT/rocprof_train32 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -40 "$F" | cut -c1-170
find $OUT -name "*kernel_trace.csv" -size +5M -delete
