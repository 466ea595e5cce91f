export TMPDIR=/tmp
OUT=gpurun_out/r04f; mkdir -p $OUT
TP_PARITY_SEEDS=8 timeout 1200 python -m pytest tests/test_gpu_bench_multi.py tests/test_gpu_gather_direct.py tests/test_gpu_tri_stats.py tests/test_gpu_round3.py tests/test_gpu_round4.py "tests/test_gpu_pair.py::test_full_size_on_the_pair_kernel_is_race_free" tests/test_gpu_forward.py -m gpu -q -x -p no:cacheprovider -k "not test_every_schedule" > $OUT/pytest_subset.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" $OUT/pytest_subset.log | tail -3; grep -E "^(FAILED|ERROR)|parity-sweep|pack-qr" $OUT/pytest_subset.log | head -30
timeout 900 python tools/parity_sweep.py --seeds 128 --scale-factors 3 4 --tags fp16 --out $OUT/parity_seed_sweep_s34_fp16.json 2>&1 | tail -4
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json | cut -c1-3000; tail -3 $OUT/bench.err
