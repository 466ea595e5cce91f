mkdir -p gpurun_out/r03o; O=gpurun_out/r03o
timeout 900 python -m pytest tests/test_gpu_tri_stats.py tests/test_gpu_round3.py -m gpu -q -s -p no:cacheprovider -k "tri or pack_qr or seed_sweep or mean or ragged" > $O/tests.log 2>&1; echo "tests exit $?"; grep -E "^\[pack|^\[parity-sweep|passed|failed" $O/tests.log | head -20
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rocprof_pack -o pack -- python $R/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$O/rocprof_pack.log 2>&1 ); echo "rocprof exit $?"
F=$(find $O/rocprof_pack -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && grep -E "qr_|center_product" "$F" | cut -c1-200
find $O -name "*kernel_trace.csv" -size +5M -delete
echo "== e2e =="; timeout 600 python bench.py --e2e --steps 5 --warmup 2 > $O/bench_e2e.json 2>> $O/bench.err; cut -c1-400 $O/bench_e2e.json
echo "== train =="; timeout 900 python tools/train_bench.py --out $O/train_bench.json > $O/train_bench.log 2>&1; echo "exit $?"; tail -6 $O/train_bench.log
