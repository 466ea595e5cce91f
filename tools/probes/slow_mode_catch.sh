# Sample a box: a 32-image forward; when it is in the slow mode (> 0.78 ms), take a kernel trace of it.
mkdir -p gpurun_out/r03x; O=gpurun_out/r03x
timeout 40 python bench.py --batch 32 --no-cpu-baseline --no-extras --steps 100 --warmup 20 --min-seconds 0 > $O/b32.json 2>> $O/err.log
MS=$(python -c "import json;print(json.load(open('$O/b32.json'))['ms_per_step'])"); echo "B=32 $MS ms"
python -c "import json;print(json.load(open('$O/b32.json'))['stages_ms'])"
if python -c "import sys;sys.exit(0 if $MS > 0.78 else 1)"; then
  R=$(pwd); export TMPDIR=/tmp
  ( cd /tmp && timeout 40 rocprofv3 --kernel-trace --output-format csv -d $R/$O/rocprof_b32 -o b32 -- python $R/bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$O/rocprof.log 2>&1 ); echo "slow mode: trace exit $?"
fi
