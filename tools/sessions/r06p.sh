#!/bin/bash
# round 6, session p: kernel timeline of the B = 1 and B = 10 forwards (both streams)
TAG=${TAG:-r06p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
for b in 1 10; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_b$b -o t -- python $R/bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$OUT/trace_b$b.log 2>&1 ); echo "trace b=$b exit $?"
python - $b <<'PY'
import csv, glob, sys
b = sys.argv[1]
f = glob.glob(f"gpurun_out/r06p/trace_b{b}/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "point_queries" in r["Kernel_Name"]]
i0, i1 = idx[30], idx[31]
t0 = int(rows[i0]["Start_Timestamp"])
print(f"== B={b}: kernels from one forward's point queries to the next (start us, end us, dur us, queue, grid, name)")
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?')} grid {r.get('Grid_Size','?'):>7} wg {r.get('Workgroup_Size','?'):>4}  {r['Kernel_Name'][:95]}")
PY
done
find $OUT -name "*kernel_trace.csv" -size +5M -delete
