#!/bin/bash
# usage: tools/prof_v2.sh TAG ARGS... -- kernel-trace profile of tools/time_v2.py ARGS (narlen | ar N)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v2_$TAG -- python $R/tools/time_v2.py "$@" > $R/gpurun_out/prof_v2_$TAG.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_v2_$TAG/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print("total kernel ms", sum(float(r['TotalDurationNs']) for r in rows) / 1e6, "launches", sum(int(r['Calls']) for r in rows))
for r in rows[:22]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.2f} {float(r['Percentage']):5.1f}%")
PY
tail -2 $R/gpurun_out/prof_v2_$TAG.log
