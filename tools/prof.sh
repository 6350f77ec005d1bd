#!/bin/bash
# usage: tools/prof.sh TAG  -- kernel-trace profile of one bench step; summary -> gpurun_out/prof_TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$1.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$1/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.2f} {float(r['Percentage']):5.1f}%")
PY
grep -o '"roofline.*' $R/gpurun_out/prof_$1.log | cut -c1-400
