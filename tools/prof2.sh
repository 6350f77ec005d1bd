#!/bin/bash
# usage: tools/prof2.sh TAG WHICH -- kernel-trace profile of tools/time_configs.py WHICH (narlen | longform)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2_$1 -- python $R/tools/time_configs.py $2 > $R/gpurun_out/prof2_$1.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof2_$1/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.2f} {float(r['Percentage']):5.1f}%")
PY
tail -2 $R/gpurun_out/prof2_$1.log
