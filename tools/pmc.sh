#!/bin/bash
# usage: tools/pmc.sh TAG -- HBM traffic counters of the decode layer chain (tools/micro_dec, eager), one counter per pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$1_$C -- $R/tools/micro_dec 16 536 0 1 > $R/gpurun_out/pmc_$1_$C.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/pmc_$1_%s/*/*_counter_collection.csv" % C)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != C: continue
        k = r["Kernel_Name"][:60]; acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    print(C)
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {k:60s} launches={n:5d}  per-launch = {v / n / 1024:9.3f} MB (raw counter KB/1024)")
PY
