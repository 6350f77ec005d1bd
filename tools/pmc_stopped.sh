#!/bin/bash
# usage: tools/pmc_stopped.sh TAG N -- HBM bytes fetched by the decode attention with N of 16 utterances finished (early exit)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MICRO_STOPPED=$2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_$1_stopped$2 -- $R/tools/micro_dec 16 536 0 1 > $R/gpurun_out/pmc_$1_stopped$2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_$1_stopped$2/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("void attn_decode"):
        acc[r["Kernel_Name"][:45]][0] += 1; acc[r["Kernel_Name"][:45]][1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(f"stopped=$2  {k}  launches={n}  fetched per launch = {v / n * 1024 * 2 / 1e6:.2f} MB (FETCH_SIZE x 2)")
PY
