#!/bin/bash
# usage: tools/pmc_lds.sh TAG -- LDS counters of the dense-pass GEMMs (tools/micro_gemm, M = 12,176), one pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/vall-e_amd:$LD_LIBRARY_PATH
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcl_$1 -- $R/tools/micro_gemm > $R/gpurun_out/pmcl_$1.log 2>&1 || { tail -5 $R/gpurun_out/pmcl_$1.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$R/gpurun_out/pmcl_$1/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
out = {}
for k, c in acc.items():
    if n[k] == 0 or "gemm" not in k: continue
    per = {x: v / n[k] for x, v in c.items()}
    per["kernel_cycles"] = per["GRBM_GUI_ACTIVE"] / 8.0
    out[k] = per
    print(k, json.dumps(per))
json.dump(out, open("$R/gpurun_out/pmcl_$1.json", "w"), indent=1)
PY
