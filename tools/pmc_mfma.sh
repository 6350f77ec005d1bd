#!/bin/bash
# usage: tools/pmc_mfma.sh TAG -- MFMA counters of the dense-pass GEMMs (tools/micro_gemm: the four GEMMs of one NAR layer, M = 12,176)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/vall-e_amd:$LD_LIBRARY_PATH
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcm_$1 -- $R/tools/micro_gemm > $R/gpurun_out/pmcm_$1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcm_$1_trace -- $R/tools/micro_gemm > $R/gpurun_out/pmcm_$1_trace.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$R/gpurun_out/pmcm_$1/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
dur = {}
for r in csv.DictReader(open(glob.glob("$R/gpurun_out/pmcm_$1_trace/*/*_kernel_stats.csv")[0])):
    dur[r["Name"][:48]] = float(r["AverageNs"]) / 1e3
out = {}
for k, c in acc.items():
    if n[k] == 0 or "gemm" not in k: continue
    per = {x: v / n[k] for x, v in c.items()}
    cyc = per["GRBM_GUI_ACTIVE"] / 8.0                      # the counter is summed over the 8 XCDs
    out[k] = dict(launches=n[k], avg_us=dur.get(k), kernel_cycles=cyc, **per,
                  mfma_busy_frac=per["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4))
    print(k, json.dumps(out[k]))
json.dump(out, open("$R/gpurun_out/pmcm_$1.json", "w"), indent=1)
PY
