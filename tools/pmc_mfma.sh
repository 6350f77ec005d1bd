#!/bin/bash
# usage: tools/pmc_mfma.sh TAG -- MFMA counters of the dense-pass GEMMs (tools/micro_gemm: the four GEMMs of one NAR layer, M = 12,176),
# one shape per run (o_proj and down_proj share a kernel instantiation); writes gpurun_out/TAG_pmc_mfma_dense_gemms.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/vall-e_amd:$LD_LIBRARY_PATH
for ROLE in qkv o gu down; do
  export MICRO_GEMM_ONLY=$ROLE
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcm_$1_$ROLE -- $R/tools/micro_gemm > $R/gpurun_out/pmcm_$1_$ROLE.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcm_$1_${ROLE}_trace -- $R/tools/micro_gemm > $R/gpurun_out/pmcm_$1_${ROLE}_trace.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections, json, re
shape = {"qkv": "M=12176 N=3072 K=1024, RMSNorm folded + RoPE", "o": "M=12176 N=1024 K=1024, + residual",
         "gu": "M=12176 N=8192 K=1024, RMSNorm folded + GEGLU", "down": "M=12176 N=1024 K=4096, + residual"}
flops = {"qkv": 2 * 12176 * 3072 * 1024, "o": 2 * 12176 * 1024 * 1024, "gu": 2 * 12176 * 8192 * 1024, "down": 2 * 12176 * 1024 * 4096}
out = {}
for role in ("qkv", "o", "gu", "down"):
    f = glob.glob("$R/gpurun_out/pmcm_$1_%s/*/*_counter_collection.csv" % role)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    dur = {}
    for r in csv.DictReader(open(glob.glob("$R/gpurun_out/pmcm_$1_%s_trace/*/*_kernel_stats.csv" % role)[0])):
        dur[re.sub(r"\(.*", "", r["Name"]).replace("void ", "")] = float(r["AverageNs"]) / 1e3
    for k, c in acc.items():
        if n[k] == 0 or "gemm" not in k: continue
        per = {x: v / n[k] for x, v in c.items()}
        cyc = per["GRBM_GUI_ACTIVE"] / 8.0                      # the counter is summed over the 8 XCDs
        us = dur.get(k)
        out[role + ": " + k] = dict(shape=shape[role], launches=n[k], avg_us=us, tflops=None if not us else flops[role] / us * 1e-6,
                                    kernel_cycles=cyc, **per, mfma_busy_frac=per["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4))
        print(role, k, json.dumps(out[role + ": " + k]))
json.dump(out, open("$R/gpurun_out/$1_pmc_mfma_dense_gemms.json", "w"), indent=1)
PY
