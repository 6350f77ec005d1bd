#!/bin/bash
# usage: tools/pmc_attn.sh TAG -- SQ counters of the dense-pass attention kernel (tools/time_attn.py), one pass per counter set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/time_attn.py || exit 1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_attn_$1_$i -- python3 $R/tools/time_attn.py 761 16 3 > /tmp/pmc_attn_$1_$i.log 2>&1 || { tail -5 /tmp/pmc_attn_$1_$i.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_attn_$1_$i/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if "attn_flash" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"  {k:36s} per launch = {v / n:16.0f}")
PY
done
