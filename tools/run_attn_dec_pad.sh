#!/bin/bash
# dev: does the row count of the KV cache (stride between (utterance, head) regions) matter to the decode attention?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VALLE_ATTN_DEC=${V:-1}
for cfg in ${CFGS:-"32 2000 0" "32 1000 0" "16 536 0" "16 330 0"}; do set -- $cfg
  export MICRO_CT_PAD=$3
  rm -rf /tmp/adp
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/adp -- $R/tools/micro_dec $1 $2 0 1 > /tmp/adp.log 2>&1 || { tail -5 /tmp/adp.log; exit 1; }
  echo "== variant $VALLE_ATTN_DEC B=$1 ctx=$2 pad=$3: $(grep -h attn_decode /tmp/adp/*/*_kernel_stats.csv | awk -F'",' '{print $2}' | cut -d, -f2,3)"
done
