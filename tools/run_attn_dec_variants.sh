#!/bin/bash
# dev: decode attention variants (VALLE_ATTN_DEC) on the decode-layer chain, kernel-trace average of attn_decode
# usage: [B=32] [CTX=2000] tools/run_attn_dec_variants.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${B:-32}; CTX=${CTX:-2000}
for v in ${VARIANTS:-0 1}; do
  export VALLE_ATTN_DEC=$v
  rm -rf /tmp/ad_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ad_$v -- $R/tools/micro_dec $B $CTX 0 1 > /tmp/ad_$v.log 2>&1 || { tail -5 /tmp/ad_$v.log; exit 1; }
  echo "== variant $v B=$B ctx=$CTX"
  grep -h "attn_decode" /tmp/ad_$v/*/*_kernel_stats.csv | cut -d, -f1-4
done
