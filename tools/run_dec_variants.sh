#!/bin/bash
# dev: decode-layer chain under the K-split variants of the wave-streaming linears
# usage: [B=16] [CTX=536] [VARIANTS='"1 4" "2 8"'] tools/run_dec_variants.sh
B=${B:-16}; CTX=${CTX:-536}
for cfg in ${VARIANTS:-"1 4" "2 8" "4 8"}; do set -- $cfg
  echo "== B=$B ctx=$CTX NWK=$1 DN=$2"
  VALLE_DEC_NWK=$1 VALLE_DEC_NWK_DN=$2 timeout -k 10 60 ./tools/micro_dec $B $CTX | grep -E "^wave|WAVE" || exit 1
done
