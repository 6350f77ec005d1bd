#!/bin/bash
# dev: decode-layer chain under the K-split variants of the wave-streaming linears
for cfg in ${VARIANTS:-"1 4" "2 8" "4 8"}; do set -- $cfg
  echo "== NWK=$1 DN=$2"
  VALLE_DEC_NWK=$1 VALLE_DEC_NWK_DN=$2 timeout -k 10 60 ./tools/micro_dec 16 536 | grep -E "^wave|WAVE" || exit 1
done
echo "== NWK=2 DN=8 attention ctx by value"
ATTN_BYVAL=1 VALLE_DEC_NWK=2 VALLE_DEC_NWK_DN=8 timeout -k 10 60 ./tools/micro_dec 16 536 | grep -E "^wave|WAVE" || exit 1
