#!/bin/bash
# dev: decode layer chain with a side-stream KV prefetch (tools/micro_dec MICRO_PREFETCH modes)
set -e
for m in 1 2 3; do
  echo "== MICRO_PREFETCH=$m"
  MICRO_PREFETCH=$m timeout -k 10 120 ./tools/micro_dec 16 536 | grep -E "WAVE LAYER|error"
done
