#!/bin/bash
# dev: two bench ranks on one GPU over gloo (the multi-rank path without a second GPU); tracebacks on a hang
export VALLE_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 PYTHONFAULTHANDLER=1
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 ${LIMIT:-150} python -u bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/rehearse_rank$r.out 2> gpurun_out/rehearse_rank$r.err &
done
wait
for r in 0 1; do echo "== rank $r"; cut -c1-300 gpurun_out/rehearse_rank$r.out; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/rehearse_rank$r.err | tail -25; done
