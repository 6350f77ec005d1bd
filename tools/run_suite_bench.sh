#!/bin/bash
# dev: the GPU test suite, then the bench line (each step under its own time limit; stops at the first failure)
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/${1}_tests.log 2>&1 || { tail -30 gpurun_out/${1}_tests.log; exit 1; }
tail -3 gpurun_out/${1}_tests.log
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/${1}_bench.json 2> gpurun_out/${1}_bench.err || { tail -20 gpurun_out/${1}_bench.err; exit 1; }
cat gpurun_out/${1}_bench.json
