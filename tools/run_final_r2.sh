#!/bin/bash
# dev: round-2 closing measurements: the bench line, its kernel trace, the other BASELINE configurations
set -o pipefail
timeout -k 10 400 python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err || { tail -20 gpurun_out/r2_final_bench.err; exit 1; }
cat gpurun_out/r2_final_bench.json
bash tools/prof3.sh r2f > gpurun_out/prof3_r2f.txt 2>&1 || { tail -20 gpurun_out/prof3_r2f.txt; exit 1; }
tail -2 gpurun_out/prof3_r2f.txt
timeout -k 10 300 python tools/time_configs.py narlen longform > gpurun_out/r2_final_configs.txt 2>&1 || { tail -20 gpurun_out/r2_final_configs.txt; exit 1; }
cat gpurun_out/r2_final_configs.txt
