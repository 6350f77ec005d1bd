#!/bin/bash
# usage: tools/prof3.sh TAG -- kernel-trace of one bench step + idle-gap analysis (tools/timeline.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$1.log 2>&1 || exit 1
mkdir -p $R/gpurun_out/prof_$1
cp /tmp/prof_$1/*/*_kernel_stats.csv $R/gpurun_out/prof_$1/kernel_stats.csv
python $R/tools/timeline.py /tmp/prof_$1 8 > $R/gpurun_out/prof_$1/timeline.txt 2>&1
cat $R/gpurun_out/prof_$1/timeline.txt
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_$1.log
