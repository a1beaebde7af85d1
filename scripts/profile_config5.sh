#!/bin/bash
# GPU box: kernel trace of the single large QP (config 5, dataflow form) - scripts/mw_probe.py with the default shape.
# Output: gpurun_out/prof_c5/ (copy kernel_stats to profiles/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_c5 && mkdir -p $R/gpurun_out/prof_c5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python3 $R/scripts/mw_probe.py 128:128 > $R/gpurun_out/prof_c5/run.log 2>&1
f=$(find $R/gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-220 > $R/gpurun_out/prof_c5/kernel_stats_head.csv
tail -3 $R/gpurun_out/prof_c5/run.log
cat $R/gpurun_out/prof_c5/kernel_stats_head.csv
