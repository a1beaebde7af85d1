#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats of bench.py and of the ops script, then the two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) the MI355X guide prescribes.  Summaries land in gpurun_out/prof_<tag>/.
# usage: bash scripts/profile_round.sh <tag>
set -e
TAG=${1:-v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $ROOT/bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/stats_bench.json 2> $OUT/stats.err
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ops -- python3 $ROOT/scripts/profile_ops.py > $OUT/ops.txt 2> $OUT/stats_ops.err
echo "ops stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
python3 $ROOT/scripts/summarize_profiles.py $OUT
