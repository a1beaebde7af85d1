#!/bin/bash
# Runs ON THE GPU BOX: per-launch durations of one bench step (rocprofv3 kernel trace), printed as a timeline.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/launches
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True))[-1]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("miosqp::", "")[:44]) for r in csv.DictReader(open(f)))
big = [e for e in ev if ("iterate" in e[2] or "factor" in e[2] or "check" in e[2] or "tail" in e[2])]
last = big[-24:]
for s, e, n in last:
    print(f"{n:46s} {(e - s) / 1e3:9.1f} us")
PY
