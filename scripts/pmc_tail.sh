#!/bin/bash
# developer script (GPU box): SQ counters of the refactorisation kernels (own --pmc pass, kernel trace only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tail
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $OUT/p1.json 2> $OUT/p1.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $OUT/p2.json 2> $OUT/p2.err
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no counters", open("$OUT/%s.err" % p).read()[-800:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("miosqp::", "")
        if not any(t in k for t in ("tail", "factor", "iterate")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in sorted(acc):
        print(k, "launches", len(cnt[k]), {c: "%.3g" % (v / len(cnt[k])) for c, v in sorted(acc[k].items())})
PY
