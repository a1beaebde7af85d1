#!/bin/bash
# GPU box: HBM traffic of the single large QP (config 5 at its literal size, dataflow form): kernel stats + the two separate
# PMC passes (FETCH_SIZE, WRITE_SIZE) of MI355X_MICROARCH.md on scripts/config5_iterate.py (200 iterations).
# Output: gpurun_out/prof_c5/ (+ config5_traffic.json for profiles/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_c5
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/scripts/config5_iterate.py > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/scripts/config5_iterate.py > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/scripts/config5_iterate.py > $O/write.log 2>&1
python3 - <<PY
import csv, glob, json, statistics
O = "$O"
def find(sub, pat):
    r = glob.glob(O + "/" + sub + "/**/" + pat, recursive=True)
    return r[0] if r else None
out = {"_how": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/config5_iterate.py; "
               "counter values are KiB; FETCH_SIZE corrected x1.84 for the 8-B-per-lane stream loads (calibration of round 1, scripts/summarize_profiles.py)", "kernels": {}}
f = find("stats", "*kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    open(O + "/kernel_stats.csv", "w").write("\n".join(",".join(c[:140] for c in r) for r in rows[:12]) + "\n")
    for r in csv.DictReader(open(f)):
        if "iterate_kernel" in r["Name"] or "check_kernel" in r["Name"] or "factor_kernel" in r["Name"]:
            out["kernels"].setdefault(r["Name"].split("(")[0], {}).update({"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])})
for sub, ctr, key in (("pmc_fetch", "FETCH_SIZE", "fetch_raw"), ("pmc_write", "WRITE_SIZE", "write")):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if f:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr: continue
            acc.setdefault(r["Kernel_Name"].split("(")[0], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"]) * 1024.0
    for k, v in acc.items():
        if not any(t in k for t in ("iterate_kernel", "check_kernel", "factor_kernel")): continue
        out["kernels"].setdefault(k, {})[key + "_bytes_median_launch"] = statistics.median(v.values())
        out["kernels"][k][key + "_launches"] = len(v)
json.dump(out, open(O + "/config5_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
tail -n 3 $O/stats.log
