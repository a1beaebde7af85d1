"""Condense the rocprofv3 output of scripts/profile_round.sh into small files fit for profiles/:
kernel_stats.csv (bench), kernel_stats_ops.csv (ops script), hbm_traffic.json (PMC passes, gfx950 correction).

hbm_traffic.json, per kernel: launches, raw / corrected FETCH bytes and WRITE bytes per launch (mean), plus the MEDIAN and
MAX launch (a bench run holds launches of very different size: setup refactors all 1024 QPs, the first rho update 605,
the stragglers 5; `max` of factor / tail kernels is the setup launch, `median` a 605-QP refactorisation at steps >= 2).
Correction of FETCH_SIZE (MI355X_MICROARCH.md, HBM section: the counter tallies 128-B requests as 64 B for wide
streaming reads): x2 for 16-B-per-lane loads (the guide's calibration); x1.84 for the 8-B-per-lane buffer loads of the
step streams (iterate / check / kkt_solve / spmv kernels: calibrated in round 1 on kkt_solve_kernel, whose bytes are
known: 1.96 GB per launch against 1.064 GB raw); kernels with mixed widths (factor_kernel, tail kernels: 8-B gathers and
16-B tile loads) use x2 and are marked uncalibrated."""
import csv, glob, json, os, statistics, sys
out = sys.argv[1]
def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None
for sub, name in (("stats", "kernel_stats.csv"), ("stats_ops", "kernel_stats_ops.csv")):
    f = find(sub, "*kernel_stats.csv")
    if f:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(out, name), "w", newline="") as g:
            csv.writer(g).writerows([[c[:160] for c in r] for r in rows[:25]])
def pmc(sub, counter):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if not f: return acc
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter: continue
        a = acc.setdefault(r["Kernel_Name"], {})
        a[r.get("Dispatch_Id")] = a.get(r.get("Dispatch_Id"), 0.0) + float(r["Counter_Value"]) * 1024.0      # counter values are KiB
    return acc
STREAM_8B = ("iterate_kernel", "check_kernel", "kkt_solve_kernel", "spmv_kernel", "warm_start_kernel", "kkt_trace_kernel")
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
res = {"_how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 bench.py "
               "--steps 2 --warmup 1 --no-cpu-baseline --no-secondary; counter values are KiB.  FETCH_SIZE under-reports wide streaming reads on "
               "gfx950 (MI355X_MICROARCH.md, HBM section): corrected x2 for 16-B-per-lane loads, x1.84 for the 8-B-per-lane buffer loads of "
               "the step-stream kernels (calibrated on kkt_solve_kernel in round 1: 1.96 GB known vs 1.064 GB raw); factor / tail kernels "
               "mix 8-B gathers and 16-B tile loads: x2, uncalibrated.  WRITE_SIZE is exact.", "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    short = k.split("(")[0]
    if "miosqp" not in short: continue
    corr = 1.84 if any(t in short for t in STREAM_8B) else 2.0
    e = {"fetch_correction": corr, "calibrated": any(t in short for t in STREAM_8B)}
    if k in fe:
        v = sorted(fe[k].values()); e["launches"] = len(v)
        e["fetch_bytes_raw_per_launch"] = sum(v) / len(v)
        e["fetch_bytes_corrected_per_launch"] = corr * e["fetch_bytes_raw_per_launch"]
        e["fetch_bytes_corrected_median_launch"] = corr * statistics.median(v)
        e["fetch_bytes_corrected_max_launch"] = corr * v[-1]
    if k in wr:
        v = sorted(wr[k].values())
        e["write_bytes_per_launch"] = sum(v) / len(v)
        e["write_bytes_median_launch"] = statistics.median(v)
        e["write_bytes_max_launch"] = v[-1]
    res["kernels"][short] = e
it = [v for k, v in res["kernels"].items() if "iterate_kernel" in k]
if it:
    res["bytes_per_launch"] = it[0].get("fetch_bytes_corrected_per_launch", 0.0) + it[0].get("write_bytes_per_launch", 0.0)
    res["kernel"] = [k for k in res["kernels"] if "iterate_kernel" in k][0]
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: round(vv / 1e9, 3) if "bytes" in kk else vv for kk, vv in v.items()} for k, v in res["kernels"].items()}, indent=1))
