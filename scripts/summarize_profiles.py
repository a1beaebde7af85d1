"""Condense the rocprofv3 output of scripts/profile_round.sh into small files fit for profiles/:
kernel_stats.csv (bench), kernel_stats_ops.csv (ops script), hbm_traffic.json (PMC passes, gfx950 correction)."""
import csv, glob, json, os, sys
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
        k = r["Kernel_Name"]
        a = acc.setdefault(k, {"launch_ids": set(), "sum": 0.0})
        a["launch_ids"].add(r.get("Dispatch_Id"))
        a["sum"] += float(r["Counter_Value"])
    return acc
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
res = {"_how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 bench.py "
               "--steps 2 --warmup 1 --no-cpu-baseline; counter values are KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports "
               "exactly 1/2 of the bytes of wide coalesced streaming reads (16 B/lane, what iterate_kernel<2,512> issues) on gfx950, so "
               "fetch bytes = FETCH_SIZE*1024*2; WRITE_SIZE is exact.", "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    short = k.split("(")[0]
    if "miosqp" not in short: continue
    e = {}
    if k in fe:
        n = len(fe[k]["launch_ids"]); e["launches"] = n
        e["fetch_bytes_raw_per_launch"] = fe[k]["sum"] * 1024 / n
        e["fetch_bytes_corrected_per_launch"] = 2 * e["fetch_bytes_raw_per_launch"]
    if k in wr:
        n = len(wr[k]["launch_ids"])
        e["write_bytes_per_launch"] = wr[k]["sum"] * 1024 / n
    res["kernels"][short] = e
it = [v for k, v in res["kernels"].items() if "iterate_kernel" in k]
if it:
    res["bytes_per_launch"] = it[0].get("fetch_bytes_corrected_per_launch", 0.0) + it[0].get("write_bytes_per_launch", 0.0)
    res["kernel"] = [k for k in res["kernels"] if "iterate_kernel" in k][0]
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: round(vv / 1e9, 3) if "bytes" in kk else vv for kk, vv in v.items()} for k, v in res["kernels"].items()}, indent=1))
