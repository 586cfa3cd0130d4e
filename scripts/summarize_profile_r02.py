#!/usr/bin/env python3
"""Condenses a scripts/profile_r02.sh output directory (rocprofv3 CSVs) into a directory to be copied to profiles/: the --stats
kernel tables (our kernels only + the total), per-kernel PMC averages, and pmc_traffic.json (HBM bytes per launch of the dominant
kernel, corrected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of wide (16 B/lane) streaming reads, so the read side is doubled)."""
import collections, csv, glob, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02"
dst = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/profiles_r02"
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "")


for wl in ("c3", "c2", "noisy"):
    f = os.path.join(src, wl, "trace_kernel_stats.csv")
    if os.path.exists(f):
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        with open(os.path.join(dst, f"r02_kernel_stats_{wl}.csv"), "w") as o:
            o.write("Name,Calls,TotalDurationNs,AverageNs,PercentageOfAllKernels,MinNs,MaxNs\n")
            for r in rows:
                n = short(r["Name"])
                if "bnn" in n:
                    o.write(f"\"{n}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{100 * float(r['TotalDurationNs']) / tot:.3f},{r['MinNs']},{r['MaxNs']}\n")
            o.write(f"\"(all kernels incl. torch input generation)\",,{tot:.0f},,100,,\n")
    b = os.path.join(src, f"{wl}_bench.json")
    if os.path.exists(b) and os.path.getsize(b):
        open(os.path.join(dst, f"r02_bench_under_rocprof_{wl}.json"), "w").write(open(b).read())
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = short(r["Kernel_Name"])
        if "bnn" in name:
            pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc[name]["_grid"] = [int(r["Grid_Size"])]
            pmc[name]["_vgpr"] = [int(r["VGPR_Count"])]
            pmc[name]["_lds"] = [int(r["LDS_Block_Size"])]
            pmc[name]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
json.dump(out, open(os.path.join(dst, "r02_pmc_summary_c3.json"), "w"), indent=1, sort_keys=True)
dom = [k for k in out if "forward_kernel" in k]
traffic = {}
old = os.path.join("profiles", "pmc_traffic.json")
if os.path.exists(old):
    try:
        prev = json.load(open(old))
        traffic = prev if "c2" in prev or "c3" in prev else {prev.get("workload", "c2"): prev}
    except Exception:
        traffic = {}
if dom:
    d = out[dom[0]]
    fetch_kib, write_kib = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
    bench = json.load(open(os.path.join(src, "c3_bench.json")))
    B, J = bench["config"]["systems_per_gpu"], bench["config"]["draws"]
    traffic["c3"] = {"systems": B, "draws": J, "kernel": dom[0], "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
                     "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024 + write_kib * 1024,
                     "correction": "read side x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane loads), write side x1",
                     "algorithmic_bytes_per_launch": B * J * 16408,
                     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --workload c3`, scripts/profile_r02.sh"}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps(out, indent=1, sort_keys=True)[:4000])
