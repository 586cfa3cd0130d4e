#!/usr/bin/env python3
"""Condenses a scripts/profile_r06.sh output directory (rocprofv3 CSVs) into a directory to be copied to profiles/:
  r06_kernel_stats_<wl>.csv          --stats kernel tables (our kernels + the total)
  r06_bench_under_rocprof_<wl>.json  the bench lines printed during those profiled runs
  r06_pmc_summary_<wl>.json          per-kernel PMC averages
  r06_issue_accounting_c3.json       where the SIMD cycles of the headline kernel go (matrix pipe, other vector issue, the rest)
  pmc_traffic.json                   HBM bytes per launch of the dominant kernel (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
                                     FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads, so the read side is
                                     doubled: MI355X_MICROARCH.md, section HBM)."""
import collections, csv, glob, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r06"
dst = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/profiles_r06"
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "")


for wl in ("c3", "c2", "noisy", "c5", "c4", "gen_v50", "gen_h64l16", "gen_noisy", "spec_v50", "spec_h64l16", "spec_noisy", "spec_t99", "small", "smallcall"):
    f = os.path.join(src, wl, "trace_kernel_stats.csv")
    if os.path.exists(f):
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        with open(os.path.join(dst, f"r06_kernel_stats_{wl}.csv"), "w") as o:
            o.write("Name,Calls,TotalDurationNs,AverageNs,PercentageOfAllKernels,MinNs,MaxNs\n")
            for r in rows:
                n = short(r["Name"])
                if "bnn" in n:
                    o.write(f"\"{n}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{100 * float(r['TotalDurationNs']) / tot:.3f},{r['MinNs']},{r['MaxNs']}\n")
            o.write(f"\"(all kernels incl. torch input generation)\",,{tot:.0f},,100,,\n")
    b = os.path.join(src, f"{wl}_bench.json")
    if os.path.exists(b) and os.path.getsize(b):
        open(os.path.join(dst, f"r06_bench_under_rocprof_{wl}.json"), "w").write(open(b).read())


def pmc(prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, prefix + "*", "pmc_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            if "bnn" in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[name]["_grid"] = [int(r["Grid_Size"])]
                acc[name]["_vgpr"] = [int(r["VGPR_Count"])]
                acc[name]["_lds"] = [int(r["LDS_Block_Size"])]
                acc[name]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


out = pmc("pmc_c3_")
json.dump(out, open(os.path.join(dst, "r06_pmc_summary_c3.json"), "w"), indent=1, sort_keys=True)
outn = pmc("pmc_noisy_")
json.dump(outn, open(os.path.join(dst, "r06_pmc_summary_noisy.json"), "w"), indent=1, sort_keys=True)
outg = pmc("pmc_gen_v50_")
json.dump(outg, open(os.path.join(dst, "r06_pmc_summary_generic_v50.json"), "w"), indent=1, sort_keys=True)
outs = pmc("pmc_spec_v50_")
json.dump(outs, open(os.path.join(dst, "r06_pmc_summary_spec_v50.json"), "w"), indent=1, sort_keys=True)


def accounting(d, n_simd=1024, n_xcd=8):
    """SIMD-cycle accounting: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is in SIMD-cycles; a non-MFMA vector
    instruction holds the issue port 4 cycles (64 lanes on a 16-lane SIMD)."""
    cyc = d["GRBM_GUI_ACTIVE"] / n_xcd
    per_simd = lambda v: v / n_simd
    mfma = per_simd(d["SQ_VALU_MFMA_BUSY_CYCLES"])
    valu = per_simd((d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) * 4.0)
    a = {"kernel_cycles": cyc, "clock_GHz": cyc / d["_dur_ns"], "matrix_pipe_busy_frac": mfma / cyc, "other_vector_issue_frac": valu / cyc,
         "neither_frac": 1.0 - (mfma + valu) / cyc,
         "mfma_per_kernel": d["SQ_INSTS_MFMA"], "other_valu_per_kernel": d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]}
    for k in ("SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
        if k in d:
            a[k] = d[k]
    if "SQ_WAVE_CYCLES" in d:   # wave-level view (quad-cycle units): share of a wave's resident time it spends in s_waitcnt / LDS waits
        a["wave_time_in_waitcnt_frac"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        a["wave_time_waiting_on_lds_frac"] = d.get("SQ_WAIT_INST_LDS", 0.0) / d["SQ_WAVE_CYCLES"]
        a["waves_resident_per_simd"] = d["SQ_WAVE_CYCLES"] * 4.0 / n_simd / cyc
    return a


acc = {}
for name, d in list(out.items()) + list(outn.items()) + list(outg.items()) + list(outs.items()):
    if ("forward_kernel" in name or "forward_generic_kernel" in name or "bnn_spec_forward" in name) and "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        acc[name] = accounting(d)
json.dump(acc, open(os.path.join(dst, "r06_issue_accounting.json"), "w"), indent=1, sort_keys=True)

dom = [k for k in out if "forward_kernel" in k]
traffic = {}
old = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "profiles", "pmc_traffic.json")
if os.path.exists(old):
    try:
        traffic = json.load(open(old))
    except Exception:
        traffic = {}
traffic.pop("c2", None)   # a round-1 figure of a kernel that no longer exists
if dom and "FETCH_SIZE" in out[dom[0]]:
    d = out[dom[0]]
    fetch_kib, write_kib = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
    bench = json.loads([l for l in open(os.path.join(src, "c3_bench.json")) if l.startswith("{")][-1])
    B, J = bench["config"]["systems_per_gpu"], bench["config"]["draws"]
    hit = d.get("TCC_HIT_sum", 0.0) / max(d.get("TCC_HIT_sum", 0.0) + d.get("TCC_MISS_sum", 0.0), 1.0)
    lease = {"round": 6, "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib, "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024 + write_kib * 1024,
             "tcc_hit_rate": hit, "kernel_ms_under_pmc": d.get("_dur_ns", 0.0) / 1e6}
    prev = traffic.get("c3", {})
    leases = prev.get("leases") or ([{"round": 4, "FETCH_SIZE_KiB": prev.get("FETCH_SIZE_KiB"), "WRITE_SIZE_KiB": prev.get("WRITE_SIZE_KiB"),
                                      "hbm_bytes_per_launch": prev.get("hbm_bytes_per_launch"), "tcc_hit_rate": prev.get("tcc_hit_rate")}] if prev else [])
    leases = [l for l in leases if l.get("round") != 6] + [lease]
    traffic["c3"] = {"systems": B, "draws": J, "kernel": dom[0], "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
                     "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024 + write_kib * 1024, "tcc_hit_rate": hit,
                     "correction": "read side x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane loads), write side x1",
                     "algorithmic_bytes_per_launch": B * J * 16408,
                     "leases": leases,
                     "note": "the figure moves from lease to lease with the L2 hit rate of the XCD work order (which workgroups happen to be co-resident): "
                             "every lease on file is listed; bench.py quotes the latest and says traffic_measured_in_run: false",
                     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum (separate passes) on `python3 bench.py --workload c3`, scripts/profile_r06.sh (round 6)"}
    sc = [k for k in out if "nonfinite_scan" in k]
    if sc and "FETCH_SIZE" in out[sc[0]]:
        ds_ = out[sc[0]]
        traffic["c3_nonfinite_scan"] = {"kernel": sc[0], "FETCH_SIZE_KiB": ds_["FETCH_SIZE"], "hbm_bytes_per_launch": 2.0 * ds_["FETCH_SIZE"] * 1024 + ds_.get("WRITE_SIZE", 0.0) * 1024,
                                        "algorithmic_bytes_per_launch": B * 16400, "kernel_ms_under_pmc": ds_.get("_dur_ns", 0.0) / 1e6}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
# the one-box consistency record: un-profiled line(s), the profiled kernel average, the clock the box held
one = {}
for tag in ("default_bench", "default_bench_after"):
    f = os.path.join(src, tag + ".json")
    if os.path.exists(f) and os.path.getsize(f):
        open(os.path.join(dst, "r06_" + tag.replace("default_bench", "bench_default_c3") + ".json"), "w").write(open(f).read())
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        one[tag] = {"ms_per_step": r["ms_per_step"], "kernel_ms": r["roofline"]["kernel_ms"], "value": r["value"], "frac": r["roofline"]["frac"],
                    "frac_executed": r["roofline"]["frac_executed"], "frac_at_held_clock": r["roofline"].get("frac_at_held_clock"),
                    "finite_check_ms": r["config"].get("finite_check_ms"), "clock": r.get("clock")}
ks = os.path.join(src, "c3", "trace_kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if "bnn_forward_kernel" in r["Name"]:
            one["rocprof_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
            one["rocprof_kernel_calls"] = int(r["Calls"])
        if "bnn_nonfinite_scan_kernel" in r["Name"]:
            one["rocprof_nonfinite_scan_avg_ms"] = float(r["AverageNs"]) / 1e6
        if "bnn_nonfinite_fixup_kernel" in r["Name"]:
            one["rocprof_nonfinite_fixup_avg_ms"] = float(r["AverageNs"]) / 1e6
for name, a in acc.items():
    if "forward_kernel<31" in name:
        one["held_clock_GHz_under_pmc"] = a["clock_GHz"]
        one["matrix_pipe_busy_frac"] = a["matrix_pipe_busy_frac"]
json.dump(one, open(os.path.join(dst, "r06_one_box.json"), "w"), indent=1)
print(json.dumps(one, indent=1))
print(json.dumps(traffic, indent=1))
print(json.dumps(acc, indent=1, sort_keys=True))
