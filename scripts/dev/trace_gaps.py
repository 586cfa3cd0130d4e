"""Reads a rocprofv3 kernel-trace CSV and prints, per distinct chain of kernels between two idle gaps, each kernel's duration and the gap
in front of it (microseconds): python scripts/dev/trace_gaps.py <kernel_trace.csv> [n_chains]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
chains, cur, prev_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    if prev_end is not None and gap > 40.0 and cur:
        chains.append(cur)
        cur = []
        gap = 0
    cur.append((r["Kernel_Name"].split("(")[0][:70], (e - s) / 1e3, gap))
    prev_end = e
if cur:
    chains.append(cur)
sig = {}
for c in chains:
    sig.setdefault(tuple(k for k, _, _ in c), []).append(c)
for names, cs in sig.items():
    if len(cs) < 5:
        continue
    print(f"--- chain of {len(names)} kernels, seen {len(cs)} times; median over occurrences (us): duration | gap in front")
    import statistics
    tot = statistics.median(sum(d + g for _, d, g in c) for c in cs)
    for i, nm in enumerate(names):
        print(f"  {nm:70s} {statistics.median(c[i][1] for c in cs):8.2f} | {statistics.median(c[i][2] for c in cs):6.2f}")
    print(f"  chain first-start to last-end: {tot:.1f} us")
