#!/usr/bin/env python3
"""Same-box A/B of library variants on the headline kernel: each variant is a full build of libbnn_chaos_hip.so with extra -D flags
(`python -m bnn_chaos_model_amd.csrc.build -DX -o libbnn_X.so`, done in the build container; the .so files travel with the
snapshot).  One child process per (variant, repetition), interleaved, so that clock drift of the box hits all variants alike.

  python scripts/ab_variants.py [--workload c3] [--reps 3] libA.so libB.so ...      (names relative to bnn_chaos_model_amd/csrc)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("libs", nargs="+")
args, extra = ap.parse_known_args()

res = {l: [] for l in args.libs}
for rep in range(args.reps):
    for l in args.libs:
        env = dict(os.environ, BNN_CHAOS_SO=os.path.join(ROOT, "bnn_chaos_model_amd", "csrc", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", "1",
                              "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", out.stderr[-500:], flush=True)
            continue
        r = json.loads(line[-1])
        res[l].append(r["roofline"]["kernel_ms"])
        print(f"rep {rep} {l:40s} kernel_ms {r['roofline']['kernel_ms']:.2f}  evals/s {r['value']:.4g}  frac {r['roofline']['frac']:.4f}", flush=True)
print()
for l, v in res.items():
    if v:
        print(f"{l:40s} min {min(v):.2f} ms  mean {sum(v) / len(v):.2f} ms  ({len(v)} runs)")
