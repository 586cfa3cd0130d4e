#!/usr/bin/env python3
"""What the specialiser's measured selection sees: every candidate form of a list of networks timed on the tuning grid (specialize.py
_measure_on: 16 384 systems x 16 draws) -> JSON lines {net, noisy, chosen, candidates: [{w8, flags, scratch, ms}]}.
usage (GPU box): python scripts/spec_tuning_r04.py > gpurun_out/r4_spec_tuning.jsonl"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bnn_chaos_model_amd import ops  # noqa: E402

NETS = [(40, 20, 1, 1, 41), (64, 16, 1, 1, 41), (20, 10, 1, 1, 41), (40, 20, 2, 2, 41), (40, 20, 1, 1, 82), (80, 20, 1, 1, 41), (64, 32, 1, 1, 41),
        (100, 30, 1, 1, 41), (128, 32, 1, 1, 41)]
for (H, L, din, dout, NF) in NETS:
    plan = ops.get_plan(hidden=H, latent=L, depth_in=din, depth_out=dout, n_features=NF)
    ops.specialize(plan, noisy=(False, True))
    for nz in (False, True):
        info = plan.spec_info[nz]
        print(json.dumps({"net": [H, L, din, dout, NF], "noisy": nz, "chosen": {"w8": info["w8"], "flags": info["flags"], "scratch": info["scratch"],
                          "vgpr": info["vgpr"], "agpr": info["agpr"], "lds": info["lds"], "nwaves": info.get("nwaves")},
                          "tuned_ms": info.get("tuned_ms"), "candidates": info.get("candidates")}), flush=True)
