#!/usr/bin/env python3
"""Fixture for the two-sided and right-sided forms of fast_truncnorm (figures/multiswag_5_planet.py:306-370; the scripts
themselves only use left = 4, right = inf).  Build container only: the function's source is cut out of the reference file with
ast and executed as it is, with numpy's generator taped (as make_golden.py does for the one-sided case).

    python tests/golden/make_golden_truncnorm2.py   ->  case_truncnorm2.npz
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, save  # noqa: E402

src5 = open(f"{REF}/figures/multiswag_5_planet.py").read()
ftn = [n for n in ast.parse(src5).body if isinstance(n, ast.FunctionDef) and n.name == "fast_truncnorm"][0]
ns5 = {"np": np, "jnp": np}
exec(compile(ast.Module(body=[ftn], type_ignores=[]), "multiswag_5_planet.py:fast_truncnorm", "exec"), ns5)
rng = np.random.default_rng(123)
loc = rng.uniform(3.0, 12.5, size=(5, 33)).astype(np.float32)
scale = rng.uniform(0.5, 6.0, size=(5, 33)).astype(np.float32)
loc[0, :4] = 30.0; scale[0, :4] = 0.1           # nothing inside (4, 9): the first candidate is returned
out = {}
for name, left, right in (("two_sided", 4.0, 9.0), ("right_only", np.inf, 7.5)):
    np.random.seed(6100)
    tape = []
    o_normal = np.random.normal
    def rec(*a, **k):
        r = o_normal(*a, **k)
        tape.append(np.asarray(r).copy())
        return r
    np.random.normal = rec
    try:
        res = np.array(ns5["fast_truncnorm"](loc, scale, left=left, right=right, d=50, nsamp=12, seed=0))
    finally:
        np.random.normal = o_normal
    out[f"{name}_out"] = res
    out[f"{name}_left"] = np.array(left); out[f"{name}_right"] = np.array(right)
    out[f"{name}_normals"] = np.concatenate(tape, axis=1)       # [nsamp, n] in element order (chunks of d = 50 elements)
save("case_truncnorm2.npz", loc=loc, scale=scale, nsamp=np.array(12), d=np.array(50), **out)
