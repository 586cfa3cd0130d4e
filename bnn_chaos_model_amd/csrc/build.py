"""Builds libbnn_chaos_hip.so (gfx950 only) in-tree with hipcc.  `python -m bnn_chaos_model_amd.csrc.build`"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libbnn_chaos_hip.so")
SRCS = ["bnn_kernels.hip", "bnn_tables.cpp"]
DEPS = SRCS + ["bnn_layout.h", "bnn_tables.h", "bnn_common.hip.h", "bnn_engine_a.hip.h", "bnn_engine_b.hip.h", os.path.join("..", "..", "include", "bnn_chaos_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; this library has no CPU build)")


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in DEPS)


def build(force=False, verbose=False, extra=(), out=None):
    """extra: additional hipcc flags (e.g. -DBNN_WAVES_PER_SIMD=3); out: alternative .so name for A/B builds."""
    so = SO if out is None else os.path.join(HERE, out)
    if not force and out is None and not stale():
        return so
    cmd = [hipcc()] + FLAGS + list(extra) + ["-x", "hip"] + [os.path.join(HERE, s) for s in SRCS] + ["-o", so + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    os.replace(so + ".tmp", so)
    return so


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = None
    if "-o" in args:
        i = args.index("-o")
        out = args[i + 1]
        del args[i:i + 2]
    print(build(force=True, verbose=True, extra=args, out=out))
