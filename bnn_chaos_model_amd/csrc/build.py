"""Builds libbnn_chaos_hip.so (gfx950 only) in-tree with hipcc: one object per translation unit, compiled in parallel, then one
link.  `python -m bnn_chaos_model_amd.csrc.build [--force] [extra hipcc flags] [-o other_name.so]`"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libbnn_chaos_hip.so")
OBJDIR = os.path.join(HERE, "build")
# translation units (bnn_internal.h says what each holds); the forward-kernel units dominate the build time
SRCS = ["bnn_fwd_generic.hip", "bnn_fwd_generic82.hip", "bnn_fwd_v50spec.hip", "bnn_fwd_k31.hip", "bnn_fwd_small.hip", "bnn_fwd_k41.hip", "bnn_fwd_noisy.hip", "bnn_fwd_stats.hip", "bnn_fwd_megno.hip", "bnn_fwd_lowp.hip",
        "bnn_abi.hip", "bnn_ops_draw.hip", "bnn_ops_reduce.hip", "bnn_ops_stats.hip", "bnn_ops_features.hip", "bnn_nonfinite.hip", "bnn_tables.cpp", "bnn_generic.cpp"]
HEADERS = ["bnn_layout.h", "bnn_tables.h", "bnn_internal.h", "bnn_abi_common.h", "bnn_common.hip.h", "bnn_stats.hip.h", "bnn_forward.hip.h",
           "bnn_lowp.hip.h", "bnn_generic.h", "bnn_generic.hip.h", os.path.join("..", "..", "include", "bnn_chaos_hip.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
# per-unit flags: the reduced-precision unit writes its packed arithmetic out by hand and needs the SLP vectoriser off, or the
# compiler rebuilds convert + v_pk_fma_f32 where v_fma_mixlo/hi_f16 is wanted (bnn_lowp.hip.h, split_pair)
UNIT_FLAGS = {"bnn_fwd_lowp.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; this library has no CPU build)")


def clean_env():
    """Environment for the compiler subprocesses: without a profiler's preloaded tool library (under `rocprofv3 --pmc` every child that
    inherits LD_PRELOAD initialises the GPU before its main() and then execs its own helpers -- which GPU pools may refuse)."""
    drop = ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_"))}


def _mtime(name):
    return os.path.getmtime(os.path.join(HERE, name))


def source_hash(extra=()):
    """Content hash of everything the library is built from (file times do not survive a copy to another machine)."""
    import hashlib
    h = hashlib.sha256((" ".join(CFLAGS + list(extra)) + repr(sorted(UNIT_FLAGS.items()))).encode())
    for name in sorted(SRCS + HEADERS):
        h.update(name.encode())
        with open(os.path.join(HERE, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stale(so=None):
    """True when the library is missing or was built from other sources (or other flags) than the default build of the ones next
    to it."""
    so = so or SO
    try:
        with open(so + ".srchash") as f:
            return not os.path.exists(so) or f.read().strip() != source_hash()
    except OSError:
        return True


def build(force=False, verbose=False, extra=(), out=None):
    """extra: additional hipcc flags; out: alternative .so name for A/B builds (always a full rebuild)."""
    so = SO if out is None else os.path.join(HERE, out)
    if not force and out is None and not stale():
        return so
    cc = hipcc()
    objdir = OBJDIR if out is None else os.path.join(HERE, "build_" + os.path.splitext(out)[0])
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(_mtime(h) for h in HEADERS)
    # objects are reusable only under the flags they were compiled with: the object directory carries a stamp of CFLAGS + extra
    flags_stamp = " ".join(CFLAGS + list(extra)) + repr(sorted(UNIT_FLAGS.items()))
    stamp_file = os.path.join(objdir, "FLAGS")
    try:
        with open(stamp_file) as f:
            same_flags = f.read() == flags_stamp
    except OSError:
        same_flags = False
    if not same_flags:
        # objects compiled under other flags are not reusable: drop them, and write the stamp only AFTER every unit has compiled (a failed
        # or interrupted build must not leave a stamp that vouches for a mix of old and new objects)
        force = True
        for name in os.listdir(objdir):
            if name.endswith(".o") or name == "FLAGS":
                os.remove(os.path.join(objdir, name))

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if not force and out is None and os.path.exists(obj) and os.path.getmtime(obj) > max(_mtime(src), hdr_t):
            return obj
        cmd = [cc] + CFLAGS + UNIT_FLAGS.get(src, []) + list(extra) + build_flags_define + ["-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE, env=clean_env())
        return obj

    # a build with extra switches names itself: bnn_build_flags() returns them, bench.py writes them into its JSON line
    build_flags_define = ['-DBNN_BUILD_FLAGS="%s"' % " ".join(a[2:] if a.startswith("-D") else a for a in extra)] if extra else []
    workers = min(len(SRCS), max(1, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(workers) as ex:
        objs = list(ex.map(compile_one, SRCS))
    with open(stamp_file, "w") as f:
        f.write(flags_stamp)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", so + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE, env=clean_env())
    os.replace(so + ".tmp", so)
    with open(so + ".srchash", "w") as f:
        f.write(source_hash(extra))
    return so


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = None
    if "-o" in args:
        i = args.index("-o")
        out = args[i + 1]
        del args[i:i + 2]
    print(build(force="--force" in sys.argv[1:] or out is not None, verbose=True, extra=args, out=out))
