"""Run-time specialisation of the generic forward engine.

The reference builds its network from hparams (spock_reg_model.py:301-321, 346-362) and PyTorch runs whatever comes out at the same
speed.  Here the ahead-of-time generic engine reads the shapes from a descriptor: every trip count is a run-time number behind an
early exit, and the kernel is a nest of small basic blocks the compiler cannot schedule across.  `specialize(plan)` compiles the
SAME kernel source (csrc/bnn_generic.hip.h) for the plan's one network with every shape a compile-time constant -- about ten seconds
of hipcc per form, cached on disk -- and attaches the code object to the plan; from then on every entry point that would take the
generic route launches it.  Accumulation order and arithmetic are those of the ahead-of-time form: results are bit-identical
(tests/test_hip_spec.py); only the schedule changes.

Needs hipcc at run time (ROCm's own compiler: this is a ROCm-only library).  No hipcc -> RuntimeError; nothing falls back silently,
and an un-specialised plan keeps working on the ahead-of-time form."""
import hashlib
import os
import subprocess
import tempfile

from . import _native as N
from .csrc import build as _build

CSRC = os.path.dirname(os.path.abspath(_build.__file__))
INCLUDE = os.path.join(CSRC, "..", "..", "include")
# the library's own code-generation flags: the specialised form must round exactly as the ahead-of-time one does
SPEC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--genco", "-x", "hip"]
_DEPS = ("bnn_generic.hip.h", "bnn_generic.h", "bnn_common.hip.h", "bnn_stats.hip.h", "bnn_internal.h", "bnn_layout.h")


def cache_dir():
    """BNN_SPEC_CACHE, else csrc/_spec next to the library (in-tree, like the built .so: it travels with the tree), else ~/.cache."""
    for d in (os.environ.get("BNN_SPEC_CACHE"), os.path.join(CSRC, "_spec"), os.path.join(os.path.expanduser("~"), ".cache", "bnn_chaos_model_amd", "spec")):
        if not d:
            continue
        try:
            os.makedirs(d, exist_ok=True)
            if os.access(d, os.W_OK):
                return d
        except OSError:
            pass
    raise RuntimeError("no writable cache directory for specialised kernels (set BNN_SPEC_CACHE)")


def _extra_flags():
    """BNN_SPEC_DEFINES="BNN_GEN_ABLATE=2 ...": measurement builds (part of the cache key; never set in production)."""
    return ["-D" + d for d in os.environ.get("BNN_SPEC_DEFINES", "").split()]


_cc_id = None


def _compiler_id():
    """First line of `hipcc --version` (a ROCm upgrade must not be served the old compiler's code objects)."""
    global _cc_id
    if _cc_id is None:
        try:
            _cc_id = subprocess.run([_build.hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout.strip().split("\n")[0]
        except Exception:
            _cc_id = "unknown"
    return _cc_id


def _key(src):
    h = hashlib.sha256(src.encode())
    h.update(" ".join(SPEC_FLAGS + _extra_flags()).encode())
    h.update(_compiler_id().encode())
    for name in _DEPS:   # the kernel source the generated file includes
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:20]


def compile_source(src, verbose=False):
    """hipcc --genco of one generated source -> (code object bytes, info) -- cached by the hash of source + kernel headers + flags.
    info = {"vgpr", "agpr", "scratch", "lds", "compile_s"}: the compiler's own resource report for the kernel."""
    import json
    import re
    import time
    path = os.path.join(cache_dir(), f"spec_{_key(src)}.hsaco")
    if not (os.path.exists(path) and os.path.exists(path + ".json")):
        cc = _build.hipcc()
        with tempfile.TemporaryDirectory() as td:
            sp = os.path.join(td, "spec.hip")
            with open(sp, "w") as f:
                f.write(src)
            tmp = os.path.join(td, "spec.hsaco")
            cmd = [cc] + SPEC_FLAGS + _extra_flags() + ["-Rpass-analysis=kernel-resource-usage", "-I", CSRC, "-I", INCLUDE, sp, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on the specialised kernel:\n" + r.stderr[-4000:])
            num = lambda k: int((re.search(k + r":\s*(\d+)", r.stderr) or [0, -1])[1])
            info = {"vgpr": num("VGPRs"), "agpr": num("AGPRs"), "scratch": num(r"ScratchSize \[bytes/lane\]"), "vgpr_spill": num("VGPRs Spill"),
                    "lds": num(r"LDS Size \[bytes/block\]"), "compile_s": round(time.time() - t0, 1)}
            _move(tmp, path)
            import uuid
            tmpj = path + f".{uuid.uuid4().hex}.json.part"
            with open(tmpj, "w") as f:
                json.dump(info, f)
            os.replace(tmpj, path + ".json")
    with open(path, "rb") as f:
        image = f.read()
    with open(path + ".json") as f:
        return image, json.load(f)


def _move(src, dst):
    import shutil
    import uuid
    part = dst + f".{uuid.uuid4().hex}.part"   # (unique per writer: threads of one process may compile the same form too)
    shutil.copyfile(src, part)
    os.replace(part, dst)   # atomic: concurrent ranks compiling the same form race to the same bytes


VARIANTS = (N.SPEC_POOL_REGS, 0, N.SPEC_BLOCK_MAJOR)   # tried in this order: the first whose code object needs no scratch is taken


def best_variant(arch, noisy, w8=None, verbose=False):
    """Compile the tuning variants of one form until one has no scratch; -> (image, info incl. "flags" and "w8").
    w8 = None searches the eight-wave forms first (two waves per SIMD won every same-box A/B where they fit without spilling:
    profiles/r04_spec_engine.jsonl), then the four-wave ones."""
    best = None
    for w in ((True, False) if w8 is None else (w8,)):
        for flags in VARIANTS:
            try:
                src = N.spec_source(arch, noisy, w, flags)
            except N.NativeError:
                if w8 is not None:
                    raise
                break          # eight waves' LDS does not fit next to this network's image
            image, info = compile_source(src, verbose=verbose)
            info = dict(info, flags=flags, w8=w)
            if info["scratch"] == 0:
                return image, info
            if best is None or info["scratch"] < best[1]["scratch"]:
                best = (image, info)
    return best


def prewarm(archs, noisy=(False, True), w8=(None,), jobs=None):
    """Compile the specialised forms of several networks into the cache, in parallel, WITHOUT a device (hipcc cross-compiles):
    archs = [(hidden, latent, depth_in, depth_out, n_features, fix_megno[, zero_mask]), ...] (the quiet forms are compiled for the
    column mask too; default: the pretrained ensemble's).  Returns [(arch, noisy, w8, info)]."""
    from concurrent.futures import ThreadPoolExecutor
    jobs_ = []
    from .ops import V50_ZERO_MASK
    for t in archs:
        H, L, din, dout, NF, megno = t[:6]
        mask = (t[6] if len(t) > 6 else V50_ZERO_MASK) | ((1 << 7) if megno else 0)
        a = N.BnnArch(NF, H, L, int(bool(megno)), mask, 0.5, 0.0, din, dout)
        for nz in noisy:
            for w in w8:
                if w is not None:
                    try:
                        N.spec_source(a, nz, w, N.SPEC_POOL_REGS)
                    except N.NativeError:
                        continue   # (a form this network cannot have, e.g. eight waves next to a large image)
                jobs_.append(((H, L, din, dout, NF, megno), a, nz, w))
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        infos = list(ex.map(lambda j: best_variant(j[1], j[2], j[3])[1], jobs_))
    return [(j[0], j[2], j[3], i) for j, i in zip(jobs_, infos)]


def specialize(plan, noisy=(False, True), w8=None, verbose=False):
    """Compile (or fetch from the cache) and attach the plan's specialised forms.  noisy: which forms -- forward(noisy_val=False) and
    forward_swag_fast use the quiet one, forward(noisy_val=True) the noisy one.  w8: None = the builder's choice of eight waves at 256
    registers vs four at 512; True / False to force (A/B).  Returns the plan (plan.spec_info[noisy] = the compiler's resource report)."""
    if isinstance(noisy, bool):
        noisy = (noisy,)
    todo = [bool(nz) for nz in noisy if not (plan.spec_attached(nz) and getattr(plan, "_spec_w8", {}).get(bool(nz), "unset") == w8)]
    if not todo:
        return plan
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(todo)) as ex:   # the forms compile side by side (hipcc subprocesses)
        built = list(ex.map(lambda nz: best_variant(plan.arch, nz, w8, verbose=verbose), todo))
    for nz, (image, info) in zip(todo, built):
        plan.attach_spec(image, nz, info["w8"], info["flags"])
        plan.__dict__.setdefault("_spec_w8", {})[nz] = w8
        plan.__dict__.setdefault("spec_info", {})[nz] = info
    return plan
