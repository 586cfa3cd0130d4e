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
    d = os.environ.get("BNN_SPEC_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "bnn_chaos_model_amd", "spec")
    os.makedirs(d, exist_ok=True)
    return d


def _key(src):
    h = hashlib.sha256(src.encode())
    h.update(" ".join(SPEC_FLAGS).encode())
    for name in _DEPS:   # the kernel source the generated file includes
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:20]


def compile_source(src, verbose=False):
    """hipcc --genco of one generated source -> code object bytes (cached by the hash of source + kernel headers + flags)."""
    path = os.path.join(cache_dir(), f"spec_{_key(src)}.hsaco")
    if not os.path.exists(path):
        cc = _build.hipcc()
        with tempfile.TemporaryDirectory() as td:
            sp = os.path.join(td, "spec.hip")
            with open(sp, "w") as f:
                f.write(src)
            tmp = os.path.join(td, "spec.hsaco")
            cmd = [cc] + SPEC_FLAGS + ["-I", CSRC, "-I", INCLUDE, sp, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on the specialised kernel:\n" + r.stderr[-4000:])
            _move(tmp, path)
    with open(path, "rb") as f:
        return f.read()


def _move(src, dst):
    import shutil
    part = dst + f".{os.getpid()}.part"
    shutil.copyfile(src, part)
    os.replace(part, dst)   # atomic: concurrent ranks compiling the same form race to the same bytes


def specialize(plan, noisy=(False, True), w8=None, verbose=False):
    """Compile (or fetch from the cache) and attach the plan's specialised forms.  noisy: which forms -- forward(noisy_val=False) and
    forward_swag_fast use the quiet one, forward(noisy_val=True) the noisy one.  w8: None = the builder's choice of eight waves at 256
    registers vs four at 512; True / False to force (A/B).  Returns the plan."""
    if isinstance(noisy, bool):
        noisy = (noisy,)
    for nz in noisy:
        if plan.spec_attached(nz) and getattr(plan, "_spec_w8", {}).get(nz, "unset") == w8:
            continue
        image = compile_source(plan.spec_source(nz, w8), verbose=verbose)
        plan.attach_spec(image, nz, w8)
        plan.__dict__.setdefault("_spec_w8", {})[nz] = w8
    return plan
