"""Run-time specialisation of the generic forward engine.

The reference builds its network from hparams (spock_reg_model.py:301-321, 346-362) and PyTorch runs whatever comes out at the same
speed.  Here the ahead-of-time generic engine reads the shapes from a descriptor: every trip count is a run-time number behind an
early exit, and the kernel is a nest of small basic blocks the compiler cannot schedule across.  `specialize(plan)` compiles the
SAME kernel source (csrc/bnn_generic.hip.h) for the plan's one network with every shape a compile-time constant -- a handful of
candidate forms (waves per workgroup x layer routine), ten to forty seconds of hipcc side by side, cached on disk --, times the
candidates on the GPU, keeps the fastest (remembered next to the code objects) and attaches it to the plan; from then on every entry
point that would take the generic route launches it.  Accumulation order and arithmetic are those of the ahead-of-time form: results
are bit-identical (tests/test_hip_spec.py); only the schedule changes.

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


def _private(d):
    """Code objects are loaded onto the GPU as they are found: the directory they come from must be ours alone (owned by this user, not
    writable by group or others)."""
    st = os.stat(d)
    return st.st_uid == os.getuid() and not (st.st_mode & 0o022)


def cache_dir():
    """BNN_SPEC_CACHE, else csrc/_spec next to the library (in-tree, like the built .so: it travels with the tree), else ~/.cache.
    Created with mode 0700; a directory somebody else owns or may write to is refused (BNN_SPEC_CACHE) or skipped (the defaults); a
    BNN_SPEC_CACHE that cannot be created or written raises instead of falling through to the defaults."""
    env = os.environ.get("BNN_SPEC_CACHE")
    for d in (env, os.path.join(CSRC, "_spec"), os.path.join(os.path.expanduser("~"), ".cache", "bnn_chaos_model_amd", "spec")):
        if not d:
            continue
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            if not _private(d):
                if d == env:
                    raise RuntimeError(f"BNN_SPEC_CACHE={d} is owned by another user or writable by group / others: code objects are loaded "
                                       "from there unchecked, so it must be private (chmod go-w, or point it somewhere else)")
                continue
            if os.access(d, os.W_OK):
                return d
            if d == env:
                raise RuntimeError(f"BNN_SPEC_CACHE={d} is not writable")
        except OSError as e:
            if d == env:   # an explicitly named cache that cannot be created is an error, never a silent fall-through to another one
                raise RuntimeError(f"BNN_SPEC_CACHE={d} cannot be created: {e}") from e
    raise RuntimeError("no writable private cache directory for specialised kernels (set BNN_SPEC_CACHE)")


def _extra_flags():
    """BNN_SPEC_DEFINES="BNN_GEN_ABLATE=2 ...": measurement builds (part of the cache key; never set in production)."""
    return ["-D" + d for d in os.environ.get("BNN_SPEC_DEFINES", "").split()]


_clean_env = _build.clean_env


_cc_id = None


def _compiler_id():
    """First line of `hipcc --version` (a ROCm upgrade must not be served the old compiler's code objects)."""
    global _cc_id
    if _cc_id is None:
        try:
            _cc_id = subprocess.run([_build.hipcc(), "--version"], capture_output=True, text=True, timeout=60, env=_clean_env()).stdout.strip().split("\n")[0]
        except Exception:
            _cc_id = "unknown"
    return _cc_id


def _library_id():
    """Identity of the LOADED library: a specialised kernel takes the library's GenParams block by value, so a code object compiled
    against other headers than the .so's own (a stale .so, an A/B build under BNN_CHAOS_SO) would read it with another layout -- a GPU
    fault, not an error code.  ABI version + sizeof(GenParams) + build flags as the library reports them; and, unless the library is the
    in-tree one built from exactly the sources next to it (then the headers hashed below ARE its headers), its own source hash / path."""
    L = N.lib()
    try:
        with open(N.SO_PATH + ".srchash") as f:
            srchash = f.read().strip()
    except OSError:
        srchash = "no-srchash"
    in_tree = not os.environ.get("BNN_CHAOS_SO") and srchash == _build.source_hash()
    origin = "in-tree" if in_tree else f"{os.path.abspath(N.SO_PATH)}:{srchash}"
    return f"abi{L.bnn_abi_version()}:genparams{L.bnn_gen_params_bytes()}:{L.bnn_build_flags().decode()}:{origin}"


def _hash_deps(h):
    h.update(_compiler_id().encode())
    h.update(_library_id().encode())
    for name in _DEPS:   # the kernel source the generated file includes
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())


def _key(src):
    h = hashlib.sha256(src.encode())
    h.update(" ".join(SPEC_FLAGS + _extra_flags()).encode())
    _hash_deps(h)
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
            r = subprocess.run(cmd, capture_output=True, text=True, env=_clean_env())
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


VARIANTS = (N.SPEC_POOL_REGS, N.SPEC_BLOCK_MAJOR)   # the candidates per wave count.  Not searched: flags 0 (input-quad-major layers with the
# pool in LDS rows) never beat POOL_REGS where both compiled; POOL_REGS | RESIDENT (feature_nn's weights in VGPRs across the tiles)
# ties the streamed form on the pretrained shapes and loses elsewhere (profiles/r04_spec_tuning.jsonl, candidates with flags 5).
# BNN_SPEC_FORCE_FLAGS=<n> builds any of them. (flags 0 -- input-quad-major layers with the pool in
                                                     # LDS rows -- never beat POOL_REGS where both compiled: BNN_SPEC_FORCE_FLAGS=0 still builds it)


def candidates(arch, noisy, w8=None, verbose=False, jobs=None):
    """Every (waves, variant) form of the network that the builder accepts, compiled side by side -> [(image, info)], info incl. "flags",
    "w8" (True / False) and the compiler's resource report.  BNN_SPEC_FORCE_FLAGS=<n>: that variant only (measurements)."""
    from concurrent.futures import ThreadPoolExecutor
    variants = VARIANTS
    if os.environ.get("BNN_SPEC_FORCE_FLAGS", "") != "":
        variants = (int(os.environ["BNN_SPEC_FORCE_FLAGS"]),)
    todo, refused = [], None
    for w in ((True, False) if w8 is None else (w8,)):
        for flags in variants:
            try:
                todo.append((w, flags, N.spec_source(arch, noisy, w, flags)))
            except N.NativeError as e:   # (a form this network cannot have, e.g. eight waves next to a large image)
                refused = e
    if not todo:
        raise refused
    uniq = {}
    for w, flags, src in todo:      # (different requests can come out as the same text, e.g. when only four waves fit either way)
        uniq.setdefault(src, (w, flags))
    with ThreadPoolExecutor(max_workers=jobs or min(8, len(uniq))) as ex:
        built = list(ex.map(lambda src: compile_source(src, verbose=verbose), uniq))
    def nwaves(src):   # waves per workgroup the builder settled on (the 17th field of the GenArch initializer)
        import re
        return int(re.search(r"constexpr GenArch value = \{\s*([^{]*)\{", src).group(1).split(",")[16])

    return [(image, dict(info, flags=uniq[src][1], w8=uniq[src][0], nwaves=nwaves(src))) for src, (image, info) in zip(uniq, built)]


def rank_static(cands):
    """Without a GPU to time them on (profiles/r04_spec_tuning.jsonl is what this order was read from): forms that put at least one wave
    on every SIMD first (hidden 128: the only scratch-free form runs ONE wave per CU and is 4x slower than a spilling four-wave one);
    among those, no scratch before scratch (82 features, noisy: eight spilling waves are 2.4x slower than four clean ones); then more
    waves per CU; then the VARIANTS order."""
    def waves_per_cu(info):
        nw = info.get("nwaves") or (16 if (info["w8"] == 2 and info["w8"] is not True) else 8 if info["w8"] else 4)
        # a workgroup of at most four waves that leaves half the LDS and half the registers free runs two to a CU
        return 2 * nw if (nw <= 4 and info["lds"] <= 80 * 1024 and info["vgpr"] + max(info["agpr"], 0) <= 256) else nw

    def key(c):
        info = c[1]
        return (waves_per_cu(info) < 4, info["scratch"] > 0, -waves_per_cu(info), info["scratch"],
                VARIANTS.index(info["flags"]) if info["flags"] in VARIANTS else 9)
    return sorted(cands, key=key)


def best_variant(arch, noisy, w8=None, verbose=False, measure=None):
    """-> (image, info) of the form to attach.  measure: callable(image, info) -> seconds (specialize() passes one when a GPU is there:
    every candidate is timed on a small synthetic grid and the fastest wins -- results are bit-identical across candidates, only the
    speed differs); None: rank_static."""
    cands = rank_static(candidates(arch, noisy, w8, verbose=verbose))
    if measure is None or len(cands) == 1:
        return cands[0]
    timed = []
    for image, info in cands:
        try:
            timed.append((measure(image, info), image, info))
        except Exception as e:   # a candidate that does not load or launch is not a candidate
            if verbose:
                print("candidate failed:", info, e)
    if not timed:
        raise RuntimeError("none of the network's specialised forms could be loaded and timed on this GPU "
                           f"({len(cands)} candidate(s)); the plan keeps its ahead-of-time form")
    fastest = min(t[0] for t in timed)
    timed.sort(key=lambda t: (t[0] > 1.03 * fastest, [id(c[0]) for c in cands].index(id(t[1]))))   # within 3 % of the fastest: the static order decides
    info = dict(timed[0][2], tuned_ms=round(timed[0][0] * 1e3, 3),
                candidates=[{"w8": i["w8"], "flags": i["flags"], "scratch": i["scratch"], "ms": round(t * 1e3, 3)} for t, _, i in timed])
    return timed[0][1], info


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


def _measure_on(live_plan, nz):
    """Times one candidate on a synthetic grid shaped like BASELINE configs[1] cut to 96 draws (10 000 systems x 100 timesteps:
    1 920 workgroups of up to 512 systems, the last of each draw ragged; in-kernel Philox): seconds.
    The candidates are attached to a PRIVATE plan of the same network on the same device: the live plan is shared by every model and
    thread of the process (ops._plans) and keeps launching whatever it had while the tuning runs; only the winner is swapped in."""
    import torch
    from . import ops
    a = live_plan.arch
    plan = N.Plan(int(a.zero_mask), float(a.lowest_std), n_features=a.n_features, hidden=a.hidden, latent=a.latent, fix_megno=bool(a.fix_megno),
                  depth_in=a.depth_in, depth_out=a.depth_out)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(10000, 100, plan.n_features, generator=g, device="cuda")
    W = torch.randn(96, plan.d, generator=g, device="cuda") * 0.1

    def run(image, info):
        plan.attach_spec(image, nz, info["w8"], info["flags"])
        best = float("inf")
        for rep in range(3):   # (the first repetition warms up: code upload, clocks); eight launches back to back per timing
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(8):
                ops.forward(x, W, philox_seed=1, plan=plan, noisy=nz, engine="spec", assume_finite=True)
            t1.record()
            t1.synchronize()
            if rep:
                best = min(best, t0.elapsed_time(t1) * 1e-3 / 8)
        return best
    return run


def _choice_path(arch, nz, w8):
    import ctypes as C
    h = hashlib.sha256(bytes(C.string_at(C.addressof(arch), C.sizeof(arch))) + repr((bool(nz), w8)).encode())
    _hash_deps(h)
    return os.path.join(cache_dir(), f"choice_{h.hexdigest()[:20]}.json")


def specialize(plan, noisy=(False, True), w8=None, verbose=False, tune=True):
    """Compile (or fetch from the cache) and attach the plan's specialised forms.  noisy: which forms -- forward(noisy_val=False) and
    forward_swag_fast use the quiet one, forward(noisy_val=True) the noisy one.  w8: None = every wave count the builder accepts;
    True / False / 2 to force eight / at most four / sixteen waves (A/B).  tune: time the candidates on this GPU and keep the fastest
    (the decision is remembered next to the code objects); False: the static ranking.  Returns the plan (plan.spec_info[noisy] = the
    compiler's resource report of the attached form, with the candidates' timings when it was tuned)."""
    import json
    if isinstance(noisy, bool):
        noisy = (noisy,)
    todo = [bool(nz) for nz in noisy if not (plan.spec_attached(nz) and getattr(plan, "_spec_w8", {}).get(bool(nz), "unset") == w8)]
    if not todo:
        return plan
    from concurrent.futures import ThreadPoolExecutor

    def one(nz):   # a decision remembered from an earlier run on this machine: compile (or fetch) just that form
        cp = _choice_path(plan.arch, nz, w8)
        if tune and os.path.exists(cp) and os.environ.get("BNN_SPEC_FORCE_FLAGS", "") == "":
            with open(cp) as f:
                want = json.load(f)
            try:
                image, info = compile_source(N.spec_source(plan.arch, nz, want["w8"], want["flags"]), verbose=verbose)
            except N.NativeError:
                return None
            return image, dict(info, w8=want["w8"], flags=want["flags"], tuned_ms=want.get("tuned_ms"), candidates=want.get("candidates"))
        return None

    with ThreadPoolExecutor(max_workers=len(todo)) as ex:   # the forms compile side by side (hipcc subprocesses)
        remembered = list(ex.map(one, todo))
        cands = list(ex.map(lambda nz: None if remembered[todo.index(nz)] else rank_static(candidates(plan.arch, nz, w8, verbose=verbose)), todo))
    for nz, rem, cs in zip(todo, remembered, cands):
        if rem is not None:
            image, info = rem
        elif tune and len(cs) > 1:
            image, info = best_variant(plan.arch, nz, w8, verbose=verbose, measure=_measure_on(plan, nz))
            if os.environ.get("BNN_SPEC_FORCE_FLAGS", "") == "":
                tmp = _choice_path(plan.arch, nz, w8) + f".{os.getpid()}.part"
                with open(tmp, "w") as f:
                    json.dump({"w8": info["w8"], "flags": info["flags"], "tuned_ms": info.get("tuned_ms"), "candidates": info.get("candidates")}, f)
                os.replace(tmp, _choice_path(plan.arch, nz, w8))
        else:
            image, info = cs[0]
        plan.attach_spec(image, nz, info["w8"], info["flags"])
        plan.__dict__.setdefault("_spec_w8", {})[nz] = w8
        plan.__dict__.setdefault("spec_info", {})[nz] = info
    return plan
