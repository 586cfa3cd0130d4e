"""Tensor-level wrappers over the C ABI (include/bnn_chaos_hip.h).  Inputs/outputs are torch
tensors on the GPU; work is enqueued on torch's current HIP stream; nothing here synchronises.

Each function names the reference method it replaces (file:line in MilesCranmer/bnn_chaos_model).
"""
import ctypes as C

import torch

from . import _native as N

D = 7583
LATENT = 20
V50_ZERO_MASK = sum(1 << c for c in (7, 3, 6, 38, 39, 40, 1, 2, 4, 5))

_plans = {}


def _on_device_of(argname_index=0):
    """Run the wrapped op with the device of its first tensor argument current: torch's stream, the plan's tables and the
    kernel launch must all belong to the same GPU (one process may drive several)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            t = args[argname_index] if len(args) > argname_index else None
            if isinstance(t, torch.Tensor) and t.is_cuda and t.device.index != N.current_device():
                with torch.cuda.device(t.device):
                    return fn(*args, **kwargs)
            return fn(*args, **kwargs)
        return wrapper
    return deco


def zero_mask_from_flags(fix_megno=False, fix_megno2=True, include_mmr=False, include_nan=False,
                         include_eplusminus=False):
    """Columns zeroed by zero_megno / zero_mmr / zero_nan / zero_eplusminus (spock_reg_model.py:452-500)."""
    cols = []
    if fix_megno or fix_megno2:
        cols += [7]
    if not include_mmr:
        cols += [3, 6]
    if not include_nan:
        cols += [38, 39, 40]
    if not include_eplusminus:
        cols += [1, 2, 4, 5]
    return sum(1 << c for c in set(cols))


def get_plan(zero_mask=V50_ZERO_MASK, lowest_std=0.5, device=None, fix_megno=False, n_features=41, hidden=40, latent=20,
             depth_in=1, depth_out=1):
    """One plan per (device, network, column mask).  n_features / hidden / latent / depth_in / depth_out describe the network the
    reference builds from hparams (spock_reg_model.py:301-321, 346-362; depth = hparams['in'] / ['out']); the defaults are the
    pretrained ensemble's.  fix_megno: hparams['fix_megno'] (:360-362): summary two wider; the mask must zero column 7."""
    dev = N.current_device() if device is None else torch.device(device).index
    key = (dev, int(zero_mask), float(lowest_std), bool(fix_megno), int(n_features), int(hidden), int(latent), int(depth_in), int(depth_out))
    if key not in _plans:
        with torch.cuda.device(dev):
            _plans[key] = N.Plan(int(zero_mask), float(lowest_std), n_features=n_features, hidden=hidden, latent=latent,
                                 fix_megno=fix_megno, depth_in=depth_in, depth_out=depth_out)
    return _plans[key]


def specialize(plan=None, noisy=(False, True), w8=None, verbose=False):
    """Compile this plan's network into its own form of the generic engine (specialize.py: ~10 s of hipcc per form, cached on disk)
    and attach it; bit-identical results, the pretrained network's schedule quality for any hparams-built network."""
    from . import specialize as S
    plan = plan or get_plan()
    with torch.cuda.device(plan_device(plan)):
        return S.specialize(plan, noisy=noisy, w8=w8, verbose=verbose)


def plan_device(plan):
    for k, v in _plans.items():
        if v is plan:
            return k[0]
    return torch.cuda.current_device()


def _check_x(x, plan):
    if x.dim() != 3 or x.shape[2] != plan.n_features:
        raise NotImplementedError(f"x must be [B, T, {plan.n_features}]")  # figures/spock/regression.py:210-211
    return x


def fused_draw_available(plan, T, K):
    """The in-prologue draw (one launch, no workspace) exists in the pretrained network's kernels only: T % 4 == 0, T >= 8, K <= 32."""
    return plan.v50net and T % 4 == 0 and T >= 8 and K <= 32


def _f32(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a float32 GPU tensor, got {t.dtype} on {t.device}")
    return t.contiguous()


ENGINES = {"auto": 0, "generic": 1, "spec": 2}


def _grid(B, T, J, nchunks, spb, noisy=False, engine="auto", chunk_B=0, chunk_off=0, nonfinite=None):
    """chunk_B / chunk_off: the batch is sharded over devices and this call holds rows [chunk_off, chunk_off + B) of its chunk_B
    systems; the draws' chunks (torch.chunk) partition the whole batch (bnn_grid in include/bnn_chaos_hip.h).
    nonfinite: the scan record of this call's x (nonfinite_scan) or None = x is assumed finite."""
    g = N.BnnGrid(int(B), int(T), int(J), int(nchunks), int(spb), int(bool(noisy)), ENGINES[engine], int(chunk_B or 0), int(chunk_off),
                  None if nonfinite is None else nonfinite.data_ptr())
    g._keepalive = nonfinite
    return g


def small_grid(plan, B, T, J, nchunks=1, chunk_B=0, systems_per_block=0, engine="auto"):
    """True where the library runs the pretrained network's kernel in its TILE-SPLIT form (bnn_abi.hip: launch_forward, TSPLIT_MAX_BLOCKS):
    the v50 mask, whole 4-step tiles, no explicit block size, and at most 256 blocks of 16 systems in the whole grid -- the evaluation
    scripts' per-chunk calls (15 .. 3 000 rows under one draw) -- or draws that cover at most 16 systems each, however many (the 5-planet
    loop as one call)."""
    if not (plan.v50net and not plan.fix_megno and plan.arch.zero_mask == V50_ZERO_MASK and T % 4 == 0 and T >= 8 and systems_per_block == 0
            and engine == "auto" and B > 0 and J > 0):
        return False
    csz = -(-(chunk_B or B) // max(nchunks, 1))
    return -(-min(csz, B) // 16) * J <= 256 or min(csz, B) <= 16


def _scan_st(x, plan, assume_finite, nonfinite):
    """(record or None, stream handle) for a call on x: the stream handle is looked up once per op (torch.cuda.current_stream() costs
    microseconds) and shared by the scan and the launch behind it."""
    st = N.stream_ptr(x.device.index)
    if nonfinite is not None:
        return _nonfinite_record(x, plan, assume_finite, nonfinite), st
    if assume_finite:
        return None, st
    B, T, _ = x.shape
    rec = _record_buffer(B, x.device, st)
    N.check(N.lib().bnn_nonfinite_scan_f32(plan.handle, x.data_ptr(), B, T, rec.data_ptr(), st))
    return rec, st


_records = {}          # (device index, stream handle) -> int32 buffer, grow-only: the scan records of eager calls
_record_views = {}     # the same key -> {B: buffer[:4 + B]}
_RECORDS_PER_DEVICE = 8


def _record_buffer(B, device, stream_handle):
    """Where the default route keeps the scan record [4 + B] of ONE call.

    Eager calls: a buffer owned by this module per (device, stream), sized on first use and grown when a larger batch arrives -- no
    allocator call per op once it exists.  Calls on one stream are serialised by the stream (the next call's header reset is ordered
    behind the previous call's fix-up); calls on different streams get different buffers, so concurrent streams never share a record.
    Under stream capture: a block of the capturing graph's memory pool (torch.empty during capture), because its lifetime must be the
    GRAPH's -- the pointer is baked into the captured kernel nodes, and two graphs replayed on different streams must not share a
    record.  A caller who wants the record out of every allocator passes `nonfinite=` (a tensor they own, filled by nonfinite_scan(x,
    out=...) inside the captured region)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((4 + B,), dtype=torch.int32, device=device)
    key = (device.index, stream_handle)
    ent = _records.get(key)
    if ent is None or ent.numel() < 4 + B:
        if ent is None and sum(1 for k in _records if k[0] == device.index) >= _RECORDS_PER_DEVICE:
            old = next(k for k in _records if k[0] == device.index)    # oldest stream of this device (freeing is stream-ordered: safe)
            _records.pop(old)
            _record_views.pop(old, None)
        ent = torch.empty((max(4 + B, 4 + 1024),), dtype=torch.int32, device=device)
        _records[key] = ent
        _record_views[key] = {}
    views = _record_views[key]
    v = views.get(B)
    if v is None:
        if len(views) > 64:
            views.clear()
        v = views[B] = ent[:4 + B]      # (a view per batch size, made once: slicing a tensor costs microseconds)
    return v


@_on_device_of(0)
def nonfinite_scan(x, plan=None, out=None):
    """One streaming pass over x [B,T,F] -> the record (int32 [4 + B], on x's device) of the systems that hold NaN / +-inf anywhere:
    [0] = how many, [1] = of which certainly NaN whatever the weights (a NaN anywhere, or +-inf in a masked column: the reference's
    `x - mask`, spock_reg_model.py:452-478), [4 + i] = (system << 1) | certain.  Hand it to any op below as `nonfinite=` -- the listed
    systems then get what the reference returns (NaN, or the exact IEEE evaluation where an infinity dies in a ReLU) -- or let the op
    make it (the default; `assume_finite=True` skips it).  The record can serve any number of calls on the same x.
    out: a caller-owned int32 [4 + B] tensor to fill (e.g. inside a captured HIP graph); default: a fresh tensor."""
    plan = plan or get_plan()
    x = _f32(x, "x")
    _check_x(x, plan)
    B, T, _ = x.shape
    if out is None:
        rec = torch.empty((4 + B,), dtype=torch.int32, device=x.device)
    else:
        rec = _nonfinite_record(x, plan, False, out)
    N.check(N.lib().bnn_nonfinite_scan_f32(plan.handle, N.ptr(x), B, T, N.ptr(rec), N.stream_ptr()))
    return rec


def _scan_into_own_record(x, plan):
    """The default route's scan: the record lives in this module's per-(device, stream) buffer (_record_buffer)."""
    B, T, _ = x.shape
    st = N.stream_ptr(x.device.index)
    rec = _record_buffer(B, x.device, st)
    N.check(N.lib().bnn_nonfinite_scan_f32(plan.handle, x.data_ptr(), B, T, rec.data_ptr(), st))
    return rec


def _nonfinite_record(x, plan, assume_finite, nonfinite):
    if nonfinite is not None:
        if nonfinite.dtype != torch.int32 or nonfinite.numel() != 4 + x.shape[0] or nonfinite.device != x.device or not nonfinite.is_contiguous():
            raise ValueError(f"nonfinite must be the int32 record [4 + {x.shape[0]}] of nonfinite_scan(x) on x's device")
        return nonfinite
    return None if assume_finite else _scan_into_own_record(x, plan)


@_on_device_of(0)
def swag_draw(w_avg, w2_avg, pre_D, seed_idx, z1=None, z2=None, scale=0.5, philox_seed=0, draw_id0=0, plan=None):
    """SWAGModel.sample_weights (spock_reg_model.py:815-838) for J draws -> W[J, d].

    w_avg, w2_avg [S,d]; pre_D [S,d,K]; seed_idx [J] int32; z1 [J,d] / z2 [J,K] explicit normals or None (Philox)."""
    plan = plan or get_plan()
    w_avg, w2_avg, pre_D, S, d, K = _ensemble(plan, w_avg, w2_avg, pre_D)
    seed_idx = seed_idx.to(device=w_avg.device, dtype=torch.int32).contiguous()
    J = seed_idx.numel()
    z1, z2 = _f32(z1, "z1"), _f32(z2, "z2")
    if (z1 is None) != (z2 is None):
        raise ValueError("z1 and z2 must both be given or both be None")
    if z1 is not None and (tuple(z1.shape) != (J, d) or tuple(z2.shape) != (J, K)):
        raise ValueError("z1 must be [J,d] and z2 [J,K]")
    W = torch.empty((J, d), dtype=torch.float32, device=w_avg.device)
    N.check(N.lib().bnn_swag_draw_f32(plan.handle, N.ptr(w_avg), N.ptr(w2_avg), N.ptr(pre_D), S, K, N.ptr(seed_idx), J,
                                      N.ptr(z1), N.ptr(z2), float(scale), int(philox_seed), int(draw_id0), N.ptr(W),
                                      N.stream_ptr()))
    return W


PRECISIONS = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3, "f16": 4, "f16x3": 5}
HALF_FORMS = ("f16", "f16x3")
HALF_MAX = 65504.0


def half_range_exceeded(x, zero_mask=V50_ZERO_MASK):
    """Rows of x [B,T,41] that the IEEE-half forms ("f16", "f16x3") cannot represent: a live column holds |v| >= 65 504.  Those
    forms SATURATE such values (MODE.FP16_OVFL), so the affected systems get finite but wrong outputs -- e.g. the 5-planet script's
    constant-4 fill of unstable systems standardises the mass columns to 1.9e5 (figures/multiswag_5_planet.py:215).  Returns a bool
    tensor [B] on x's device (no host sync; `.any().item()` is the caller's).  The bfloat16 forms have fp32's range."""
    live = [c for c in range(41) if not (int(zero_mask) >> c) & 1]
    hi, lo = torch.amax(x, dim=1), torch.amin(x, dim=1)            # [B,41] each: two passes over x, no x-sized temporary
    return (torch.maximum(hi, -lo)[:, live] >= HALF_MAX).any(1)


@_on_device_of(0)
def forward(x, W, eps=None, eps_in=None, eps_sum=None, nchunks=1, philox_seed=0, draw_id0=0, system_id0=0, plan=None,
            debug=False, systems_per_block=0, noisy=False, precision="f32", engine="auto", chunk_B=0, chunk_off=0, assume_finite=False,
            nonfinite=None):
    """VarModel.forward (spock_reg_model.py:486-528) for materialised weight vectors W[J,d] -> out[J/nchunks,B,2].

    Non-finite inputs: by default x is scanned once (nonfinite_scan: one streaming pass) and the systems holding NaN / +-inf get the
    reference's result (`x - mask`, :452-478; NaN-propagating nn.ReLU); assume_finite=True skips the scan (x is known to be clean:
    the kernels alone are exact on finite data only), nonfinite= takes a record made earlier for the same x.

    eps [R,B,2,20] = the two randn_like of compute_summary_stats (:426-427) or None (Philox);
    eps_in [R,B,T,41] + eps_sum [R,B,40] switch on noisy_val=True (:444-450); noisy=True with no noise tensors at all
    is noisy_val=True with every normal generated in-kernel (Philox).
    precision: "f32" (default: the parity path) or the OPT-IN reduced-precision forms "bf16" / "bf16x3" / "bf16x6" / "f16" /
    "f16x3" (feature_nn on the bf16 / half matrix pipe; BASELINE configs[4] sweep; v50 mask, quiet forward only).
    engine: "auto" = the pretrained network at T % 4 == 0, T >= 8 runs on its register-resident kernels, every other shape on the
    generic engine (in the network's run-time-compiled form once `specialize(plan)` has attached one); "generic" forces the generic
    engine's ahead-of-time form (cross-checks, measurements); "spec" insists on the specialised form.
    PRECONDITION of "f16" / "f16x3": |x| < 65 504 in the live columns (half_range_exceeded(x) names the rows that violate it;
    their outputs are finite but wrong).  This op never synchronises, so it does not check; the FeatureRegressor surface does."""
    plan = plan or get_plan()
    x, W = _f32(x, "x"), _f32(W, "W")
    _check_x(x, plan)
    B, T, NF = x.shape
    LAT = plan.latent
    if W.dim() != 2 or W.shape[1] != plan.d:
        raise ValueError(f"W must be [J,{plan.d}]")
    J = W.shape[0]
    _check_grid(J, nchunks, draw_id0)
    R = J // nchunks
    eps, eps_in, eps_sum = _f32(eps, "eps"), _f32(eps_in, "eps_in"), _f32(eps_sum, "eps_sum")
    _check_same_device(x, W=W, eps=eps, eps_in=eps_in, eps_sum=eps_sum)
    if eps is not None and tuple(eps.shape) != (R, B, 2, LAT):
        raise ValueError(f"eps must be [{R},{B},2,{LAT}]")
    if (eps_in is None) != (eps_sum is None):
        raise ValueError("eps_in and eps_sum must both be given or both be None")
    SM = plan.summary_width
    if eps_in is not None and (tuple(eps_in.shape) != (R, B, T, NF) or tuple(eps_sum.shape) != (R, B, SM)):
        raise ValueError(f"eps_in must be [{R},{B},{T},{NF}] and eps_sum [{R},{B},{SM}]")
    out = torch.empty((R, B, 2), dtype=torch.float32, device=x.device)
    pre = torch.empty_like(out) if debug else None
    summ = torch.empty((R, B, SM), dtype=torch.float32, device=x.device) if debug else None
    g = _grid(B, T, J, nchunks, systems_per_block, noisy, engine, chunk_B, chunk_off, _nonfinite_record(x, plan, assume_finite, nonfinite))
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    if precision != "f32":
        if noisy or eps_in is not None:
            raise NotImplementedError("the reduced-precision kernels have no noisy form")
        N.check(N.lib().bnn_forward_lowp_f32(plan.handle, C.byref(g), N.ptr(x), N.ptr(W), N.ptr(eps), int(philox_seed), int(draw_id0),
                                             int(system_id0), PRECISIONS[precision], N.ptr(out), N.ptr(pre), N.ptr(summ), N.stream_ptr()))
        return (out, pre, summ) if debug else out
    N.check(N.lib().bnn_forward_f32(plan.handle, C.byref(g), N.ptr(x), N.ptr(W), N.ptr(eps), N.ptr(eps_in), N.ptr(eps_sum),
                                    int(philox_seed), int(draw_id0), int(system_id0), N.ptr(out), N.ptr(pre), N.ptr(summ),
                                    N.stream_ptr()))
    return (out, pre, summ) if debug else out


@_on_device_of(0)
def feature_latents(x, W, eps_in=None, noisy=False, philox_seed=0, draw_id0=0, system_id0=0, plan=None, assume_finite=False, nonfinite=None):
    """feature_nn alone -> the per-timestep latents [J, B, T, latent] that compute_summary_stats leaves in self.latents
    (spock_reg_model.py:417, 433): masks applied, optional input noise (explicit eps_in [J,B,T,F], or noisy=True for in-kernel Philox)."""
    plan = plan or get_plan()
    x, W, eps_in = _f32(x, "x"), _f32(W, "W"), _f32(eps_in, "eps_in")
    _check_x(x, plan)
    B, T, NF = x.shape
    if W.dim() != 2 or W.shape[1] != plan.d:
        raise ValueError(f"W must be [J,{plan.d}]")
    J = W.shape[0]
    if eps_in is not None and tuple(eps_in.shape) != (J, B, T, NF):
        raise ValueError(f"eps_in must be [{J},{B},{T},{NF}]")
    lat = torch.empty((J, B, T, plan.latent), dtype=torch.float32, device=x.device)
    g = _grid(B, T, J, 1, 0, noisy, nonfinite=_nonfinite_record(x, plan, assume_finite, nonfinite))
    N.check(N.lib().bnn_feature_nn_f32(plan.handle, C.byref(g), N.ptr(x), N.ptr(W), N.ptr(eps_in), int(philox_seed), int(draw_id0), int(system_id0),
                                       N.ptr(lat), N.stream_ptr()))
    return lat


def _check_grid(J, nchunks, draw_id0):
    if nchunks < 1 or J % nchunks:
        raise ValueError("the number of draws must be a multiple of nchunks")
    if draw_id0 % nchunks:
        raise ValueError("draw_id0 must be a multiple of nchunks")


def _check_same_device(x, **tensors):
    for name, t in tensors.items():
        if t is not None and t.device != x.device:
            raise ValueError(f"{name} is on {t.device}, x on {x.device}")


def _ensemble(plan, w_avg, w2_avg, pre_D):
    w_avg, w2_avg, pre_D = _f32(w_avg, "w_avg"), _f32(w2_avg, "w2_avg"), _f32(pre_D, "pre_D")
    if pre_D.dim() != 3:
        raise ValueError("pre_D must be [S,d,K]")
    S, d, K = pre_D.shape
    if d != plan.d or tuple(w_avg.shape) != (S, d) or tuple(w2_avg.shape) != (S, d):
        raise ValueError(f"ensemble tensors must be w_avg, w2_avg [S,{plan.d}] and pre_D [S,{plan.d},K]")
    return w_avg, w2_avg, pre_D, S, d, K


def _workspace(J, d, device):
    """[J,d] scratch for the draws of ONE call.  torch's caching allocator is stream-ordered, so a fresh block per call is
    cheap, safe when several streams run the op concurrently, and a captured HIP graph keeps its own block alive."""
    return torch.empty((J, d), dtype=torch.float32, device=device)


@_on_device_of(0)
def multiswag(x, w_avg, w2_avg, pre_D, seed_idx, z1=None, z2=None, eps=None, nchunks=1, scale=0.5, philox_seed=0,
              draw_id0=0, system_id0=0, plan=None, debug=False, systems_per_block=0, out=None, single_launch=None, precision="f32",
              engine="auto", chunk_B=0, chunk_off=0, assume_finite=False, nonfinite=None):
    """Fused SWAGModel.forward_swag_fast (spock_reg_model.py:878-908) over the MC loop of
    figures/multiswag_5_planet.py:295-298 -> out[J/nchunks, B, 2].

    single_launch: True = every workgroup samples its draw in its prologue (no scratch memory); False = draws are
    sampled once into a [J,d] workspace allocated for this call and read by the forward kernel of the same call (same
    bits, faster when a draw serves many workgroups).  None = choose by chunk size.
    assume_finite / nonfinite: see forward()."""
    plan = plan or get_plan()
    if precision != "f32":  # opt-in reduced precision: exact fp32 draw, then the bf16-pipe forward (same noise streams)
        W = swag_draw(w_avg, w2_avg, pre_D, seed_idx, z1, z2, scale=scale, philox_seed=philox_seed, draw_id0=draw_id0, plan=plan)
        res = forward(x, W, eps=eps, nchunks=nchunks, philox_seed=philox_seed, draw_id0=draw_id0, system_id0=system_id0, plan=plan,
                      debug=debug, systems_per_block=systems_per_block, precision=precision, chunk_B=chunk_B, chunk_off=chunk_off,
                      assume_finite=assume_finite, nonfinite=nonfinite)
        if out is not None and not debug:
            out.copy_(res)
            return out
        return res
    x = _f32(x, "x")
    _check_x(x, plan)
    w_avg, w2_avg, pre_D, S, d, K = _ensemble(plan, w_avg, w2_avg, pre_D)
    B, T, _ = x.shape
    if seed_idx.device != x.device or seed_idx.dtype != torch.int32 or not seed_idx.is_contiguous():
        seed_idx = seed_idx.to(device=x.device, dtype=torch.int32).contiguous()
    J = seed_idx.numel()
    _check_grid(J, nchunks, draw_id0)
    R = J // nchunks
    z1, z2, eps = _f32(z1, "z1"), _f32(z2, "z2"), _f32(eps, "eps")
    _check_same_device(x, w_avg=w_avg, w2_avg=w2_avg, pre_D=pre_D, z1=z1, z2=z2, eps=eps)
    if (z1 is None) != (z2 is None):
        raise ValueError("z1 and z2 must both be given or both be None")
    if z1 is not None and (tuple(z1.shape) != (J, d) or tuple(z2.shape) != (J, K)):
        raise ValueError("z1 must be [J,d] and z2 [J,K]")
    if eps is not None and tuple(eps.shape) != (R, B, 2, plan.latent):
        raise ValueError(f"eps must be [{R},{B},2,{plan.latent}]")
    if out is None:
        out = torch.empty((R, B, 2), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (R, B, 2) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError(f"out must be a contiguous float32 [{R},{B},2] tensor on x's device")
    pre = torch.empty_like(out) if debug else None
    summ = torch.empty((R, B, plan.summary_width), dtype=torch.float32, device=x.device) if debug else None
    small = small_grid(plan, B, T, J, nchunks, chunk_B, systems_per_block, engine)
    rec, st = _scan_st(x, plan, assume_finite, nonfinite)
    g = _grid(B, T, J, nchunks, systems_per_block, engine=engine, chunk_B=chunk_B, chunk_off=chunk_off, nonfinite=rec)
    if not fused_draw_available(plan, T, K) or engine != "auto":
        if single_launch:
            raise NotImplementedError("single_launch (in-prologue draw) exists for the pretrained network at T % 4 == 0, K <= 32 only")
        single_launch = False
    if single_launch is None:
        # Few systems per draw: every workgroup samples its draw in its prologue (no workspace, no draw launch).  In the tile-split form of
        # the small grids a draw shared by MANY workgroups is cheaper drawn once: 188 prologue draws side by side (a 3 000-row batch of
        # figures/main_figures.py:154-156) re-read the member's 0.9 MB through L2 and cost the kernel 17 us against 12 for one, a draw
        # launch in front of it 5 (scripts/dev/small_forward_timing.py: 48 against 58 us for the pair).
        csz = -(-(chunk_B or B) // max(nchunks, 1))
        single_launch = (-(-min(csz, B) // 16) <= 8) if small else csz <= 256
    ws = None if single_launch else _workspace(J, d, x.device)
    # (x, the ensemble, the noise tensors and out were checked above -- float32, contiguous, on x's device: their addresses as they are)
    dp = lambda t: None if t is None else t.data_ptr()
    N.check(N.lib().bnn_multiswag_f32(plan.handle, C.byref(g), x.data_ptr(), w_avg.data_ptr(), w2_avg.data_ptr(), pre_D.data_ptr(), S, K,
                                      seed_idx.data_ptr(), dp(z1), dp(z2), dp(eps), float(scale), int(philox_seed),
                                      int(draw_id0), int(system_id0), dp(ws), out.data_ptr(), dp(pre), dp(summ), st))
    return (out, pre, summ) if debug else out


@_on_device_of(0)
def quantiles(samples, q=(50.0,)):
    """Per-system percentiles over the draws, numpy 'linear' interpolation: samples [R,B,2] -> [B,2,len(q)] float32.
    q=50 is np.median(_preds[..., c], 0) of figures/main_figures.py:277-278."""
    import numpy as np
    samples = _f32(samples, "samples")
    R, B, _ = samples.shape
    qa = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
    out = torch.empty((B, 2, qa.size), dtype=torch.float32, device=samples.device)
    N.check(N.lib().bnn_quantiles_f32(N.ptr(samples), R, B, qa.ctypes.data, qa.size, N.ptr(out), N.stream_ptr()))
    return out


def feature_pack(tseries=None, mass=None, X=None, mean=None, scale=None, want_x64=False):
    """data_setup_kernel (figures/spock/regression.py:183-213) + ssX.transform + .float() on the GPU.

    tseries [N,T,26] + mass [N,3] float64 (or X [N,T,41] float64 already packed) -> x [N,T,41] float32 standardised with
    mean/scale [41] float64; want_x64 also returns the unstandardised float64 X (data_setup_kernel's return value)."""
    f64 = lambda t: None if t is None else torch.as_tensor(t, dtype=torch.float64).cuda().contiguous()
    tseries, mass, X, mean, scale = f64(tseries), f64(mass), f64(X), f64(mean), f64(scale)
    src = tseries if tseries is not None else X
    if src is None or src.dim() != 3 or src.shape[2] != (26 if tseries is not None else 41):
        raise NotImplementedError("Need to change indexes above for angles, replace ssX.")  # regression.py:210-211
    Nn, T = src.shape[0], src.shape[1]
    x32 = torch.empty((Nn, T, 41), dtype=torch.float32, device=src.device) if mean is not None else None
    x64 = torch.empty((Nn, T, 41), dtype=torch.float64, device=src.device) if (want_x64 or x32 is None) else None
    N.check(N.lib().bnn_feature_pack_f64(N.ptr(tseries), N.ptr(mass), N.ptr(X), Nn, T, N.ptr(mean), N.ptr(scale), N.ptr(x64),
                                         N.ptr(x32), N.stream_ptr()))
    if x32 is None:
        return x64
    return (x32, x64) if want_x64 else x32


@_on_device_of(0)
def moments(samples, mom=None):
    """samples [R,B,2] -> float64 [B,4] = sum mu, sum mu^2, sum std, sum std^2 (accumulates into `mom` if given)."""
    samples = _f32(samples, "samples")
    R, B, _ = samples.shape
    acc = mom is not None
    if mom is None:
        mom = torch.empty((B, 4), dtype=torch.float64, device=samples.device)
    N.check(N.lib().bnn_moments_f64(N.ptr(samples), R, B, N.ptr(mom), int(acc), N.stream_ptr()))
    return mom


@_on_device_of(0)
def regress(summary, W, plan=None, debug=False):
    """predict_instability on an explicit summary: summary [J,B,40], W [J,d] -> out [J,B,2] (and pre_clamp with debug)."""
    summary = _f32(summary, "summary")
    W = _f32(W, "W")
    J, B, S = summary.shape
    plan = plan or get_plan()
    if S != plan.summary_width or W.shape != (J, plan.d):
        raise ValueError(f"regress needs summary [J,B,{plan.summary_width}] and W [J,{plan.d}]")
    out = torch.empty((J, B, 2), dtype=torch.float32, device=summary.device)
    pre = torch.empty_like(out) if debug else None
    N.check(N.lib().bnn_regress_f32(plan.handle, N.ptr(summary), N.ptr(W), J, B, N.ptr(out), N.ptr(pre) if debug else None,
                                    N.stream_ptr()))
    return (out, pre) if debug else out


def philox_normal(kind, philox_seed, id0, n_rows, width=0, B=0, system_id0=0, device="cuda", n_features=41):
    """The normals the kernels generate in-kernel: kind 0 -> z1[n_rows,width], 1 -> z2[n_rows,width], 2 -> eps[n_rows,B,2,latent=width or 20],
    3 -> eps_in[n_rows,B,T=width,n_features], 4 -> eps_sum[n_rows,B,width or 40] (the summary width: 2 latent, + 2 with fix_megno);
    statistics epilogue: 5 -> truncated-normal candidates [n_rows,B,nsamp=width], 6 -> survival level of the prior draw [n_rows,B]."""
    shape = {2: (n_rows, B, 2, width or LATENT), 3: (n_rows, B, width, n_features), 4: (n_rows, B, width or 2 * LATENT), 5: (n_rows, B, width),
             6: (n_rows, B)}.get(kind, (n_rows, width))
    out = torch.empty(shape, dtype=torch.float32, device=device)
    N.check(N.lib().bnn_philox_normal_f32(kind, int(philox_seed), int(id0), n_rows, B, int(system_id0), width, int(n_features), N.ptr(out),
                                          N.stream_ptr()))
    return out


def philox_raw(ctr, key, n, device="cuda"):
    out = torch.empty((n, 4), dtype=torch.int32, device=device)
    N.check(N.lib().bnn_philox_raw_u32(*[int(c) for c in ctr], int(key[0]), int(key[1]), n, N.ptr(out), N.stream_ptr()))
    return out


# ---- streaming statistics epilogue (include/bnn_chaos_hip.h "streaming statistics epilogue") -----------
_prior_tables = {}


def stats_params(nsamp=40, left=4.0, prior_threshold=9.0, prior_top=100.0, prior_knots=8192, device=None):
    """bnn_stats for fast_truncnorm(left, nsamp) + prior resampling at/above prior_threshold (None or inf: no resampling)
    (figures/multiswag_5_planet.py:388-422).  The survival table of the prior is built once per device and cached."""
    import numpy as np
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    thr = float("inf") if prior_threshold is None else float(prior_threshold)
    st = N.BnnStats(int(nsamp), float(left), thr, 0, 0.0, 0, None)
    if thr != float("inf"):
        key = (dev.index, thr, float(prior_top), int(prior_knots))
        if key not in _prior_tables:
            host = np.zeros(int(prior_knots), np.float32)
            step = C.c_double()
            N.check(N.lib().bnn_prior_table_f32(thr, float(prior_top), int(prior_knots), host.ctypes.data, C.byref(step)))
            _prior_tables[key] = (torch.as_tensor(host).to(dev), step.value)
        tab, step = _prior_tables[key]
        st.prior_m, st.prior_step, st.prior_surv = int(prior_knots), step, tab.data_ptr()
        st._keepalive = tab
    return st


@_on_device_of(0)
def stats_draw(musd, st=None, philox_seed=0, row_id0=0, system_id0=0):
    """The statistics epilogue on materialised pairs: musd [R,B,2] (the forward's output) -> t [R,B], one log10 instability
    time per evaluation (truncated-normal draw, then prior resampling); Philox keyed by (global row, global system)."""
    musd = _f32(musd, "musd")
    if musd.dim() != 3 or musd.shape[2] != 2:
        raise ValueError("musd must be [R,B,2]")
    st = st or stats_params(device=musd.device)
    R, B, _ = musd.shape
    out = torch.empty((R, B), dtype=torch.float32, device=musd.device)
    N.check(N.lib().bnn_stats_draw_f32(N.ptr(musd), R, B, C.byref(st), int(philox_seed), int(row_id0), int(system_id0), N.ptr(out),
                                       N.stream_ptr()))
    return out


@_on_device_of(0)
def multiswag_stats(x, w_avg, w2_avg, pre_D, seed_idx, st=None, z1=None, z2=None, eps=None, nchunks=1, scale=0.5, philox_seed=0,
                    draw_id0=0, system_id0=0, plan=None, systems_per_block=0, out=None, chunk_B=0, chunk_off=0, assume_finite=False,
                    nonfinite=None):
    """multiswag with the statistics epilogue fused into the kernel's tail -> t [J/nchunks, B]: (mu, std) never reach memory.
    Bit-identical to stats_draw(multiswag(...), row_id0=draw_id0 // nchunks, system_id0=system_id0)."""
    plan = plan or get_plan()
    x = _f32(x, "x")
    _check_x(x, plan)
    w_avg, w2_avg, pre_D, S, d, K = _ensemble(plan, w_avg, w2_avg, pre_D)
    B, T, _ = x.shape
    if seed_idx.device != x.device or seed_idx.dtype != torch.int32 or not seed_idx.is_contiguous():
        seed_idx = seed_idx.to(device=x.device, dtype=torch.int32).contiguous()
    J = seed_idx.numel()
    _check_grid(J, nchunks, draw_id0)
    R = J // nchunks
    z1, z2, eps = _f32(z1, "z1"), _f32(z2, "z2"), _f32(eps, "eps")
    _check_same_device(x, w_avg=w_avg, w2_avg=w2_avg, pre_D=pre_D, z1=z1, z2=z2, eps=eps)
    if (z1 is None) != (z2 is None) or (z1 is not None and (tuple(z1.shape) != (J, d) or tuple(z2.shape) != (J, K))):
        raise ValueError("z1 must be [J,d] and z2 [J,K] (or both None)")
    if eps is not None and tuple(eps.shape) != (R, B, 2, plan.latent):
        raise ValueError(f"eps must be [{R},{B},2,{plan.latent}]")
    st = st or stats_params(device=x.device)
    if out is None:
        out = torch.empty((R, B), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (R, B) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out has the wrong shape/dtype")
    g = _grid(B, T, J, nchunks, systems_per_block, chunk_B=chunk_B, chunk_off=chunk_off, nonfinite=_nonfinite_record(x, plan, assume_finite, nonfinite))
    ws = _workspace(max(J, 1), d, x.device)
    N.check(N.lib().bnn_multiswag_stats_f32(plan.handle, C.byref(g), N.ptr(x), N.ptr(w_avg), N.ptr(w2_avg), N.ptr(pre_D), S, K,
                                            N.ptr(seed_idx), N.ptr(z1), N.ptr(z2), N.ptr(eps), float(scale), int(philox_seed),
                                            int(draw_id0), int(system_id0), N.ptr(ws), C.byref(st), N.ptr(out), N.stream_ptr()))
    return out


class QuantileSketch:
    """Streaming per-simulation quantile sketch over any number of draws in O(n_sims * bins) memory: min over `group`
    consecutive systems (min over trios, figures/multiswag_5_planet.py:428), then a histogram over piecewise-uniform bins
    + float64 sum / sum of squares.  percentiles() = np.percentile(..., 'linear') read off the histogram: every estimate is
    within one bin width (`resolution(t)`) of the exact order statistic.

    Default bins: [4, 9) in steps of 1/128 (the truncated-normal range below the prior threshold), [9, 13) in 1/32,
    [13, 101) in 1/2 (the prior's tail: P(t > 13 | t >= 9) = 0.18, P(t > 30 | t >= 9) = 1.4e-4); below 4 -> bin 0."""

    DEFAULT = ((4.0, 9.0, 640), (9.0, 13.0, 128), (13.0, 101.0, 176))

    def __init__(self, n_systems, group=1, segments=None, device=None):
        if n_systems % group:
            raise ValueError("n_systems must be a multiple of group")
        self.group, self.n_sims = int(group), n_systems // group
        self.segments = tuple(segments or self.DEFAULT)
        sk = N.BnnSketch()
        sk.nseg = len(self.segments)
        for i, (lo, hi, n) in enumerate(self.segments):
            sk.lo[i], sk.hi[i], sk.n[i] = lo, hi, n
        self.spec = sk
        self.nbins = N.check(N.lib().bnn_sketch_bins(C.byref(sk)))
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.hist = torch.zeros((self.nbins, self.n_sims), dtype=torch.int32, device=dev)   # uint32 counters
        self.mom = torch.zeros((self.n_sims, 2), dtype=torch.float64, device=dev)
        self.count = 0

    def resolution(self, t):
        for lo, hi, n in self.segments:
            if t < hi:
                return (hi - lo) / n
        lo, hi, n = self.segments[-1]
        return (hi - lo) / n

    def update(self, t):
        """t [R, n_systems] float32 on the sketch's device: R more draws."""
        t = _f32(t, "t")
        if t.dim() != 2 or t.shape[1] != self.n_sims * self.group or t.device != self.hist.device:
            raise ValueError("t must be [R, n_systems] on the sketch's device")
        with torch.cuda.device(t.device):
            N.check(N.lib().bnn_sketch_update_u32(N.ptr(t), t.shape[0], t.shape[1], self.group, C.byref(self.spec), N.ptr(self.hist),
                                                  N.ptr(self.mom), N.stream_ptr()))
        self.count += t.shape[0]
        return self

    def percentiles(self, q):
        import numpy as np
        qa = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
        out = torch.empty((self.n_sims, qa.size), dtype=torch.float32, device=self.hist.device)
        with torch.cuda.device(self.hist.device):
            N.check(N.lib().bnn_sketch_quantiles_f32(N.ptr(self.hist), self.n_sims, C.byref(self.spec), qa.ctypes.data, qa.size,
                                                     N.ptr(out), N.stream_ptr()))
        return out

    def mean(self):
        return self.mom[:, 0] / max(self.count, 1)


# ---- slab drivers in the native library: the whole grid reduced on the fly ---------------------------------------------------------
def _slab_common(x, w_avg, w2_avg, pre_D, seed_idx, nchunks, draws_per_launch, draw_id0, plan):
    plan = plan or get_plan()
    x = _f32(x, "x")
    _check_x(x, plan)
    w_avg, w2_avg, pre_D, S, d, K = _ensemble(plan, w_avg, w2_avg, pre_D)
    _check_same_device(x, w_avg=w_avg, w2_avg=w2_avg, pre_D=pre_D)
    seed_idx = seed_idx.to(device=x.device, dtype=torch.int32).contiguous()
    J = seed_idx.numel()
    _check_grid(J, nchunks, draw_id0)
    dpl = max(nchunks, (int(draws_per_launch) // nchunks) * nchunks)
    dpl = min(dpl, max(J, nchunks))
    return plan, x, w_avg, w2_avg, pre_D, S, d, K, seed_idx, J, dpl


@_on_device_of(0)
def multiswag_moments(x, w_avg, w2_avg, pre_D, seed_idx, nchunks=1, scale=0.5, philox_seed=0, draw_id0=0, system_id0=0,
                      draws_per_launch=256, plan=None, chunk_B=0, chunk_off=0, assume_finite=False, nonfinite=None):
    """Predictive moments of the whole grid -> float64 [B,4] (sum mu, sum mu^2, sum std, sum std^2 over the output rows), the
    draws evaluated `draws_per_launch` at a time inside ONE native call (bnn_multiswag_moments_f64): [J,B,2] never exists."""
    plan, x, w_avg, w2_avg, pre_D, S, d, K, seed_idx, J, dpl = _slab_common(x, w_avg, w2_avg, pre_D, seed_idx, nchunks, draws_per_launch, draw_id0, plan)
    B, T, _ = x.shape
    mom = torch.empty((B, 4), dtype=torch.float64, device=x.device)
    ws = _workspace(dpl, d, x.device)
    outw = torch.empty((dpl // nchunks, B, 2), dtype=torch.float32, device=x.device)
    g = _grid(B, T, J, nchunks, 0, chunk_B=chunk_B, chunk_off=chunk_off, nonfinite=_nonfinite_record(x, plan, assume_finite, nonfinite))   # ONE scan for all slabs
    N.check(N.lib().bnn_multiswag_moments_f64(plan.handle, C.byref(g), N.ptr(x), N.ptr(w_avg), N.ptr(w2_avg), N.ptr(pre_D), S, K,
                                              N.ptr(seed_idx), float(scale), int(philox_seed), int(draw_id0), int(system_id0), dpl,
                                              N.ptr(ws), N.ptr(outw), N.ptr(mom), N.stream_ptr()))
    if B and J == 0:
        mom.zero_()
    return mom


@_on_device_of(0)
def multiswag_bands(x, w_avg, w2_avg, pre_D, seed_idx, sketch, st=None, nchunks=1, scale=0.5, philox_seed=0, draw_id0=0,
                    system_id0=0, draws_per_launch=256, plan=None, chunk_B=0, chunk_off=0, assume_finite=False, nonfinite=None):
    """The whole grid streamed into `sketch` (a QuantileSketch over x's systems) inside ONE native call
    (bnn_multiswag_bands_f32): statistics epilogue in the forward tail, min over the sketch's group, histogram update."""
    plan, x, w_avg, w2_avg, pre_D, S, d, K, seed_idx, J, dpl = _slab_common(x, w_avg, w2_avg, pre_D, seed_idx, nchunks, draws_per_launch, draw_id0, plan)
    B, T, _ = x.shape
    if sketch.n_sims * sketch.group != B or sketch.hist.device != x.device:
        raise ValueError("the sketch must cover x's systems on x's device")
    st = st or stats_params(device=x.device)
    ws = _workspace(dpl, d, x.device)
    tw = torch.empty((dpl // nchunks, B), dtype=torch.float32, device=x.device)
    g = _grid(B, T, J, nchunks, 0, chunk_B=chunk_B, chunk_off=chunk_off, nonfinite=_nonfinite_record(x, plan, assume_finite, nonfinite))
    N.check(N.lib().bnn_multiswag_bands_f32(plan.handle, C.byref(g), N.ptr(x), N.ptr(w_avg), N.ptr(w2_avg), N.ptr(pre_D), S, K,
                                            N.ptr(seed_idx), float(scale), int(philox_seed), int(draw_id0), int(system_id0), dpl,
                                            N.ptr(ws), N.ptr(tw), C.byref(st), sketch.group, C.byref(sketch.spec), N.ptr(sketch.hist),
                                            N.ptr(sketch.mom), N.stream_ptr()))
    sketch.count += J // nchunks
    return sketch
