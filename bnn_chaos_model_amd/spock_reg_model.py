"""Drop-in inference surface of the reference's spock_reg_model.py, backed by the gfx950 kernels.

Mirrors (reference file:line):
    soft_clamp            spock_reg_model.py:295-296
    VarModel              :339-545   forward / sample / compute_summary_stats / predict_instability / masks
    SWAGModel             :689-908   init_params / flatten / load / sample_weights / forward_swag / forward_swag_fast
    save_swag, load_swag  :911-967

Only inference is here; the training half of the reference class is out of scope.
The arithmetic runs on the GPU through bnn_chaos_model_amd.ops -- there is no CPU path; CPU tensors are
copied to the GPU and results come back on the caller's device, as the reference's callers expect.

RNG contract.  With `rng = "torch"` (default) every random number is drawn
from torch's global generator with the reference's calls, shapes and order (randn((1,d)), randn((K,1)),
randn_like(x), randn_like([B,20]) x2, randn_like([B,40])) on the device the reference would use, and handed
to the kernels as explicit noise: after torch.manual_seed(s) the outputs match the reference run with the same
seed to fp32 rounding.  With `rng = "philox"` the kernels generate counter-based noise themselves
(no noise tensors in HBM); set `philox_seed` for reproducibility.
"""
import random
import warnings
from collections import OrderedDict
from copy import deepcopy as copy  # noqa: F401  (the reference module exports it: spock_reg_model.py, `from copy import deepcopy as copy`)

import numpy as np
import os

import torch
from torch import nn

from . import _native as _N
from . import ops
from .checkpoint import AttributeDict, read_swag_file, write_swag_file

EPSILON = 1e-5  # spock_reg_model.py:337

# The reference attaches one fixed StandardScaler to every 'v50' checkpoint (spock_reg_model.py:931-957); its 2 x 41
# float64 constants live in data/v50_ssx.json.
_V50 = None


def _v50_constants():
    global _V50
    if _V50 is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "v50_ssx.json")) as f:
            d = json.load(f)
        _V50 = (np.array(d["mean_"], dtype=np.float64), np.array(d["scale_"], dtype=np.float64))
    return _V50


class StandardScaler:
    """The three attributes and one method of sklearn's StandardScaler that the evaluation scripts use."""

    def __init__(self, mean_=None, scale_=None):
        self.mean_ = None if mean_ is None else np.asarray(mean_, dtype=np.float64)
        self.scale_ = None if scale_ is None else np.asarray(scale_, dtype=np.float64)
        self.var_ = None if scale_ is None else self.scale_ ** 2

    def transform(self, X):
        X = np.array(X, dtype=np.float64)  # copy, float64 like sklearn
        X -= self.mean_
        X /= self.scale_
        return X

    def inverse_transform(self, X):
        return np.asarray(X, dtype=np.float64) * self.scale_ + self.mean_


def v50_scaler():
    mean_, scale_ = _v50_constants()
    return StandardScaler(mean_.copy(), scale_.copy())


def soft_clamp(x, lo, high):
    return 0.5 * (torch.tanh(x) + 1) * (high - lo) + lo


def mlp(in_n, out_n, hidden, layers):
    """Same module sequence as the reference's mlp() (:301-321, act='relu'); used for its state_dict layout and init."""
    if layers == 0:
        return nn.Linear(in_n, out_n)
    result = [nn.Linear(in_n, hidden), nn.ReLU()]
    for _ in range(layers):
        result.extend([nn.Linear(hidden, hidden), nn.ReLU()])
    result.extend([nn.Linear(hidden, out_n)])
    return nn.Sequential(*result)


def _mlp_layout(prefix, in_n, out_n, hidden, layers):
    """state_dict keys / shapes of mlp(in_n, out_n, hidden, layers) (:301-321): a bare Linear for layers == 0, else a Sequential whose
    Linear modules sit at the even indices."""
    if layers == 0:
        return [(prefix + ".weight", (out_n, in_n)), (prefix + ".bias", (out_n,))]
    dims = [(in_n, hidden)] + [(hidden, hidden)] * layers + [(hidden, out_n)]
    out = []
    for i, (k, n) in enumerate(dims):
        out += [(f"{prefix}.{2 * i}.weight", (n, k)), (f"{prefix}.{2 * i}.bias", (n,))]
    return out


def _state_layout(fix_megno=False, n_features=41, hidden=40, latent=20, depth_in=1, depth_out=1):
    """state_dict order and shapes (reference :734-761: own parameters, then feature_nn, then regress_nn) of the network VarModel builds
    from hparams (:358-362); with fix_megno the summary is two wider."""
    sm = 2 * latent + (2 if fix_megno else 0)
    return tuple([("input_noise_logvar", (n_features,)), ("summary_noise_logvar", (sm,))]
                 + _mlp_layout("feature_nn", n_features, latent, hidden, depth_in)
                 + _mlp_layout("regress_nn", sm, 2, hidden, depth_out))


_STATE_LAYOUT = _state_layout(False)


_GPU_SEEN = False
_GPU_DEVS = {}
_CPU = torch.device("cpu")


def _gpu():
    """The current GPU as a torch.device (availability is asked once; the device objects are made once per index: this runs per call of
    the scripts' per-chunk loop)."""
    global _GPU_SEEN
    if not _GPU_SEEN:
        if not torch.cuda.is_available():
            raise RuntimeError("bnn_chaos_model_amd needs an MI355X (gfx950) GPU: there is no CPU implementation")
        _GPU_SEEN = True
    i = _N.current_device()
    d = _GPU_DEVS.get(i)
    if d is None:
        d = _GPU_DEVS[i] = torch.device("cuda", i)
    return d


_TOPS = None       # torch.ops.bnn_chaos once torch_ops has registered it
_IDX0 = {}         # device -> int32 [1] zeros: "ensemble member 0" of a one-member state (made once, not once per call)


def _idx0(g):
    t = _IDX0.get(g)
    if t is None:
        t = _IDX0[g] = torch.zeros(1, dtype=torch.int32, device=g)
    return t


_STRIDED_OK = {}   # device -> bool
_TORCH_RANDN = torch.randn


def _strided_normal_is_randn(g):
    """True when `view.normal_()` on a strided [B, latent] view of a [B, 2, latent] tensor -- and `torch.randn(shape, out=t)` -- draw exactly
    what torch.randn(B, latent) draws from the same generator state -- asked of the GPU's generator once per device, on a copy of its state (the global stream is not
    advanced).  It does with torch 2.10 (the kernel maps logical element index -> Philox counter whatever the strides); if a torch version
    ever changes that, the surface goes back to two torch.randn calls + a stack and stays on the reference's stream."""
    if torch.randn is not _TORCH_RANDN:   # somebody records or replays the draws (tests' tapes and players patch torch.randn): draw through it
        return False
    ok = _STRIDED_OK.get(g)
    if ok is None:
        state = torch.cuda.get_rng_state(g)
        try:
            ok = True
            for B in (3, 700):
                torch.cuda.set_rng_state(state, g)
                a, b = torch.randn(B, 20, device=g), torch.randn(B, 20, device=g)
                torch.cuda.set_rng_state(state, g)
                buf = torch.empty((B, 2, 20), device=g)
                buf[:, 0].normal_()
                buf[:, 1].normal_()
                ok = ok and bool(torch.equal(buf[:, 0], a)) and bool(torch.equal(buf[:, 1], b))
                torch.cuda.set_rng_state(state, g)
                oa, ob = torch.empty((B, 20), device=g), torch.empty((B, 20), device=g)
                torch.randn((B, 20), out=oa)          # (randn(out=) into an existing tensor: the z_1 / z_2 buffers)
                torch.randn((B, 20), out=ob)
                ok = ok and bool(torch.equal(oa, a)) and bool(torch.equal(ob, b))
        finally:
            torch.cuda.set_rng_state(state, g)
        _STRIDED_OK[g] = ok
    return ok


def _as_gpu_f32(t, g):
    """t as a contiguous float32 tensor on GPU g -- t itself when it already is one (a no-op .to() / .contiguous() pair still costs
    microseconds, and the scripts' loop pays it per 15-row chunk)."""
    t = t.detach()
    if t.device == g and t.dtype == torch.float32 and t.is_contiguous():
        return t
    return t.to(g, torch.float32).contiguous()


class VarModel:
    """Bayesian neural network predicting (mu, std) of log10 instability time (reference :339-545), inference only."""

    def __init__(self, hparams):
        hparams = AttributeDict(hparams)
        if "seed" not in hparams:
            hparams["seed"] = 0
        # pl.seed_everything(hparams['seed']) (:345): part of the RNG contract
        random.seed(hparams["seed"])
        np.random.seed(hparams["seed"])
        torch.manual_seed(hparams["seed"])
        hparams["include_derivatives"] = hparams.get("include_derivatives", False)
        if "time_series_features" not in hparams:
            hparams["time_series_features"] = 38 + 3
        if hparams["time_series_features"] == 82:
            hparams["time_series_features"] = 41
        self.fix_megno = hparams.get("fix_megno", False)
        self.fix_megno2 = hparams.get("fix_megno2", False)
        self.include_angles = hparams.get("include_angles", False)
        self.n_features = hparams["time_series_features"] * (1 + int(hparams["include_derivatives"]))
        self.fix_megno = bool(self.fix_megno)
        self._arch = dict(n_features=int(self.n_features), hidden=int(hparams["hidden"]), latent=int(hparams["latent"]),
                          depth_in=int(hparams["in"]), depth_out=int(hparams["out"]))
        if self.n_features not in (41, 82):
            raise NotImplementedError("time_series_features must be 41 (82 with include_derivatives): the reference's feature packing "
                                      "produces nothing else (figures/spock/regression.py:210-211)")
        if max(self._arch["hidden"], self._arch["latent"], 2 * self._arch["latent"] + 2 * int(self.fix_megno)) > 128:
            raise NotImplementedError("hidden / latent / summary widths above 128 are not built for gfx950 (DESIGN.md section 4.9)")
        # reference init order (:359-362): feature_nn, regress_nn, then the two noise parameters
        feature_nn = mlp(self.n_features, hparams["latent"], hparams["hidden"], hparams["in"])
        regress_nn = mlp(hparams["latent"] * 2 + int(self.fix_megno) * 2, 2, hparams["hidden"], hparams["out"])
        self.lowest = 0.1 if hparams.get("lower_std", False) else 0.5
        sd = OrderedDict()
        sd["input_noise_logvar"] = torch.zeros(self.n_features) - 2
        sd["summary_noise_logvar"] = torch.zeros(hparams["latent"] * 2 + int(self.fix_megno) * 2) - 2
        for k, v in feature_nn.state_dict().items():
            sd["feature_nn." + k] = v
        for k, v in regress_nn.state_dict().items():
            sd["regress_nn." + k] = v
        self._layout = _state_layout(self.fix_megno, **self._arch)
        assert tuple((k, tuple(v.shape)) for k, v in sd.items()) == self._layout
        self._pending_draw = None
        self._w = torch.cat([v.detach().reshape(-1) for v in sd.values()]).float().contiguous()  # flat vector [d]

        self._latents_cache, self._last_latents_args = None, None
        self._spec = ((False, True), None) if os.environ.get("BNN_AUTO_SPECIALIZE", "") not in ("", "0") else None
        self.megno_location = 7
        self.mmr_location = [3, 6]
        self.nan_location = [38, 39, 40]
        self.eplusminus_location = [1, 2, 4, 5]
        hparams["scheduler_choice"] = "swa"
        for k, v in (("save_freq", 25), ("eval_freq", 5), ("momentum", 0.9), ("weight_decay", 1e-4), ("noisy_val", True)):
            hparams.setdefault(k, v)
        self.hparams = hparams
        self.random_sample = hparams.get("random_sample", False)
        self.include_mmr = hparams["include_mmr"]
        self.include_nan = hparams["include_nan"]
        self.include_eplusminus = hparams.get("include_eplusminus", True)
        self._summary_kl = 0.0
        self._last_forward = None
        self._cur_summary_cache = None
        self.ssX = None
        self.ssy = None
        self.training = True
        self._device = torch.device("cpu")
        # rng policy, see module docstring
        self.rng = "torch"
        self.philox_seed = 0
        self._philox_calls = 0
        # Non-finite inputs (not a reference attribute).  False: every call scans x once and systems that hold NaN / +-inf get what the
        # reference returns for them -- NaN through `x - mask` (:452-478) and the NaN-propagating nn.ReLU, or the exact IEEE value
        # where an infinity dies in a ReLU (ops.nonfinite_scan).  True: x is known to be clean and the scan is skipped.
        self.assume_finite = False

    # forward()'s side effect `self._cur_summary = summary_stats` (:512): the kernels do not write the summary unless asked, so it is
    # produced on demand by re-running the last forward with its debug output (same weights, same noise, the Philox seed of THAT call).
    # `_summary_kl` (:515-520) is read by the training loss only and stays 0.  The records behind the two lazy side effects hold the
    # call's x on the GPU: when the caller's x already was a contiguous float32 GPU tensor that is the caller's own storage, by
    # reference -- mutate it in place before reading `_cur_summary` / `latents` and they follow the new values (the reference computes
    # both at forward time).  Reading either once caches it.
    @property
    def _cur_summary(self):
        if self._cur_summary_cache is None and self._last_forward is not None:
            xg, Wg, eps, eps_in, eps_sum, noisy, did, plan, seed = self._last_forward
            _, _, summ = ops.forward(xg, Wg, eps=eps, eps_in=eps_in, eps_sum=eps_sum, philox_seed=seed, draw_id0=did, plan=plan,
                                     debug=True, noisy=noisy, assume_finite=self.assume_finite)
            self._cur_summary_cache = summ[0]
        return self._cur_summary_cache

    @property
    def latents(self):
        """compute_summary_stats' side effect `self.latents = feature_nn(x)` [B,T,latent] (:417, :433) of the last forward / of the
        last compute_summary_stats call, produced on demand (same weights, same masks, same input noise)."""
        if self._latents_cache is None and self._last_latents_args is not None:
            xg, Wg, eps_in, noisy, did, plan, seed = self._last_latents_args
            if callable(Wg):   # after forward_swag_fast: the weights its fused kernel drew, re-drawn on demand from the call's own normals
                Wg = Wg()[None].to(xg.device)   # (a closure over the draw, not over the module: load() / sample_weights() since do not matter)
            self._latents_cache = ops.feature_latents(xg, Wg, eps_in=eps_in, noisy=noisy, philox_seed=seed, draw_id0=did, plan=plan or self._plan(),
                                                      assume_finite=self.assume_finite)[0]
        return self._latents_cache

    @latents.setter
    def latents(self, v):
        self._latents_cache, self._last_latents_args = v, None

    def drop_records(self):
        """Not in the reference: let go of what the two lazy side effects keep alive -- the last call's x on the GPU, its noise tensors
        (forward(noisy_val=True): an x-sized eps_in) and weights.  `_cur_summary` / `latents` read None afterwards, until the next call."""
        self._last_forward = self._last_latents_args = None
        self._cur_summary_cache = self._latents_cache = None
        self.__dict__.pop("_noise_bufs", None)   # (the GPU-resident route's per-(stream, batch size) noise buffers)
        return self

    # ---- nn.Module-like conveniences used by the evaluation scripts --------------------------------------------
    @property
    def device(self):
        return self._device

    # The flat parameter vector.  forward_swag_fast samples the weights inside the fused kernel; the module's
    # "currently loaded weights" (reference :838) are then materialised lazily, only if somebody asks for them.
    @property
    def _w(self):
        if self._pending_draw is not None:
            fn, self._pending_draw = self._pending_draw, None
            self._w_store = fn().to(self._device)
        return self._w_store

    @_w.setter
    def _w(self, v):
        self._pending_draw = None   # (a pending `latents` record keeps its own closure over the draw: nothing is evaluated here)
        self._w_store = v

    def to(self, device):
        self._device = device if isinstance(device, torch.device) else torch.device(device)
        if self._pending_draw is None:  # a pending in-kernel draw stays pending: it lands on the new device when asked for
            self._w_store = self._w_store.to(self._device)
        return self

    def cpu(self):
        return self.to(_CPU)

    def cuda(self, device=None):
        return self.to(_gpu() if device is None else device)

    def eval(self):
        self.training = False
        return self

    def specialize(self, noisy=(False, True), w8=None):
        """Not in the reference (PyTorch needs no such step): compile this model's network into its own form of the generic forward
        engine (specialize.py: ~10 s of hipcc per form, cached on disk) -- bit-identical outputs, the schedule of a kernel written
        for these shapes.  The pretrained ensemble's network at T % 4 == 0 keeps its own kernels either way.  Every plan the model
        hands out from now on (other devices of a multi-GPU call, another column mask) is specialised the same way.
        BNN_AUTO_SPECIALIZE=1 in the environment does this for every model at construction."""
        self._spec = (tuple(noisy) if not isinstance(noisy, bool) else (noisy,), w8)
        self._plan()
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def state_dict(self):
        out, i = OrderedDict(), 0
        for k, shp in self._layout:
            n = int(np.prod(shp))
            out[k] = self._w[i:i + n].reshape(shp)
            i += n
        return out

    def load_state_dict(self, sd):
        self._w = torch.cat([sd[k].detach().reshape(-1).float() for k, _ in self._layout]).to(self._device).contiguous()

    def flatten(self):
        """Convert state dict into a vector (:734-746)."""
        return self._w.clone()

    def load(self, p_vec):
        """Load a vector into the state dict (:748-761)."""
        p_vec = torch.as_tensor(p_vec).detach().reshape(-1)
        if p_vec.numel() != self._w.numel():
            raise RuntimeError(f"size mismatch: expected {self._w.numel()} parameters, got {p_vec.numel()}")
        self._w = p_vec.to(device=self._device, dtype=torch.float32).contiguous().clone()

    # ---- kernel plumbing ---------------------------------------------------------------------------------------
    def zero_mask(self):
        return ops.zero_mask_from_flags(self.fix_megno, self.fix_megno2, self.include_mmr, self.include_nan,
                                        self.include_eplusminus)

    def _op_args(self):
        """(zero_mask, lowest_std, net) as the torch.ops.bnn_chaos.* entry points take them (torch_ops.py).  Rebuilt only when one of the
        flags it is made of has changed (the scripts' loop calls this once per chunk per sample)."""
        key = (self.fix_megno, self.fix_megno2, self.include_mmr, self.include_nan, self.include_eplusminus, self.lowest)
        hit = self.__dict__.get("_op_args_cache")
        if hit is None or hit[0] != key:
            a = self._arch
            hit = (key, (self.zero_mask(), float(self.lowest),
                         [a["n_features"], a["hidden"], a["latent"], a["depth_in"], a["depth_out"], int(self.fix_megno)]))
            self.__dict__["_op_args_cache"] = hit
        return hit[1]

    @staticmethod
    def _tops():
        global _TOPS
        if _TOPS is None:
            from . import torch_ops  # noqa: F401  (registers torch.ops.bnn_chaos.*)
            _TOPS = torch.ops.bnn_chaos
        return _TOPS

    def _plan(self, zero_mask=None, device=None):
        plan = ops.get_plan(self.zero_mask() if zero_mask is None else zero_mask, self.lowest, device=device, fix_megno=self.fix_megno,
                            **self._arch)
        spec = getattr(self, "_spec", None)
        if spec is not None and plan.__dict__.get("_spec_req") != spec and not (plan.v50net and not self.fix_megno):
            ops.specialize(plan, noisy=spec[0], w8=spec[1])   # (the pretrained network has its forms in the library: nothing to compile)
            plan.__dict__["_spec_req"] = spec                  # once per plan
        elif spec is None and not plan.v50net and not plan.__dict__.get("_warned_generic") and not (plan.spec_attached(False) or plan.spec_attached(True)):
            # Every shape the reference's CLI defaults produce (parse_swag_args.py:11-16, find_minima.py:33-65: hidden 40, latent 20,
            # in = out = 1, any mask, --megno) has hand-scheduled kernels in the library.  Other widths / depths run on the ahead-of-time
            # generic engine -- correct to the same bits, but at 0.3-0.6 of the fp32 matrix roof; say so once, with the way out.
            plan.__dict__["_warned_generic"] = True
            a = self._arch
            warnings.warn(f"bnn_chaos_model_amd: the network hidden={a['hidden']} latent={a['latent']} in={a['depth_in']} out={a['depth_out']} "
                          f"features={a['n_features']} runs on the ahead-of-time generic forward engine (about 0.3-0.6 of the fp32 matrix "
                          "roof, against 0.9 for the pretrained shapes).  model.specialize() -- or BNN_AUTO_SPECIALIZE=1 -- compiles this "
                          "network's own form once (hipcc, ~10 s, cached on disk; 0.73-0.88; bit-identical results).  Without a compiler on "
                          "the machine the ahead-of-time form is what runs.", GenericEngineWarning, stacklevel=3)
        return plan

    @property
    def _latent(self):
        return self._arch["latent"]

    @property
    def _summary_width(self):
        return 2 * self._latent + (2 if self.fix_megno else 0)

    def _check_x(self, x):
        if x.dim() != 3 or x.shape[-1] != self.n_features:
            raise NotImplementedError(f"x must be [batch, time, {self.n_features}]")  # figures/spock/regression.py:210-211
        if x.shape[1] < 2:
            raise ValueError("the time pool needs at least 2 timesteps (torch.std of one is NaN, spock_reg_model.py:419)")
        return x

    def _next_philox_id(self, n=1):
        i = self._philox_calls
        self._philox_calls += n
        return i

    def _masked(self, x):
        """Every mask the flags switch on, as forward() applies them (:488-500)."""
        if self.fix_megno or self.fix_megno2:
            x = self.zero_megno(x)
        if not self.include_mmr:
            x = self.zero_mmr(x)
        if not self.include_nan:
            x = self.zero_nan(x)
        if not self.include_eplusminus:
            x = self.zero_eplusminus(x)
        return x

    def _zero(self, x, location):
        """The reference's masks are subtractions (:452-478): x - mask with mask = x on the masked columns -- 0 for a finite value, NaN for
        NaN / +-inf.  (Host-side helpers with the reference's names; forward() does the same inside the kernels)"""
        with torch.no_grad():
            mask = torch.zeros_like(x)
            mask[..., location] = x[..., location].clone()
            return x - mask

    def zero_megno(self, x):
        return self._zero(x, self.megno_location)

    def zero_mmr(self, x):
        return self._zero(x, self.mmr_location)

    def zero_nan(self, x):
        return self._zero(x, self.nan_location)

    def zero_eplusminus(self, x):
        return self._zero(x, self.eplusminus_location)

    def summarize_megno(self, x):
        """[mean_t, std_t] of the raw MEGNO column (:480-484) -> [B, 2]; what fix_megno appends to the summary (the kernels compute it in place)."""
        col = x[:, :, [self.megno_location]]
        return torch.cat([torch.mean(col, 1), torch.std(col, 1)], dim=1)

    def set_flag(self, flag_name, value):
        """(:410-414; this module has no child modules that carry flags)"""
        setattr(self, flag_name, value)

    def _forward_gpu(self, x, W, noisy, want_debug=False, plan=None, record=False):
        """x [B,T,41] on any device, W [1,d] -> (out[B,2] on x.device, pre, summ)."""
        self._check_x(x)
        plan = plan or self._plan()
        dev_in = x.device
        g = _gpu()
        xg = x.detach().to(g, torch.float32).contiguous()
        B, T, F = xg.shape
        eps = eps_in = eps_sum = None
        if self.rng == "torch":
            # the reference's draws, in its order, on its devices (:445, :426-427, :449)
            if noisy:
                eps_in = torch.randn_like(x.detach().float())
            e1 = torch.randn(B, self._latent, device=dev_in)
            e2 = torch.randn(B, self._latent, device=dev_in)
            if noisy:
                eps_sum = torch.randn(B, self._summary_width, device=dev_in)
            eps = torch.stack([e1, e2], dim=1)[None].to(g).contiguous()
            if noisy:
                eps_in = eps_in[None].to(g).contiguous()
                eps_sum = eps_sum[None].to(g).contiguous()
            did = 0
        else:
            did = self._next_philox_id()
        Wg = W.to(g)
        seed = int(self.philox_seed)
        if record:   # what the lazily evaluated side effect (_cur_summary, :512) needs to re-run this forward with its debug outputs
            self._last_forward = (xg, Wg, eps, eps_in, eps_sum, noisy, did, plan, seed)
            self._cur_summary_cache = None
        self._last_latents_args = (xg, Wg, eps_in, noisy, did, plan, seed)   # self.latents (:433): set by every compute_summary_stats
        self._latents_cache = None
        if want_debug or plan is not self._plan():   # debug outputs / a plan other than the model's own (compute_summary_stats): direct
            res = ops.forward(xg, Wg, eps=eps, eps_in=eps_in, eps_sum=eps_sum, philox_seed=seed, draw_id0=did, plan=plan,
                              debug=want_debug, noisy=noisy, assume_finite=self.assume_finite)
            if want_debug:
                return tuple(r[0].to(dev_in) for r in res)
            return res[0].to(dev_in)
        mask, lowest, net = self._op_args()
        with torch.cuda.device(g):
            res = self._tops().forward(xg, Wg, eps, eps_in, eps_sum, 1, bool(noisy), seed, int(did), 0, mask, lowest, net, bool(self.assume_finite))
        return res[0].to(dev_in)

    # ---- reference API -----------------------------------------------------------------------------------------
    def compute_summary_stats(self, x):
        """feature_nn -> mean/std time pool with sampled moments (:416-435).  x is used as given (no masking)."""
        # the reference applies the masks in forward(), not here: run the kernel with an empty mask; the MEGNO statistics of
        # fix_megno are appended by forward() (:509-510), not by this method (:416-435)
        plan = self._plan(zero_mask=0)
        _, _, summ = self._forward_gpu(x, self._w[None], noisy=False, want_debug=True, plan=plan)
        return summ[:, :2 * self._latent]

    def predict_instability(self, summary_stats):
        """regress_nn + soft_clamp (:437-442) on an explicit summary -> (mu [B,1], std [B,1]) on its device."""
        g = _gpu()
        s = summary_stats.detach().to(g, torch.float32).contiguous()
        out = ops.regress(s[None], self._w[None].to(g), plan=self._plan())[0].to(summary_stats.device)
        return out[:, [0]], out[:, [1]]

    def add_input_noise(self, x):
        """x + randn_like(x) * exp(input_noise_logvar / 2) (:444-446); forward() fuses this step into the kernel."""
        lv = self._w[:self.n_features].to(x.device)
        return x + torch.randn_like(x) * torch.exp(lv[None, None, :] / 2)

    def add_summary_noise(self, summary_stats):
        """summary + randn_like(summary) * exp(summary_noise_logvar / 2) (:448-450); forward() fuses this step."""
        lv = self._w[self.n_features:self.n_features + self._summary_width].to(summary_stats.device)
        return summary_stats + torch.randn_like(summary_stats) * torch.exp(lv[None, :] / 2)

    def forward(self, x, noisy_val=True):
        """VarModel.forward (:486-528) with the currently loaded weights -> cat(mu, std) [B,2] on x.device."""
        if self.random_sample:   # `x = self.augment(x)` (:502-503), behind the column masks and in front of the input noise: the masks are
            if self.fix_megno:   # per column, so they commute with the choice of timesteps -- the raw MEGNO statistics of fix_megno do not
                raise NotImplementedError("random_sample with fix_megno: the reference summarises MEGNO over the WHOLE series (:488-491) and "
                                          "pools the augmented one; not built")
            x = self.augment(x)
        return self._forward_gpu(x, self._w[None], noisy=bool(noisy_val), record=True)

    def augment(self, x):
        """"This randomly samples times." (:404-408): a random number (hparams['samp'] .. T) of timesteps, drawn WITH replacement from numpy's
        global generator -- the same two np.random.randint calls as the reference; the series the kernels then see has that length
        (any T >= 2 runs: the generic engine / the pretrained network's embedded forms)."""
        samples = np.random.randint(self.hparams["samp"], x.shape[1] + 1)
        return x[:, np.random.randint(0, x.shape[1], size=samples)]

    def sample(self, x, samples=10):
        """VarModel.sample (:530-545): mean over `samples` noisy forwards of mu + N(0,1)*std -> float64 ndarray [B].
        rng = "torch": the reference's loop, call for call (CPU generators).  rng = "philox": the `samples` noisy forwards are
        ONE launch (one output row per sample, all noise in-kernel); numpy's randn(B) per sample is drawn as the reference does."""
        x = x.cpu()
        init_device = self._device
        init_random_sample, self.random_sample = self.random_sample, False   # (:532-535, restored at :543)
        try:
            return self._sample(x, samples, init_device)
        finally:
            self.random_sample = init_random_sample

    def _sample(self, x, samples, init_device):
        if self.rng == "philox":
            self._check_x(x)
            g = _gpu()
            xg = x.detach().to(g, torch.float32).contiguous()
            W = self._w[None].to(g).expand(samples, -1).contiguous()
            out = ops.forward(xg, W, philox_seed=self.philox_seed, draw_id0=self._next_philox_id(samples), plan=self._plan(),
                              noisy=True, assume_finite=self.assume_finite).cpu().numpy()   # [samples, B, 2]
            all_samp = [out[s, :, 0] + np.random.randn(out.shape[1]) * out[s, :, 1] for s in range(samples)]
            return np.average(all_samp, axis=0)
        self.cpu()  # the reference forces CPU here, so its noise comes from the CPU generators
        all_samp = []
        for _ in range(samples):
            out = self(x).detach().numpy()
            mu, std = out[:, 0], out[:, 1]
            all_samp.append(mu + np.random.randn(len(out)) * std)
        self.to(init_device)
        return np.average(all_samp, axis=0)


class GenericEngineWarning(UserWarning):
    """A network other than the pretrained shapes is running un-specialised (DESIGN.md sections 4.9 / 4.10)."""


class SWAGModel(VarModel):
    """SWAG posterior over the weights (reference :689-908), inference half."""

    def init_params(self, swa_params):
        self.swa_params = swa_params
        self.swa_params.setdefault("swa_lr", 0.001)
        self.swa_params.setdefault("swa_start", 1000)
        self.swa_params.setdefault("swa_recording_lr_factor", 0.5)
        self.n_models = 0
        self.w_avg = None
        self.w2_avg = None
        self.pre_D = None
        self.K = self.swa_params.get("K", 20)
        self.c = self.swa_params.get("c", 2)
        self.swa_params["c"] = self.c
        self.swa_params["K"] = self.K
        return self

    def _state_gpu(self):
        """(w_avg [1,d], w2_avg [1,d], pre_D [1,d,K]) on the GPU.  The copy is kept while the three attributes stay the same tensor objects at
        the same version (in-place edits and reassignments -- the scripts reassign them with .cuda() / .cpu() copies, regression.py:82-90 --
        invalidate it, and so does `set_()`: the storage address is part of the key; writes through `.data` bypass torch's version counter and do NOT
        -- call invalidate_state() or reassign the attribute after such an edit; inference tensors have no version and are never cached): a
        model whose state lives in host memory (cuda=False) would otherwise send its 0.97 MB across PCIe on every sample_full_swag call
        (0.27 ms of a 0.45 ms call)."""
        g = _gpu()
        if self.w_avg is None:
            raise RuntimeError("SWAG state (w_avg, w2_avg, pre_D) is not set")
        src = (self.w_avg, self.w2_avg, self.pre_D)
        f = lambda t: t.detach().to(g, torch.float32).contiguous()
        try:   # (id, version, storage address, device): reassignment, in-place edits and set_() all change the key
            key = tuple((id(t), t._version, t.data_ptr(), t.device) for t in src) + (g,)
        except RuntimeError:   # inference tensors (state assigned under torch.inference_mode()) have no version counter: no cache for them
            self.__dict__.pop("_state_cache", None)
            return f(src[0])[None], f(src[1])[None], f(src[2])[None]
        hit = self.__dict__.get("_state_cache")
        if hit is None or hit[0] != key:
            hit = (key, src, (f(src[0])[None], f(src[1])[None], f(src[2])[None]))   # (src is held: the ids in the key stay unique)
            self.__dict__["_state_cache"] = hit
        return hit[2]

    def invalidate_state(self):
        """Drop the cached GPU copy of (w_avg, w2_avg, pre_D): call after editing them through `.data` (the one kind of write the cache key
        cannot see)."""
        self.__dict__.pop("_state_cache", None)

    def _draw_noise(self):
        """z_1 = randn((1,d)), z_2 = randn((K,1)) on self.device (:830-831)."""
        d = self.w_avg.shape[0]
        z1 = torch.randn((1, d), device=self._device)
        z2 = torch.randn((self.K, 1), device=self._device)
        return z1, z2

    def sample_weights(self, scale=1):
        """w ~ N(w_avg, scale^2 (diag/2 + D D^T / 2(K-1))) (:815-838), loaded into the model."""
        wa, w2, pd = self._state_gpu()
        g = wa.device
        idx = torch.zeros(1, dtype=torch.int32, device=g)
        mask, lowest, net = self._op_args()
        with torch.cuda.device(g):
            if self.rng == "torch":
                z1, z2 = self._draw_noise()
                W = self._tops().swag_draw(wa, w2, pd, idx, z1.to(g).contiguous(), z2.reshape(1, -1).to(g).contiguous(), float(scale), 0, 0,
                                           mask, lowest, net)
            else:
                W = self._tops().swag_draw(wa, w2, pd, idx, None, None, float(scale), int(self.philox_seed), int(self._next_philox_id()),
                                           mask, lowest, net)
        self.load(W[0])

    def forward_swag(self, x, scale=0.5):
        """Same output as forward_swag_fast; the reference's extra `_summary_kl` bookkeeping (:866-871) is training-only."""
        return self.forward_swag_fast(x, scale=scale)

    def forward_swag_fast(self, x, scale=0.5):
        """Sample weights, then forward without input/summary noise (:878-908), in ONE fused kernel.
        The scripts call this once per chunk per sample (figures/multiswag_5_planet.py:295-298: 15 rows a call), so the host path is kept
        short: nothing is converted or copied that already is what the kernel takes (_as_gpu_f32), the per-device constants are made once
        (_idx0), the op's arguments once per model (_op_args), and the plan is looked up only if somebody reads the sampled weights."""
        self._check_x(x)
        dev_in = x.device
        wa, w2, pd = self._state_gpu()
        g = wa.device
        xg = _as_gpu_f32(x, g)
        B = xg.shape[0]
        idx = _idx0(g)
        mask, lowest, net = self._op_args()
        here = _N.current_device() == g.index
        if self.rng == "torch":
            L_, d_ = self._latent, self.w_avg.shape[0]
            if self._device.type == "cpu" and dev_in.type == "cpu":
                # the reference's four draws, in its order (:830-831, :426-427), each straight into its slice of ONE host buffer
                # (normal_() on a contiguous view consumes the generator exactly like torch.randn of that shape) -> one copy to the GPU
                buf = torch.empty(d_ + self.K + 2 * B * L_)
                z1v, z2v = buf[:d_].view(1, d_), buf[d_:d_ + self.K].view(self.K, 1)
                ev = buf[d_ + self.K:].view(2, B, L_)
                z1v.normal_(); z2v.normal_(); ev[0].normal_(); ev[1].normal_()
                bg = buf.to(g)
                z1g, z2g = bg[:d_].view(1, d_), bg[d_:d_ + self.K].view(1, self.K)
                eps = bg[d_ + self.K:].view(2, B, L_).permute(1, 0, 2).contiguous()[None]
            else:
                if dev_in == g and self._device == g and _strided_normal_is_randn(g):
                    # Model, x and generator all on this GPU (FeatureRegressor(cuda=True), the scripts' per-chunk loop): the reference's four
                    # draws (:830-831, :426-427) go into buffers this MODEL keeps per (stream, batch size) -- randn(out=) and normal_() on
                    # views made once draw what torch.randn of those shapes draws (the second: checked once per device) -- so a call
                    # allocates nothing and slices nothing for its noise.  The buffers belong to one model and one stream: the next call
                    # of this model on this stream overwrites them in stream order, behind the kernel that read them; the weights the
                    # module "has loaded" (:838) are re-drawn lazily from them, and are the LAST call's, as in the reference.
                    st = _N.stream_ptr(g.index)
                    bufs = self.__dict__.setdefault("_noise_bufs", {})
                    nb = bufs.get((g, st, B))
                    if nb is None:
                        if len(bufs) >= 4:      # (a ragged last chunk alternates two batch sizes: both stay; an unbounded series does not pile up)
                            bufs.clear()
                        eps = torch.empty((1, B, 2, L_), dtype=torch.float32, device=g)
                        z2 = torch.empty((self.K, 1), dtype=torch.float32, device=g)
                        nb = bufs[(g, st, B)] = (torch.empty((1, d_), dtype=torch.float32, device=g), z2, eps, eps[0, :, 0], eps[0, :, 1], z2.view(1, self.K))
                    z1g, z2, eps, ev0, ev1, z2g = nb
                    torch.randn((1, d_), out=z1g)                             # :830
                    torch.randn((self.K, 1), out=z2)                          # :831
                    ev0.normal_()                                             # :426  randn_like([B, latent])
                    ev1.normal_()                                             # :427
                else:
                    z1, z2 = self._draw_noise()                              # :830-831
                    e1 = torch.randn(B, L_, device=dev_in)                   # :426
                    e2 = torch.randn(B, L_, device=dev_in)                   # :427
                    eps = _as_gpu_f32(torch.stack((e1, e2), dim=1), g)[None]
                    z1g, z2g = _as_gpu_f32(z1, g), _as_gpu_f32(z2.reshape(1, -1), g)
            if here:
                out = self._tops().multiswag(xg, wa, w2, pd, idx, z1g, z2g, eps, 1, float(scale), 0, 0, 0, mask, lowest, net, bool(self.assume_finite))
            else:
                with torch.cuda.device(g):
                    out = self._tops().multiswag(xg, wa, w2, pd, idx, z1g, z2g, eps, 1, float(scale), 0, 0, 0, mask, lowest, net,
                                                 bool(self.assume_finite))
            # the reference leaves the sampled weights loaded in the module (:838): drawn on demand, from the same normals
            draw = lambda: ops.swag_draw(wa, w2, pd, idx, z1g, z2g, scale=scale, plan=self._plan())[0]
        else:
            did, seed = self._next_philox_id(), self.philox_seed
            with torch.cuda.device(g):
                out = self._tops().multiswag(xg, wa, w2, pd, idx, None, None, None, 1, float(scale), int(seed), int(did), 0, mask, lowest, net,
                                             bool(self.assume_finite))
            draw = lambda: ops.swag_draw(wa, w2, pd, idx, scale=scale, philox_seed=seed, draw_id0=did, plan=self._plan())[0]
        self._pending_draw = draw
        self._last_forward = None          # (_cur_summary belongs to forward(); forward_swag_fast does not set it, :878-908)
        self._cur_summary_cache = None
        self._last_latents_args = (xg, draw, None, False, 0, None, 0)   # its compute_summary_stats call (:893) sets self.latents
        self._latents_cache = None
        return out[0] if dev_in == g else out[0].to(dev_in)


def save_swag(swag_model, path):
    """spock_reg_model.py:911-920."""
    write_swag_file(path, swag_model.hparams, swag_model.swa_params, swag_model.w_avg, swag_model.w2_avg, swag_model.pre_D)


def load_swag(path):
    """spock_reg_model.py:922-967: checkpoint -> SWAGModel with w_avg / w2_avg / pre_D attached and, for 'v50' paths,
    the hard-coded StandardScaler; otherwise `<path minus .pkl>_ssX.pkl` is read if present."""
    items = read_swag_file(path)
    swag_model = SWAGModel(items["hparams"]).init_params(dict(items["swa_params"]))
    swag_model.w_avg = items["w_avg"]
    swag_model.w2_avg = items["w2_avg"]
    swag_model.pre_D = items["pre_D"]
    if "v50" in str(path):
        swag_model.ssX = v50_scaler()
    else:
        ssX_file = str(path)[:-4] + "_ssX.pkl"
        try:
            import pickle as pkl
            with open(ssX_file, "rb") as f:
                swag_model.ssX = pkl.load(f)  # an sklearn StandardScaler written by run_swag.py:96-97
        except FileNotFoundError:
            print(f"ssX file not found! {ssX_file}")
    return swag_model


__all__ = ["EPSILON", "VarModel", "SWAGModel", "GenericEngineWarning", "StandardScaler", "load_swag", "save_swag", "soft_clamp", "mlp",
           "AttributeDict", "copy"]
