"""Drop-in for the hot-path half of figures/spock/regression.py: FeatureRegressor.__init__ (:36-72) and
sample_full_swag (:74-92), plus a batched driver for the MC loops that call it
(figures/multiswag_5_planet.py:295-298, figures/main_figures.py:154-156, figures/spock/regression.py:149).

`FeatureRegressor.sample(sim)` / `.predict(sim)` need REBOUND N-body features upstream of the path
 and are not built.
"""
import glob
import os

import numpy as np
import torch

from . import ops
from . import spock_reg_model
from .multidevice import DeviceSet
from .spock_reg_model import _gpu


def data_setup_kernel(mass_array, cur_tseries):
    """figures/spock/regression.py:183-213 on the GPU: mass_array [3], cur_tseries [1,T,26] -> X [1,T,41] float64 ndarray.
    (`pack_features` below is the batched, fused form that feeds the network directly.)"""
    ts = np.asarray(cur_tseries, dtype=np.float64)
    if ts.ndim != 3 or ts.shape[-1] != 26:
        raise NotImplementedError("Need to change indexes above for angles, replace ssX.")
    mass = np.tile(np.asarray(mass_array, dtype=np.float64)[None], (ts.shape[0], 1))
    return ops.feature_pack(ts, mass).cpu().numpy()


def pack_features(tseries, mass, ssX=None):
    """tseries [N,T,26], mass [N,3] float64 -> standardised float32 x [N,T,41] on the GPU:
    data_setup_kernel + ssX.transform + .float() (regression.py:143-145) in one kernel."""
    ssX = ssX or spock_reg_model.v50_scaler()
    return ops.feature_pack(tseries, mass, mean=ssX.mean_, scale=ssX.scale_)


def draw_reference_noise(samples, parts, S, d, K, latent, noise_dev, eps_dev):
    """The generator draws of the scripts' MC loop (figures/multiswag_5_planet.py:295-298 around regression.py:74-92), in the
    reference's order, for `samples` x len(parts) calls of sample_full_swag: per call np.random.randint(0, S) (regression.py:78),
    randn((1, d)), randn((K, 1)) (spock_reg_model.py:830-831) and the two randn_like of the chunk's [n, latent] pool
    (:426-427).  Every draw lands in its row of a preallocated tensor (normal_() on a contiguous view: the same generator consumption as a
    fresh tensor of that shape, no temporary, no per-draw copy) -> seed_idx [J] int32 (numpy), z1 [J, d], z2 [J, K] on noise_dev,
    eps [samples, 2, B, latent] on eps_dev -- kind-major, so that a chunk's rows are one contiguous block; the kernels take
    eps.permute(0, 2, 1, 3)."""
    nch = len(parts)
    J = samples * nch
    B = sum(len(pp) for pp in parts)
    seed_idx = np.empty(J, np.int32)
    z1 = torch.empty((J, 1, d), device=noise_dev)
    z2 = torch.empty((J, K, 1), device=noise_dev)
    eps = torch.empty((samples, 2, B, latent), device=eps_dev)
    # views made once (one unbind / narrow each), filled with Tensor.normal_() -- what torch.randn(shape) is underneath: an empty tensor
    # of that shape + normal_() -- so the generator is consumed exactly as by the reference's calls
    z1v, z2v = z1.unbind(0), z2.unbind(0)
    lo, ev = 0, []
    for pp in parts:
        ev.append([(eps[s_, 0, lo:lo + len(pp)], eps[s_, 1, lo:lo + len(pp)]) for s_ in range(samples)])
        lo += len(pp)
    randint = np.random.randint
    for e in range(J):
        s_, c_ = divmod(e, nch)
        seed_idx[e] = randint(0, S)                                     # regression.py:78
        z1v[e].normal_()                                                # spock_reg_model.py:830  randn((1, d))
        z2v[e].normal_()                                                # :831                    randn((K, 1))
        e1, e2 = ev[c_][s_]
        e1.normal_()                                                    # :426  randn_like([n, latent])
        e2.normal_()                                                    # :427
    return seed_idx, z1.view(J, d), z2.view(J, K), eps


class FeatureRegressor(object):
    def __init__(self, cuda=False, filebase="long_zero_megno_with_angles_power_v14_*_output.pkl", sort=False, devices=None):
        """filebase is resolved like the reference (relative to this file's directory + '/../'); absolute globs work too.
        The reference's ensemble order is the unsorted glob order (regression.py:45); sort=True makes it deterministic.
        devices (not in the reference): the GPUs the batched drivers below use when their own `devices` argument is None -- None = the
        current device, "all" = every visible GPU from this one process (multidevice.py; BNN_CHAOS_DEVICES=all in the environment
        does the same for a script that cannot be edited), an int or a list."""
        super(FeatureRegressor, self).__init__()
        pwd = os.path.dirname(__file__)
        self.cuda = cuda
        self.devices = devices
        self.last_run = None
        pattern = filebase if os.path.isabs(filebase) else pwd + "/../" + filebase
        names = glob.glob(pattern)
        if sort:
            names = sorted(names)
        self.swag_ensemble = [spock_reg_model.load_swag(fname).cpu() for fname in names]
        self.ssX = spock_reg_model.v50_scaler()  # "Assume fixed scale" (regression.py:47-71)
        self._stacked = None
        self._device_sets = {}
        self._cpu_state = None

    # ---- every visible GPU from this one process (multidevice.py) ------------------------------------------------
    def specialize(self, noisy=(False,), w8=None):
        """Not in the reference: compile the ensemble's network into its own form of the generic forward engine (VarModel.specialize;
        the members share one network, hence one plan per device and column mask).  The MC drivers are quiet forwards: noisy=(False,)
        by default.  Returns self."""
        for m in self.swag_ensemble:
            m.specialize(noisy=noisy, w8=w8)
        return self

    def device_set(self, devices=None):
        """devices: None = the constructor's `devices` (default: the CURRENT device -- a script that chose a GPU, or a rank of a
        process-per-GPU launch, keeps it); "all" = every visible GPU from this one process; an int n = the first n; a list of indices /
        torch.devices = exactly those (a device may be named several times: logical shards on one card)."""
        if devices is None:
            devices = self.devices
        key = (None, torch.cuda.current_device()) if devices is None else (devices if isinstance(devices, (int, str)) else tuple(str(d) for d in devices))
        if key not in self._device_sets:
            self._device_sets[key] = DeviceSet(devices)
        return self._device_sets[key]

    def _ensemble_cpu(self):
        if self._cpu_state is None or self._cpu_state[0] != len(self.swag_ensemble):
            f = lambda name: torch.stack([getattr(m, name).detach().float().cpu() for m in self.swag_ensemble]).contiguous()
            self._cpu_state = (len(self.swag_ensemble), (f("w_avg"), f("w2_avg"), f("pre_D")))
        return self._cpu_state[1]

    def _check_X(self, X):
        nf = self.swag_ensemble[0].n_features if self.swag_ensemble else 41
        if X.dim() != 3 or X.shape[-1] != nf:
            raise NotImplementedError(f"X must be [B, T, {nf}]")

    # ---- reference API -----------------------------------------------------------------------------------------
    def sample_full_swag(self, X_sample):
        """Pick a random model from the ensemble and sample from it (regression.py:74-92)."""
        swag_i = np.random.randint(0, len(self.swag_ensemble))
        swag_model = self.swag_ensemble[swag_i]
        swag_model.eval()
        if self.cuda:
            # the reference shuttles the model and its SWAG state to the GPU and back per call (regression.py:81-91);
            # here only the device of the noise draws matters (row R of SURVEY.md section 8).
            swag_model.cuda()
        out = swag_model.forward_swag_fast(X_sample, scale=0.5)
        if self.cuda:
            swag_model.cpu()   # (a device label while the call's draw is pending: nothing is copied, spock_reg_model.VarModel.to)
        return out

    def sample(self, sim, indices=None, samples=1000):
        raise NotImplementedError("needs REBOUND feature generation (get_extended_tseries), upstream of the accelerated path; "
                                  "give its output to sample_tseries()")

    def sample_tseries(self, tseries, mass_arrays, samples=1000, rng="torch", philox_seed=0):
        """FeatureRegressor.sample (regression.py:110-179) downstream of the N-body integration:
        tseries [trios, Nout, 26] (get_extended_tseries' output; every 10th step is used, :141), mass_arrays [trios, 3]
        -> (mu, std) ndarrays [trios, samples], drawing the generators exactly like the loop at :149
        (`samples` calls of sample_full_swag with B = 1 per trio), one launch per trio."""
        tseries = np.asarray(tseries, dtype=np.float64)
        alltime = []
        for i in range(tseries.shape[0]):
            cur_tseries = tseries[None, i, ::10]                                  # :141
            X = pack_features(cur_tseries, np.asarray(mass_arrays[i], dtype=np.float64)[None], self.ssX)  # :143-145
            if not self.cuda:
                X = X.cpu()
            time = self.sample_full_swag_many(X, samples=samples, chunks=1, rng=rng, philox_seed=philox_seed,
                                              draw_id0=i * samples)              # :149
            alltime.append(time.detach().cpu().numpy())
        out = np.array(alltime)[..., 0, :]                                        # :159
        return out[..., 0], out[..., 1]

    def predict(self, sim, indices=None, samples=1000):
        raise NotImplementedError("needs REBOUND feature generation (get_extended_tseries), upstream of the accelerated path; "
                                  "give its output to predict_tseries()")

    def predict_tseries(self, tseries, mass_arrays, samples=1000, rng="torch", philox_seed=0):
        """FeatureRegressor.predict (regression.py:94-108) downstream of the N-body integration: the reference returns
        `np.median(self.sample(...))`, i.e. the median over the stacked (mu, std) arrays that `sample` returns (:107-108, :179) --
        reproduced as it stands."""
        return np.median(self.sample_tseries(tseries, mass_arrays, samples=samples, rng=rng, philox_seed=philox_seed))

    # ---- batched driver ----------------------------------------------------------------------------------------
    def ensemble_state(self, device=None):
        """w_avg [S,d], w2_avg [S,d], pre_D [S,d,K] of the whole ensemble, resident on the GPU (29 MB for 30 seeds)."""
        dev = _gpu() if device is None else torch.device(device)
        if self._stacked is None or self._stacked[0].device != dev:
            self._stacked = tuple(t.to(dev).contiguous() for t in self._ensemble_cpu())
        return self._stacked

    def sample_full_swag_many(self, X, samples, chunks=1, rng="torch", philox_seed=0, draw_id0=0, system_id0=0,
                              scale=0.5, out=None, precision="f32", devices=None, assume_finite=False):
        """The whole MC loop in one launch per device:

            torch.cat([torch.cat([self.sample_full_swag(Xpart) for Xpart in torch.chunk(X, chunks)])[None]
                       for _ in range(samples)])          # figures/multiswag_5_planet.py:295-298

        -> [samples, B, 2].  rng="torch" consumes numpy's and torch's global generators exactly as that loop does
        (one randint + randn((1,d)) + randn((K,1)) + 2 randn_like([Bc,latent]) per chunk per sample);
        rng="philox" draws the seed picks from numpy and everything else in-kernel.
        devices: None = the current device (or the constructor's `devices`); "all" = every visible GPU of this process (the systems
        are sharded over them, the ensemble and the draws replicated; the chunks stay those of the whole batch, so the result does not
        depend on the device list: multidevice.py); an int or a list picks devices.  Host-resident X: every device's rows are on their
        way before the first launch, the PCIe links side by side (DeviceSet.stage); `self.last_run` holds h2d / exchange / devices.
        assume_finite: False = X is scanned once per shard and systems that hold NaN / +-inf get the reference's result (ops.forward).
        precision: "f32" (the parity path) or an OPT-IN reduced-precision form of ops.forward ("f16x3": fp32-level error at ~1.7x
        the throughput; "bf16", "f16", ...: approximate) -- DESIGN.md section 4.6.  The IEEE-half forms ("f16", "f16x3") require
        |X| < 65 504 in the live columns: rows beyond that (e.g. the script's constant-4 fill of unstable systems,
        figures/multiswag_5_planet.py:215) get finite but WRONG outputs; this method checks and warns (RuntimeWarning, with the
        number of rows); use a bfloat16 form ("bf16x6": fp32 range and fp32-level error) or "f32" for such inputs."""
        self._check_X(X)
        ds = self.device_set(devices)
        m0 = self.swag_ensemble[0]
        state = ds.replicate("ensemble", self._ensemble_cpu())
        S, d, K = state[0][2].shape
        LAT = m0._latent
        B = X.shape[0]
        parts = torch.chunk(torch.arange(B), chunks) if B else []
        nch = len(parts)  # torch.chunk may return fewer chunks than asked
        if nch == 0:
            return torch.empty((samples, 0, 2), device=X.device)
        csz = -(-B // nch)
        if any(len(pp) != min(csz, B - i * csz) for i, pp in enumerate(parts)):
            raise NotImplementedError("unexpected torch.chunk partition")
        J = samples * nch
        if precision in ops.HALF_FORMS:
            nbad = int(ops.half_range_exceeded(X.detach().float(), m0.zero_mask()).sum())
            if nbad:
                import warnings
                warnings.warn(f"precision={precision!r}: {nbad} of {B} rows hold |x| >= 65504 in a live column; IEEE half saturates there and "
                              "the outputs of those rows are wrong (finite); use 'bf16x6' or 'f32' for them", RuntimeWarning, stacklevel=2)
        g0 = ds.devices[0]
        noise_dev = g0 if self.cuda else torch.device("cpu")
        xs = ds.stage(X)            # every device's rows are on their way before the generators are even touched
        z1 = z2 = eps = None
        if rng == "torch":
            seed_idx, z1, z2, eps = draw_reference_noise(samples, parts, S, d, K, LAT, noise_dev, X.device)
        elif rng == "philox":
            seed_idx = np.empty(J, np.int32)
            for e in range(J):
                seed_idx[e] = np.random.randint(0, S)
        else:
            raise ValueError("rng must be 'torch' or 'philox'")
        seed_t = torch.as_tensor(seed_idx)
        rep = {}    # the draws' normals: ONE copy per distinct device (logical shards on one card share it)

        def shard(i, dev, lo, hi):
            wa, w2, pd = state[i]
            kw = dict(nchunks=nch, scale=scale, plan=m0._plan(device=dev), precision=precision, chunk_B=B, chunk_off=lo,
                      assume_finite=assume_finite)
            if rng == "torch":
                if dev not in rep:
                    rep[dev] = (z1.to(dev).contiguous(), z2.to(dev).contiguous())
                e_dev = eps[:, :, lo:hi].to(dev).permute(0, 2, 1, 3).contiguous()     # [samples, 2, n, L] -> the kernels' [samples, n, 2, L]
                return ops.multiswag(xs[i], wa, w2, pd, seed_t, rep[dev][0], rep[dev][1], e_dev, **kw)
            return ops.multiswag(xs[i], wa, w2, pd, seed_t, philox_seed=philox_seed, draw_id0=draw_id0, system_id0=system_id0 + lo, **kw)

        res = [r for r in ds.run(B, shard) if r is not None]
        ds.release_sources()   # (a pinned X may be refilled by the caller from here on)
        self.last_run = {"devices": [str(dv) for dv in ds.devices], "exchange": "concatenation of the shards' rows", "h2d": ds.h2d_ms}
        target = out.device if out is not None else X.device
        nb = torch.device(target).type == "cuda"   # (a non-blocking copy to HOST memory would return before the data has landed)
        res = res[0].to(target) if len(res) == 1 else torch.cat([r.to(target, non_blocking=nb) for r in res], 1)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def predictive_bands(self, X, samples, chunks=1, trios=1, q=(50.0, 84.0, 16.0, 97.5, 2.5), philox_seed=0, system_id0=0,
                         samples_per_launch=64, scale=0.5, stats=None, segments=None, devices=None, assume_finite=False):
        """Everything figures/multiswag_5_planet.py does between the features and the `cleaned` table (:295-298, 388-428,
        484-489), streamed: the MC loop (one random ensemble member + one weight draw per chunk per sample), the truncated-normal
        draw, the prior resampling past 9, the min over `trios` consecutive rows and, per simulation, the percentiles `q`
        (default: median, l, u, ll, uu) and the average -- with neither [samples, B, 2] nor [samples, B] in memory beyond
        `samples_per_launch` samples: the epilogue runs in the forward kernel's tail, a quantile sketch (ops.QuantileSketch,
        one bin width of error) collects the draws.  Seed picks come from numpy's generator (regression.py:78, one per chunk per
        sample, in the reference's order); all other noise is in-kernel Philox keyed by (philox_seed, sample, row).
        devices: as in sample_full_swag_many -- with several, WHOLE simulations are sharded over them, each device streams its shard
        into its own sketch, and ONE exchange assembles the [simulations, len(q) + 1] table on the first device.
        Returns {"percentiles": [B / trios, len(q)], "average": [B / trios]} on the GPU, plus "exchange" (which exchange ran) and
        "h2d" (a callable: the staging's copy time, DeviceSet.h2d_ms)."""
        self._check_X(X)
        if stats is not None and len(self.device_set(devices)) > 1:
            raise ValueError("an explicit `stats` block lives on one device; leave it None when several devices are used")
        ds = self.device_set(devices)
        m0 = self.swag_ensemble[0]
        state = ds.replicate("ensemble", self._ensemble_cpu())
        S = state[0][2].shape[0]
        B = X.shape[0]
        nch = len(torch.chunk(torch.arange(B), chunks)) if B else 1
        seed_idx = torch.as_tensor(np.array([np.random.randint(0, S) for _ in range(samples * nch)], np.int32))   # one pick per chunk per sample
        sketches = [None] * len(ds)
        xs = ds.stage(X, group=trios)     # every device's rows on their way before the first launch

        def shard(i, dev, lo, hi):
            wa, w2, pd = state[i]
            sk = ops.QuantileSketch(hi - lo, group=trios, segments=segments, device=dev)
            st = stats or ops.stats_params(device=dev)
            ops.multiswag_bands(xs[i], wa, w2, pd, seed_idx, sk, st=st, nchunks=nch, scale=scale, philox_seed=philox_seed,
                                system_id0=system_id0 + lo, draws_per_launch=samples_per_launch * nch, plan=m0._plan(device=dev),
                                chunk_B=B, chunk_off=lo, assume_finite=assume_finite)
            sketches[i] = sk
            return torch.cat([sk.percentiles(q), sk.mean().float()[:, None]], 1)

        table = ds.gather_rows(ds.run(B, shard, group=trios))
        ds.release_sources()   # (a pinned X may be refilled by the caller from here on)
        nq = len(tuple(q))
        live = [sk for sk in sketches if sk is not None]
        self.last_run = {"devices": [str(dv) for dv in ds.devices], "exchange": ds.last_exchange, "h2d": ds.h2d_ms}
        return {"percentiles": table[:, :nq], "average": table[:, nq], "sketch": live[0] if len(live) == 1 else live,
                "exchange": ds.last_exchange, "h2d": ds.h2d_ms}
