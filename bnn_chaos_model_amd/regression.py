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


class FeatureRegressor(object):
    def __init__(self, cuda=False, filebase="long_zero_megno_with_angles_power_v14_*_output.pkl", sort=False):
        """filebase is resolved like the reference (relative to this file's directory + '/../'); absolute globs work too.
        The reference's ensemble order is the unsorted glob order (regression.py:45); sort=True makes it deterministic."""
        super(FeatureRegressor, self).__init__()
        pwd = os.path.dirname(__file__)
        self.cuda = cuda
        pattern = filebase if os.path.isabs(filebase) else pwd + "/../" + filebase
        names = glob.glob(pattern)
        if sort:
            names = sorted(names)
        self.swag_ensemble = [spock_reg_model.load_swag(fname).cpu() for fname in names]
        self.ssX = spock_reg_model.v50_scaler()  # "Assume fixed scale" (regression.py:47-71)
        self._stacked = None

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
            swag_model.cpu()
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
            f = lambda name: torch.stack([getattr(m, name).detach().float().cpu() for m in self.swag_ensemble]).to(dev).contiguous()
            self._stacked = (f("w_avg"), f("w2_avg"), f("pre_D"))
        return self._stacked

    def sample_full_swag_many(self, X, samples, chunks=1, rng="torch", philox_seed=0, draw_id0=0, system_id0=0,
                              scale=0.5, out=None, precision="f32"):
        """The whole MC loop in one launch:

            torch.cat([torch.cat([self.sample_full_swag(Xpart) for Xpart in torch.chunk(X, chunks)])[None]
                       for _ in range(samples)])          # figures/multiswag_5_planet.py:295-298

        -> [samples, B, 2].  rng="torch" consumes numpy's and torch's global generators exactly as that loop does
        (one randint + randn((1,d)) + randn((K,1)) + 2 randn_like([Bc,20]) per chunk per sample);
        rng="philox" draws the seed picks from numpy and everything else in-kernel.
        precision: "f32" (the parity path) or an OPT-IN reduced-precision form of ops.forward ("f16x3": fp32-level error at ~1.7x
        the throughput; "bf16", "f16", ...: approximate) -- DESIGN.md section 4.6.  The IEEE-half forms ("f16", "f16x3") require
        |X| < 65 504 in the live columns: rows beyond that (e.g. the script's constant-4 fill of unstable systems,
        figures/multiswag_5_planet.py:215) get finite but WRONG outputs; this method checks and warns (RuntimeWarning, with the
        number of rows); use a bfloat16 form ("bf16x6": fp32 range and fp32-level error) or "f32" for such inputs."""
        if X.dim() != 3 or X.shape[-1] != 41:
            raise NotImplementedError("X must be [B, T, 41]")
        g = _gpu()
        wa, w2, pd = self.ensemble_state(g)
        S, d, K = pd.shape
        m0 = self.swag_ensemble[0]
        plan = ops.get_plan(m0.zero_mask(), m0.lowest)
        B = X.shape[0]
        parts = torch.chunk(torch.arange(B), chunks) if B else []
        nch = len(parts)  # torch.chunk may return fewer chunks than asked
        if nch == 0:
            return torch.empty((samples, 0, 2), device=X.device)
        csz = -(-B // nch)
        if any(len(pp) != min(csz, B - i * csz) for i, pp in enumerate(parts)):
            raise NotImplementedError("unexpected torch.chunk partition")
        J = samples * nch
        xg = X.detach().to(g, torch.float32).contiguous()
        if precision in ops.HALF_FORMS:
            nbad = int(ops.half_range_exceeded(xg, m0.zero_mask()).sum())
            if nbad:
                import warnings
                warnings.warn(f"precision={precision!r}: {nbad} of {B} rows hold |x| >= 65504 in a live column; IEEE half saturates there and "
                              "the outputs of those rows are wrong (finite); use 'bf16x6' or 'f32' for them", RuntimeWarning, stacklevel=2)
        noise_dev = g if self.cuda else torch.device("cpu")
        seed_idx = np.empty(J, np.int32)
        if rng == "torch":
            z1 = torch.empty((J, d), device=noise_dev)
            z2 = torch.empty((J, K), device=noise_dev)
            eps = torch.empty((samples, B, 2, 20), device=X.device)
            for e in range(J):
                s_, c_ = divmod(e, nch)
                seed_idx[e] = np.random.randint(0, S)                               # regression.py:78
                z1[e] = torch.randn((1, d), device=noise_dev)[0]                    # spock_reg_model.py:830
                z2[e] = torch.randn((K, 1), device=noise_dev)[:, 0]                 # :831
                n = len(parts[c_])
                lo = c_ * csz
                eps[s_, lo:lo + n, 0] = torch.randn(n, 20, device=X.device)         # :426
                eps[s_, lo:lo + n, 1] = torch.randn(n, 20, device=X.device)         # :427
            res = ops.multiswag(xg, wa, w2, pd, torch.as_tensor(seed_idx), z1.to(g).contiguous(), z2.to(g).contiguous(),
                                eps.to(g).contiguous(), nchunks=nch, scale=scale, plan=plan, out=out, precision=precision)
        elif rng == "philox":
            for e in range(J):
                seed_idx[e] = np.random.randint(0, S)
            res = ops.multiswag(xg, wa, w2, pd, torch.as_tensor(seed_idx), nchunks=nch, scale=scale, philox_seed=philox_seed,
                                draw_id0=draw_id0, system_id0=system_id0, plan=plan, out=out, precision=precision)
        else:
            raise ValueError("rng must be 'torch' or 'philox'")
        return res if out is not None else res.to(X.device)

    def predictive_bands(self, X, samples, chunks=1, trios=1, q=(50.0, 84.0, 16.0, 97.5, 2.5), philox_seed=0, system_id0=0,
                         samples_per_launch=64, scale=0.5, stats=None, segments=None):
        """Everything figures/multiswag_5_planet.py does between the features and the `cleaned` table (:295-298, 388-428,
        484-489), streamed: the MC loop (one random ensemble member + one weight draw per chunk per sample), the truncated-normal
        draw, the prior resampling past 9, the min over `trios` consecutive rows and, per simulation, the percentiles `q`
        (default: median, l, u, ll, uu) and the average -- with neither [samples, B, 2] nor [samples, B] in memory beyond
        `samples_per_launch` samples: the epilogue runs in the forward kernel's tail, a quantile sketch (ops.QuantileSketch,
        one bin width of error) collects the draws.  Seed picks come from numpy's generator (regression.py:78, one per chunk per
        sample, in the reference's order); all other noise is in-kernel Philox keyed by (philox_seed, sample, row).
        Returns {"percentiles": [B / trios, len(q)], "average": [B / trios]} on the GPU."""
        if X.dim() != 3 or X.shape[-1] != 41:
            raise NotImplementedError("X must be [B, T, 41]")
        g = _gpu()
        wa, w2, pd = self.ensemble_state(g)
        S = pd.shape[0]
        m0 = self.swag_ensemble[0]
        plan = ops.get_plan(m0.zero_mask(), m0.lowest)
        B = X.shape[0]
        nch = len(torch.chunk(torch.arange(B), chunks)) if B else 1
        xg = X.detach().to(g, torch.float32).contiguous()
        sk = ops.QuantileSketch(B, group=trios, segments=segments, device=g)
        st = stats or ops.stats_params(device=g)
        seed_idx = np.array([np.random.randint(0, S) for _ in range(samples * nch)], np.int32)   # one pick per chunk per sample
        ops.multiswag_bands(xg, wa, w2, pd, torch.as_tensor(seed_idx), sk, st=st, nchunks=nch, scale=scale, philox_seed=philox_seed,
                            system_id0=system_id0, draws_per_launch=samples_per_launch * nch, plan=plan)
        return {"percentiles": sk.percentiles(q), "average": sk.mean().float(), "sketch": sk}
