"""Multi-GPU MultiSWAG: shard the systems, replicate the ensemble and the draw list, gather the moments.

SURVEY.md section 8(e): every (system, draw) evaluation is independent and the time pool is inside a system,
so the path shards BY SYSTEM with no data-path collective.  Each rank evaluates the same J draws (same seed
indices, same Philox seed and global draw / system ids) on its contiguous slice of the systems, reduces its
samples [J, B_local, 2] to per-system predictive moments [B_local, 4] locally, and one all-gather (RCCL over
xGMI; gloo on CPU in the tests) assembles [B, 4] on every rank.  Results are bit-identical for any world size
because the in-kernel noise is keyed by global ids.

One process per GPU: launch with torch.distributed.run; rank r uses cuda:LOCAL_RANK.  The same decomposition from ONE process
(every visible GPU driven by the caller's process, no launcher): multidevice.py, reached through MultiSwagSharded(devices=...) and
FeatureRegressor's batched drivers.
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world, group=1):
    """Contiguous, balanced partition of B systems: list of (lo, hi), the first shards one unit longer.  group > 1 keeps
    `group` consecutive systems (the trios of one simulation) on one rank, so that the min over trios stays local."""
    if int(B) % int(group):
        raise ValueError("B must be a multiple of group")
    base, rem = divmod(int(B) // int(group), int(world))
    out, lo = [], 0
    for r in range(world):
        n = (base + (1 if r < rem else 0)) * int(group)
        out.append((lo, lo + n))
        lo += n
    return out


def all_gather_moments(local, B, group=None, force=False):
    """local [B_r, M] (this rank's slice, in shard_bounds order) -> [B, M] on every rank.  One collective.
    force: issue the collective even at world size 1 (a self-gather; lets a one-GPU box exercise the RCCL call)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return local
    bounds = shard_bounds(B, world)
    nmax = max(hi - lo for lo, hi in bounds)
    M = local.shape[1]
    rank = dist.get_rank(group)
    if local.shape[0] != bounds[rank][1] - bounds[rank][0]:
        raise ValueError("local shard does not match shard_bounds(B, world)[rank]")
    if local.is_cuda and dist.get_backend(group) == "gloo":  # CPU rehearsal of the N>1 path: stage through host memory
        return all_gather_moments(local.cpu(), B, group).to(local.device)
    if nmax == 0:
        return local
    if all(hi - lo == nmax for lo, hi in bounds):
        out = torch.empty((world * nmax, M), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((nmax, M), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * nmax, M), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(bounds)])


def moments_to_mean_std(mom, n_draws):
    """[B,4] sums -> dict of predictive mean/std of mu and mean of std over the draws (float64)."""
    n = float(n_draws)
    mean_mu = mom[:, 0] / n
    var_mu = (mom[:, 1] / n - mean_mu ** 2).clamp_min(0)
    return {"mean_mu": mean_mu, "std_mu": var_mu.sqrt(), "mean_std": mom[:, 2] / n, "rms_std": (mom[:, 3] / n).sqrt()}


def sharded_predictive_moments(local_moments_fn, B, group=None):
    """Run `local_moments_fn(lo, hi) -> [hi-lo, 4]` on this rank's slice and all-gather.

    On the GPU path local_moments_fn is a closure over bnn_chaos_model_amd.ops: multiswag(x[lo:hi], ..., system_id0=lo)
    in slabs of draws, reduced with ops.moments (see MultiSwagSharded)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(B, world)[rank]
    return all_gather_moments(local_moments_fn(lo, hi), B, group)


class MultiSwagSharded:
    """Predictive moments of the dense (systems x draws) MultiSWAG grid on this rank's GPU, gathered over ranks."""

    def __init__(self, w_avg, w2_avg, pre_D, zero_mask=None, lowest_std=0.5, group=None, draws_per_launch=256, devices=None, specialize=False,
                 **arch):
        """devices=None: one process per GPU (torch.distributed), this rank's current device.  devices="all" / an int / a list:
        ONE process drives those GPUs (multidevice.DeviceSet): predictive_moments / predictive_quantiles then take the WHOLE x
        (host or device memory), shard it, and return the assembled table on the first device.
        **arch: n_features / hidden / latent / depth_in / depth_out / fix_megno of a network other than the pretrained one;
        specialize=True compiles that network's own form of the generic engine (ops.specialize; every rank / device finds the code
        objects in the shared cache after the first has written them)."""
        from . import ops
        self.ops = ops
        self.state = (w_avg, w2_avg, pre_D)
        self._mask = ops.V50_ZERO_MASK if zero_mask is None else zero_mask
        self._lowest, self._arch = lowest_std, arch
        self._specialize = bool(specialize)
        self.devset = None
        if devices is not None:
            from .multidevice import DeviceSet
            self.devset = DeviceSet(devices)
        self.plan = None if self.devset else self._plan_on(None)
        self.group = group
        self.draws_per_launch = int(draws_per_launch)

    def _plan_on(self, device):
        plan = self.ops.get_plan(self._mask, self._lowest, device=device, **self._arch)
        if self._specialize and not plan.v50net and not plan.__dict__.get("_spec_req"):
            self.ops.specialize(plan, noisy=(False,))   # the MC drivers are quiet forwards
            plan.__dict__["_spec_req"] = ((False,), None)
        return plan

    def local_moments(self, x_local, seed_idx, philox_seed, system_id0, scale=0.5):
        """x_local [B_r,T,41] on this rank's GPU; seed_idx [J] (identical on every rank) -> float64 [B_r,4].
        One native call: the draws are evaluated in slabs of `draws_per_launch`, samples[J_slab, B_r, 2] stays small at C4 scale."""
        wa, w2, pd = self.state
        return self.ops.multiswag_moments(x_local, wa, w2, pd, seed_idx, scale=scale, philox_seed=philox_seed, system_id0=system_id0,
                                          draws_per_launch=self.draws_per_launch, plan=self.plan)

    def _all_devices(self, x, B_total, group, per_shard):
        """Single-process form: shard the whole x over the device set, run per_shard(x_shard_on_device, state_on_device, plan, lo) on
        every device, ONE exchange."""
        ds = self.devset
        if x.shape[0] != B_total:
            raise ValueError("with devices=..., x is the WHOLE batch")
        states = ds.replicate("state", self.state)
        xs = ds.stage(x, group=group)    # every device's rows on their way before the first launch (pinned / threaded: multidevice.py)

        def shard(i, dev, lo, hi):
            return per_shard(xs[i], states[i], self._plan_on(dev), lo)

        parts = ds.run(B_total, shard, group=group)
        ds.release_sources()   # (a pinned x may be refilled by the caller from here on)
        return ds.gather_rows(parts)

    def predictive_moments(self, x_local, B_total, seed_idx, philox_seed=0, scale=0.5):
        if self.devset is not None:
            return self._all_devices(x_local, B_total, 1, lambda xs, st, plan, lo: self.ops.multiswag_moments(
                xs, *st, seed_idx, scale=scale, philox_seed=philox_seed, system_id0=lo, draws_per_launch=self.draws_per_launch, plan=plan))
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(B_total, world)[rank]
        if x_local.shape[0] != hi - lo:
            raise ValueError(f"rank {rank} must hold systems [{lo}, {hi})")
        return all_gather_moments(self.local_moments(x_local, seed_idx, philox_seed, lo, scale), B_total, self.group)


    # ---- what the evaluation scripts consume, at any number of draws, in O(B_r * bins) memory ----------
    def local_bands(self, x_local, seed_idx, q, philox_seed, system_id0, trios=1, scale=0.5, stats=None, segments=None):
        """Fused forward + statistics epilogue in slabs of draws -> quantile sketch of this rank's simulations.
        Returns float32 [B_r / trios, len(q) + 1]: the percentiles q, then the mean (np.average) over the draws."""
        ops = self.ops
        wa, w2, pd = self.state
        sk = ops.QuantileSketch(x_local.shape[0], group=trios, segments=segments, device=x_local.device)
        st = stats or ops.stats_params(device=x_local.device)
        ops.multiswag_bands(x_local, wa, w2, pd, seed_idx, sk, st=st, scale=scale, philox_seed=philox_seed, system_id0=system_id0,
                            draws_per_launch=self.draws_per_launch, plan=self.plan)
        if sk.n_sims == 0:
            return torch.empty((0, len(q) + 1), dtype=torch.float32, device=x_local.device)
        return torch.cat([sk.percentiles(q), sk.mean().float()[:, None]], 1)

    def predictive_quantiles(self, x_local, B_total, seed_idx, q=(2.5, 16.0, 50.0, 84.0, 97.5), philox_seed=0, trios=1, scale=0.5,
                             stats=None, segments=None):
        """Per-simulation percentile bands + mean of the post-epilogue log10 instability time (truncated-normal draw, prior
        resampling, min over `trios` consecutive systems: figures/multiswag_5_planet.py:388-428, 484-489) for the dense
        (systems x draws) grid, sharded by simulation: [B_total / trios, len(q) + 1] on every rank after ONE all-gather.
        Neither [J,B,2] nor [J,B] is ever materialised beyond one slab of `draws_per_launch` draws."""
        if self.devset is not None:
            def per_shard(xs, st, plan, lo):
                sk = self.ops.QuantileSketch(xs.shape[0], group=trios, segments=segments, device=xs.device)
                self.ops.multiswag_bands(xs, *st, seed_idx, sk, st=stats or self.ops.stats_params(device=xs.device), scale=scale,
                                         philox_seed=philox_seed, system_id0=lo, draws_per_launch=self.draws_per_launch, plan=plan)
                return torch.cat([sk.percentiles(q), sk.mean().float()[:, None]], 1)
            return self._all_devices(x_local, B_total, trios, per_shard)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(B_total, world, trios)[rank]
        if x_local.shape[0] != hi - lo:
            raise ValueError(f"rank {rank} must hold systems [{lo}, {hi})")
        local = self.local_bands(x_local, seed_idx, q, philox_seed, lo, trios, scale, stats, segments)
        return all_gather_moments(local, B_total // trios, self.group)
