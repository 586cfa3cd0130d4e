"""PyTorch custom-op registration of the hot-path entry points (torch.ops.bnn_chaos.*): swag_draw, forward, multiswag,
multiswag_moments, multiswag_stats.

BASELINE.json's north_star asks for "PyTorch-ROCm custom ops": these are torch.library operators over
bnn_chaos_model_amd.ops (ctypes -> C ABI -> HIP kernels) with shape-only fake implementations, so the ops can be
called from torch code, traced by torch.export / captured in HIP graphs by the caller, and show up by name in profiles.
They are registered with the low-level torch.library.Library API (define + impl for the CUDA key + register_fake), not with the
torch.library.custom_op decorator: the decorator's autograd and mutation-check wrappers cost 19 us per call against 6 us for a plain
dispatcher entry (measured), and the scripts' route makes one such call per 15-row chunk (figures/multiswag_5_planet.py:295-298).
None of the ops is differentiable (the reference's callers .detach() every result).
The module surface routes through them: VarModel.forward, SWAGModel.sample_weights and SWAGModel.forward_swag_fast
(spock_reg_model.py) call torch.ops.bnn_chaos.forward / swag_draw / multiswag.  The network (`net` = [n_features, hidden, latent,
depth_in, depth_out, fix_megno]; None = the pretrained ensemble's), its column mask and its clamp floor select the plan -- for all five
ops: a checkpoint built with other hparams (spock_reg_model.py:343-362) goes through every one of them.
assume_finite (the four ops that read x): False = scan x once and give systems that hold NaN / +-inf the reference's result
(ops.nonfinite_scan); True = x is known to be clean.
"""
import torch

from . import ops

LIB = "bnn_chaos"
_lib = torch.library.Library(LIB, "DEF")
_NET = f"int zero_mask={ops.V50_ZERO_MASK}, float lowest_std=0.5, int[]? net=None"


def _plan(zero_mask, lowest_std, net):
    if net is None:
        return ops.get_plan(zero_mask, lowest_std)
    return ops.get_plan(zero_mask, lowest_std, fix_megno=bool(net[5]), n_features=net[0], hidden=net[1], latent=net[2], depth_in=net[3],
                        depth_out=net[4])


def _register(name, schema, impl, fake):
    _lib.define(f"{name}({schema}) -> Tensor")
    _lib.impl(name, impl, "CUDA")
    torch.library.register_fake(f"{LIB}::{name}", fake, lib=_lib)
    return getattr(getattr(torch.ops, LIB), name)


def _swag_draw(w_avg, w2_avg, pre_D, seed_idx, z1, z2, scale, philox_seed, draw_id0, zero_mask=ops.V50_ZERO_MASK, lowest_std=0.5, net=None):
    """SWAGModel.sample_weights for J draws (spock_reg_model.py:815-838) -> W[J,d]."""
    return ops.swag_draw(w_avg, w2_avg, pre_D, seed_idx, z1, z2, scale=scale, philox_seed=philox_seed, draw_id0=draw_id0,
                         plan=_plan(zero_mask, lowest_std, net))


def _swag_draw_fake(w_avg, w2_avg, pre_D, seed_idx, z1, z2, scale, philox_seed, draw_id0, zero_mask=ops.V50_ZERO_MASK, lowest_std=0.5, net=None):
    return w_avg.new_empty((seed_idx.numel(), w_avg.shape[1]))


swag_draw = _register("swag_draw", "Tensor w_avg, Tensor w2_avg, Tensor pre_D, Tensor seed_idx, Tensor? z1, Tensor? z2, float scale, int philox_seed, "
                      f"int draw_id0, {_NET}", _swag_draw, _swag_draw_fake)


def _forward(x, W, eps, eps_in, eps_sum, nchunks, noisy, philox_seed, draw_id0, system_id0, zero_mask=ops.V50_ZERO_MASK, lowest_std=0.5, net=None,
             assume_finite=False):
    """VarModel.forward (spock_reg_model.py:486-528) for materialised weights -> [J/nchunks, B, 2]."""
    return ops.forward(x, W, eps, eps_in, eps_sum, nchunks=nchunks, noisy=noisy, philox_seed=philox_seed, draw_id0=draw_id0,
                       system_id0=system_id0, plan=_plan(zero_mask, lowest_std, net), assume_finite=assume_finite)


def _forward_fake(x, W, eps, eps_in, eps_sum, nchunks, noisy, philox_seed, draw_id0, system_id0, zero_mask=ops.V50_ZERO_MASK, lowest_std=0.5, net=None,
                  assume_finite=False):
    return x.new_empty((W.shape[0] // nchunks, x.shape[0], 2))


forward = _register("forward", "Tensor x, Tensor W, Tensor? eps, Tensor? eps_in, Tensor? eps_sum, int nchunks, bool noisy, int philox_seed, int draw_id0, "
                    f"int system_id0, {_NET}, bool assume_finite=False", _forward, _forward_fake)


def _multiswag(x, w_avg, w2_avg, pre_D, seed_idx, z1, z2, eps, nchunks, scale, philox_seed, draw_id0, system_id0, zero_mask=ops.V50_ZERO_MASK,
               lowest_std=0.5, net=None, assume_finite=False):
    """Fused forward_swag_fast over the MC loop (spock_reg_model.py:878-908, figures/multiswag_5_planet.py:295-298)."""
    return ops.multiswag(x, w_avg, w2_avg, pre_D, seed_idx, z1, z2, eps, nchunks=nchunks, scale=scale, philox_seed=philox_seed,
                         draw_id0=draw_id0, system_id0=system_id0, plan=_plan(zero_mask, lowest_std, net), assume_finite=assume_finite)


def _multiswag_fake(x, w_avg, w2_avg, pre_D, seed_idx, z1, z2, eps, nchunks, scale, philox_seed, draw_id0, system_id0, zero_mask=ops.V50_ZERO_MASK,
                    lowest_std=0.5, net=None, assume_finite=False):
    return x.new_empty((seed_idx.numel() // nchunks, x.shape[0], 2))


multiswag = _register("multiswag", "Tensor x, Tensor w_avg, Tensor w2_avg, Tensor pre_D, Tensor seed_idx, Tensor? z1, Tensor? z2, Tensor? eps, int nchunks, "
                      f"float scale, int philox_seed, int draw_id0, int system_id0, {_NET}, bool assume_finite=False", _multiswag, _multiswag_fake)


def _multiswag_moments(x, w_avg, w2_avg, pre_D, seed_idx, scale, philox_seed, draw_id0, system_id0, draws_per_launch, zero_mask=ops.V50_ZERO_MASK,
                       lowest_std=0.5, net=None, assume_finite=False):
    """Predictive moments of the dense (systems x draws) grid -> float64 [B, 4] (sum mu, sum mu^2, sum std, sum std^2), the draws
    evaluated in slabs of `draws_per_launch` so that [J,B,2] is never materialised (the multi-GPU gather payload):
    the native slab driver bnn_multiswag_moments_f64, one C-ABI call."""
    return ops.multiswag_moments(x, w_avg, w2_avg, pre_D, seed_idx, scale=scale, philox_seed=philox_seed, draw_id0=draw_id0,
                                 system_id0=system_id0, draws_per_launch=draws_per_launch, plan=_plan(zero_mask, lowest_std, net),
                                 assume_finite=assume_finite)


def _multiswag_moments_fake(x, w_avg, w2_avg, pre_D, seed_idx, scale, philox_seed, draw_id0, system_id0, draws_per_launch,
                            zero_mask=ops.V50_ZERO_MASK, lowest_std=0.5, net=None, assume_finite=False):
    return x.new_empty((x.shape[0], 4), dtype=torch.float64)


multiswag_moments = _register("multiswag_moments", "Tensor x, Tensor w_avg, Tensor w2_avg, Tensor pre_D, Tensor seed_idx, float scale, int philox_seed, "
                              f"int draw_id0, int system_id0, int draws_per_launch, {_NET}, bool assume_finite=False",
                              _multiswag_moments, _multiswag_moments_fake)


def _multiswag_stats(x, w_avg, w2_avg, pre_D, seed_idx, nchunks, scale, philox_seed, draw_id0, system_id0, zero_mask=ops.V50_ZERO_MASK,
                     lowest_std=0.5, net=None, assume_finite=False):
    """multiswag with the scripts' statistics epilogue (truncated-normal draw at 4, prior resampling at 9;
    figures/multiswag_5_planet.py:388-422) fused into the kernel tail -> t [J/nchunks, B]."""
    return ops.multiswag_stats(x, w_avg, w2_avg, pre_D, seed_idx, nchunks=nchunks, scale=scale, philox_seed=philox_seed,
                               draw_id0=draw_id0, system_id0=system_id0, plan=_plan(zero_mask, lowest_std, net), assume_finite=assume_finite)


def _multiswag_stats_fake(x, w_avg, w2_avg, pre_D, seed_idx, nchunks, scale, philox_seed, draw_id0, system_id0, zero_mask=ops.V50_ZERO_MASK,
                          lowest_std=0.5, net=None, assume_finite=False):
    return x.new_empty((seed_idx.numel() // nchunks, x.shape[0]))


multiswag_stats = _register("multiswag_stats", "Tensor x, Tensor w_avg, Tensor w2_avg, Tensor pre_D, Tensor seed_idx, int nchunks, float scale, int philox_seed, "
                            f"int draw_id0, int system_id0, {_NET}, bool assume_finite=False", _multiswag_stats, _multiswag_stats_fake)
