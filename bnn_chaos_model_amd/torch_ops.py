"""PyTorch custom-op registration of the hot-path entry points (torch.ops.bnn_chaos.*): swag_draw, forward, multiswag,
multiswag_moments, multiswag_stats.

BASELINE.json's north_star asks for "PyTorch-ROCm custom ops": these are thin torch.library wrappers over
bnn_chaos_model_amd.ops (ctypes -> C ABI -> HIP kernels) with shape-only fake implementations, so the ops can be
called from torch code, traced by torch.export / captured in HIP graphs by the caller, and show up by name in profiles.
Importing this module registers them; nothing else in the package depends on it.
"""
from typing import Optional

import torch

from . import ops

LIB = "bnn_chaos"


@torch.library.custom_op(f"{LIB}::swag_draw", mutates_args=())
def swag_draw(w_avg: torch.Tensor, w2_avg: torch.Tensor, pre_D: torch.Tensor, seed_idx: torch.Tensor,
              z1: Optional[torch.Tensor], z2: Optional[torch.Tensor], scale: float, philox_seed: int, draw_id0: int) -> torch.Tensor:
    """SWAGModel.sample_weights for J draws (spock_reg_model.py:815-838) -> W[J,d]."""
    return ops.swag_draw(w_avg, w2_avg, pre_D, seed_idx, z1, z2, scale=scale, philox_seed=philox_seed, draw_id0=draw_id0)


@swag_draw.register_fake
def _(w_avg, w2_avg, pre_D, seed_idx, z1, z2, scale, philox_seed, draw_id0):
    return w_avg.new_empty((seed_idx.numel(), w_avg.shape[1]))


@torch.library.custom_op(f"{LIB}::forward", mutates_args=())
def forward(x: torch.Tensor, W: torch.Tensor, eps: Optional[torch.Tensor], eps_in: Optional[torch.Tensor],
            eps_sum: Optional[torch.Tensor], nchunks: int, noisy: bool, philox_seed: int, draw_id0: int,
            system_id0: int) -> torch.Tensor:
    """VarModel.forward (spock_reg_model.py:486-528) for materialised weights -> [J/nchunks, B, 2]."""
    return ops.forward(x, W, eps, eps_in, eps_sum, nchunks=nchunks, noisy=noisy, philox_seed=philox_seed, draw_id0=draw_id0,
                       system_id0=system_id0)


@forward.register_fake
def _(x, W, eps, eps_in, eps_sum, nchunks, noisy, philox_seed, draw_id0, system_id0):
    return x.new_empty((W.shape[0] // nchunks, x.shape[0], 2))


@torch.library.custom_op(f"{LIB}::multiswag", mutates_args=())
def multiswag(x: torch.Tensor, w_avg: torch.Tensor, w2_avg: torch.Tensor, pre_D: torch.Tensor, seed_idx: torch.Tensor,
              z1: Optional[torch.Tensor], z2: Optional[torch.Tensor], eps: Optional[torch.Tensor], nchunks: int, scale: float,
              philox_seed: int, draw_id0: int, system_id0: int) -> torch.Tensor:
    """Fused forward_swag_fast over the MC loop (spock_reg_model.py:878-908, figures/multiswag_5_planet.py:295-298)."""
    return ops.multiswag(x, w_avg, w2_avg, pre_D, seed_idx, z1, z2, eps, nchunks=nchunks, scale=scale, philox_seed=philox_seed,
                         draw_id0=draw_id0, system_id0=system_id0)


@multiswag.register_fake
def _(x, w_avg, w2_avg, pre_D, seed_idx, z1, z2, eps, nchunks, scale, philox_seed, draw_id0, system_id0):
    return x.new_empty((seed_idx.numel() // nchunks, x.shape[0], 2))


@torch.library.custom_op(f"{LIB}::multiswag_moments", mutates_args=())
def multiswag_moments(x: torch.Tensor, w_avg: torch.Tensor, w2_avg: torch.Tensor, pre_D: torch.Tensor, seed_idx: torch.Tensor,
                      scale: float, philox_seed: int, draw_id0: int, system_id0: int, draws_per_launch: int) -> torch.Tensor:
    """Predictive moments of the dense (systems x draws) grid -> float64 [B, 4] (sum mu, sum mu^2, sum std, sum std^2), the draws
    evaluated in slabs of `draws_per_launch` so that [J,B,2] is never materialised (the multi-GPU gather payload):
    the native slab driver bnn_multiswag_moments_f64, one C-ABI call."""
    return ops.multiswag_moments(x, w_avg, w2_avg, pre_D, seed_idx, scale=scale, philox_seed=philox_seed, draw_id0=draw_id0,
                                 system_id0=system_id0, draws_per_launch=draws_per_launch)


@multiswag_moments.register_fake
def _(x, w_avg, w2_avg, pre_D, seed_idx, scale, philox_seed, draw_id0, system_id0, draws_per_launch):
    return x.new_empty((x.shape[0], 4), dtype=torch.float64)


@torch.library.custom_op(f"{LIB}::multiswag_stats", mutates_args=())
def multiswag_stats(x: torch.Tensor, w_avg: torch.Tensor, w2_avg: torch.Tensor, pre_D: torch.Tensor, seed_idx: torch.Tensor,
                    nchunks: int, scale: float, philox_seed: int, draw_id0: int, system_id0: int) -> torch.Tensor:
    """multiswag with the scripts' statistics epilogue (truncated-normal draw at 4, prior resampling at 9;
    figures/multiswag_5_planet.py:388-422) fused into the kernel tail -> t [J/nchunks, B]."""
    return ops.multiswag_stats(x, w_avg, w2_avg, pre_D, seed_idx, nchunks=nchunks, scale=scale, philox_seed=philox_seed,
                               draw_id0=draw_id0, system_id0=system_id0)


@multiswag_stats.register_fake
def _(x, w_avg, w2_avg, pre_D, seed_idx, nchunks, scale, philox_seed, draw_id0, system_id0):
    return x.new_empty((seed_idx.numel() // nchunks, x.shape[0]))
