"""All the GPUs of a node from ONE process: what lets the reference's single-process evaluation scripts
(figures/multiswag_5_planet.py:61, 280-298; figures/main_figures.py:39-42, 148-156) use eight MI355X unchanged.

Same decomposition as distributed.py: shard the systems (whole simulations when trios are grouped),
replicate the ensemble and the draw list per device, evaluate every shard on its own device -- the launches are asynchronous, so a
Python loop over the devices keeps all of them busy -- and assemble the per-system results with ONE exchange.  The in-kernel noise
is keyed by global (draw, system) ids and the draws' chunks are cut over the whole batch (bnn_grid.chunk_B / chunk_off), so the
result is bit-identical for any device list, including the same device named several times (how the one-GPU test box checks it).

The exchange: `torch.cuda.nccl.all_gather` (RCCL over xGMI, the single-process form: one communicator per device inside this
process) when the devices are distinct and RCCL accepts them; otherwise peer copies onto the first device.  `last_exchange` says
which ran; BNN_MULTIDEVICE_EXCHANGE=copies forces the copies (the payload is a few floats per simulation either way).
"""
import os

import torch

from .distributed import shard_bounds


def resolve_devices(devices=None):
    """None -> every visible GPU; an int n -> the first n; else an explicit list of device indices / torch.devices (repeats allowed:
    several logical shards on one card)."""
    if devices is None:
        devices = list(range(torch.cuda.device_count()))
    elif isinstance(devices, int):
        devices = list(range(devices))
    out = [torch.device("cuda", d) if isinstance(d, int) else torch.device(d) for d in devices]
    if not out:
        raise RuntimeError("bnn_chaos_model_amd needs an MI355X (gfx950) GPU: there is no CPU implementation")
    for d in out:
        if d.type != "cuda" or d.index is None or d.index >= torch.cuda.device_count():
            raise ValueError(f"not a visible GPU: {d}")
    return out


class DeviceSet:
    """A list of devices + per-device replicas of read-only tensors (the ensemble: 29 MB for the 30 pretrained members)."""

    def __init__(self, devices=None):
        self.devices = resolve_devices(devices)
        self._replicas = {}
        self.last_exchange = None

    def __len__(self):
        return len(self.devices)

    def replicate(self, key, tensors):
        """tensors: tuple of tensors (any device) -> list over devices of tuples on that device; cached under `key` while the source
        tensors stay the same objects."""
        ident = tuple(id(t) for t in tensors)
        hit = self._replicas.get(key)
        if hit is None or hit[0] != ident:
            per_index = {}
            for d in self.devices:
                if d.index not in per_index:
                    per_index[d.index] = tuple(t.detach().to(d, torch.float32 if t.is_floating_point() else t.dtype).contiguous() for t in tensors)
            hit = (ident, [per_index[d.index] for d in self.devices], tensors)   # keep the sources alive: ids stay unique
            self._replicas[key] = hit
        return hit[1]

    def bounds(self, B, group=1):
        return shard_bounds(B, len(self.devices), group)

    def run(self, B, fn, group=1):
        """fn(i, device, lo, hi) -> tensor [hi - lo (/ group), M] on `device`, for every non-empty shard; the launches of one shard
        are enqueued before the next shard's, nothing waits.  Returns the list of per-shard results (None for empty shards)."""
        parts = []
        for i, (d, (lo, hi)) in enumerate(zip(self.devices, self.bounds(B, group))):
            if hi == lo:
                parts.append(None)
                continue
            with (torch.cuda.device(d) if d.type == "cuda" else _nullctx()):
                parts.append(fn(i, d, lo, hi))
        return parts

    def gather_rows(self, parts):
        """The path's ONE exchange: per-shard [n_i, M] tensors (rows in shard order) -> [sum n_i, M] on the first device."""
        parts = [p for p in parts if p is not None and p.shape[0] > 0]
        if not parts:
            raise ValueError("nothing to gather")
        if len(parts) == 1:
            self.last_exchange = "none (one shard)"
            return parts[0]
        devs = [p.device for p in parts]
        if os.environ.get("BNN_MULTIDEVICE_EXCHANGE", "rccl") == "copies":
            self.last_exchange = "peer copies (BNN_MULTIDEVICE_EXCHANGE=copies)"
        elif len(set(devs)) == len(devs):
            try:
                res = _rccl_all_gather(parts)
                self.last_exchange = "rccl all_gather (torch.cuda.nccl, one process)"
                return res
            except Exception as e:   # RCCL unavailable in this process: the copies below are always valid
                self.last_exchange = f"peer copies (torch.cuda.nccl.all_gather failed: {type(e).__name__}: {e})"
        else:
            self.last_exchange = "peer copies (a device is named more than once)"
        d0 = parts[0].device
        return torch.cat([p.to(d0, non_blocking=True) for p in parts], 0)


class _nullctx:   # (host-side tests drive the partition / exchange logic with CPU tensors)
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _rccl_all_gather(parts):
    """Single-process RCCL all-gather of unequal row counts: shards padded to the longest, gathered on every device, the first
    device's copy trimmed.  Stream-ordered on every device's current stream."""
    import torch.cuda.nccl as nccl
    nmax = max(p.shape[0] for p in parts)
    M, dt = parts[0].shape[1], parts[0].dtype
    inputs, outputs = [], []
    for p in parts:
        with torch.cuda.device(p.device):
            if p.shape[0] == nmax:
                inp = p.contiguous()
            else:
                inp = torch.zeros((nmax, M), dtype=dt, device=p.device)
                inp[: p.shape[0]] = p
            inputs.append(inp)
            outputs.append(torch.empty((len(parts) * nmax, M), dtype=dt, device=p.device))
    if not nccl.is_available(inputs):
        raise RuntimeError("torch.cuda.nccl is not available for these tensors")
    nccl.all_gather(inputs, outputs)
    out = outputs[0]
    if all(p.shape[0] == nmax for p in parts):
        return out
    return torch.cat([out[i * nmax: i * nmax + p.shape[0]] for i, p in enumerate(parts)], 0)
