"""Several GPUs of a node from ONE process: what lets the reference's single-process evaluation scripts
(figures/multiswag_5_planet.py:61, 280-298; figures/main_figures.py:39-42, 148-156) use eight MI355X with one argument.

Same decomposition as distributed.py: shard the systems (whole simulations when trios are grouped),
replicate the ensemble and the draw list per device, evaluate every shard on its own device -- the launches are asynchronous, so a
Python loop over the devices keeps all of them busy -- and assemble the per-system results with ONE exchange.  The in-kernel noise
is keyed by global (draw, system) ids and the draws' chunks are cut over the whole batch (bnn_grid.chunk_B / chunk_off), so the
result is bit-identical for any device list, including the same device named several times (how the one-GPU test box checks it).

Which devices: `devices=None` is the CURRENT device -- a script that called torch.cuda.set_device(k), or a torch.distributed rank that
sees every GPU of the node, keeps its one GPU.  `devices="all"` (or the environment variable BNN_CHAOS_DEVICES=all, for scripts that
cannot be edited), an int n (the first n) or an explicit list opt in to more; under a process-per-GPU launcher (WORLD_SIZE > 1 or
LOCAL_RANK set) "all" still means the current device: the launcher already gave every GPU its own process.

Host-resident inputs: every device's rows are on their way BEFORE the first kernel is launched, and the eight PCIe links run
concurrently -- asynchronous copies when the source is pinned (or already on a GPU), one host thread per device when it is pageable
memory (a copy from pageable memory blocks the thread that issues it: issued from one thread the links would take turns).
`h2d_ms()` reports the slowest device's copy time of the last staging.

The exchange: peer copies onto the first device by default (a few floats per simulation).  `torch.cuda.nccl.all_gather` (RCCL over
xGMI, the single-process form: one communicator per device inside this process) is built and exercised at world size 1
(tests/test_multidevice.py) but has never run between distinct GPUs from this environment, so it is opt-in:
BNN_MULTIDEVICE_EXCHANGE=rccl or DeviceSet(exchange="rccl").  `last_exchange` says which ran.
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import torch

from .distributed import shard_bounds


def under_launcher():
    """True inside a process-per-GPU launch (torch.distributed.run / torchrun / mpirun wrappers set these)."""
    try:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            return True
    except ValueError:
        pass
    return "LOCAL_RANK" in os.environ


def resolve_devices(devices=None):
    """None -> the current device (BNN_CHAOS_DEVICES in the environment stands in for the argument when it is None); "all" -> every
    visible GPU, except under a process-per-GPU launcher (the current device); an int n -> the first n; else an explicit list of device
    indices / torch.devices (repeats allowed: several logical shards on one card)."""
    if devices is None:
        env = os.environ.get("BNN_CHAOS_DEVICES", "").strip()
        if env:
            devices = "all" if env.lower() == "all" else [int(v) for v in env.split(",") if v.strip() != ""]
    n_vis = torch.cuda.device_count()
    if n_vis == 0:
        raise RuntimeError("bnn_chaos_model_amd needs an MI355X (gfx950) GPU: there is no CPU implementation")
    if devices is None or (isinstance(devices, str) and devices == "all" and under_launcher()):
        devices = [torch.cuda.current_device()]
    elif isinstance(devices, str):
        if devices != "all":
            raise ValueError("devices must be None, 'all', an int or a list of devices")
        devices = list(range(n_vis))
    elif isinstance(devices, int):
        devices = list(range(devices))
    out = [torch.device("cuda", d) if isinstance(d, int) else torch.device(d) for d in devices]
    if not out:
        raise ValueError("empty device list")
    for d in out:
        if d.type != "cuda" or d.index is None or d.index >= n_vis:
            raise ValueError(f"not a visible GPU: {d}")
    return out


class RcclUnavailable(RuntimeError):
    """torch.cuda.nccl cannot serve these tensors in this process (not compiled in / refuses the device list)."""


class DeviceSet:
    """A list of devices + per-device replicas of read-only tensors (the ensemble: 29 MB for the 30 pretrained members)."""

    def __init__(self, devices=None, exchange=None):
        self.devices = resolve_devices(devices)
        self._replicas = {}
        self.last_exchange = None
        self.exchange = exchange or os.environ.get("BNN_MULTIDEVICE_EXCHANGE", "copies")
        if self.exchange not in ("copies", "rccl"):
            raise ValueError("exchange must be 'copies' or 'rccl'")
        self._h2d = None

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

    # ---- inputs: every shard on its way before anything is launched ---------------------------------------------------------------
    def stage(self, X, group=1, dtype=torch.float32):
        """Rows of X [B, ...] (host or device memory) -> list over devices of this device's rows ON the device (None for an empty
        shard).  A source that is pinned or already on a GPU is copied asynchronously (`non_blocking`), every device's copy issued
        before this returns; pageable host memory is copied by one host thread per device, concurrently.  See h2d_ms().

        Contract for a PINNED host source: the copies may still be in flight when this returns -- the source must stay untouched until
        release_sources() has returned (the batched drivers call it before they hand control back, so their callers may refill a pinned
        staging buffer as soon as the driver returns).  Every copy -- and the shard's allocation -- belongs to the stream that is
        current for its device in the CALLING thread (the worker threads of the pageable path enter it: torch's current stream is
        thread-local), so kernels the caller then enqueues on that stream are ordered behind the copy and own the same allocator block."""
        bounds = self.bounds(X.shape[0], group)
        X = X.detach()
        on_host = not X.is_cuda
        use_async = X.is_cuda or X.is_pinned()
        shards = [None] * len(self.devices)
        marks = [None] * len(self.devices)
        caller_streams = [torch.cuda.current_stream(d) if d.type == "cuda" else None for d in self.devices]
        t0 = time.perf_counter()

        def copy_one(i):
            d, (lo, hi) = self.devices[i], bounds[i]
            if hi == lo:
                return
            if d.type != "cuda":   # (host-side tests drive the partition logic with CPU "devices")
                shards[i] = X[lo:hi].to(dtype).contiguous()
                return
            with torch.cuda.device(d), torch.cuda.stream(caller_streams[i]):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                shards[i] = X[lo:hi].to(d, dtype, non_blocking=use_async).contiguous()
                ev1.record()
                marks[i] = (ev0, ev1)

        live = [i for i, (lo, hi) in enumerate(bounds) if hi > lo]
        if use_async or len(live) <= 1:
            for i in live:
                copy_one(i)
        else:
            with ThreadPoolExecutor(len(live)) as ex:   # the copy releases the GIL: the devices' links run side by side
                list(ex.map(copy_one, live))
        self._h2d = {"marks": marks, "host_ms": (time.perf_counter() - t0) * 1e3, "bytes": int(X[0:0].element_size() * X.numel()) if on_host else 0,
                     "mode": ("device-to-device" if X.is_cuda else "pinned, asynchronous" if use_async else f"pageable, {max(len(live), 1)} host thread(s)"),
                     "pending_host_source": bool(on_host and use_async)}
        return shards

    def release_sources(self):
        """Returns once the last stage()'s asynchronous copies out of PINNED host memory have completed (their end events; kernels enqueued
        behind them are not waited for): from then on the caller may overwrite the source.  No-op for every other kind of source."""
        if self._h2d and self._h2d.get("pending_host_source"):
            for m in self._h2d["marks"]:
                if m is not None:
                    m[1].synchronize()
            self._h2d["pending_host_source"] = False

    def h2d_ms(self):
        """Copy time of the last stage(): {"ms": slowest device (HIP events around its copy), "host_ms": wall time of the staging call,
        "bytes": host bytes moved, "mode": how}.  Waits for the copies' events (long done once results exist)."""
        if not self._h2d:
            return None
        ms = 0.0
        for m in self._h2d["marks"]:
            if m is not None:
                m[1].synchronize()
                ms = max(ms, m[0].elapsed_time(m[1]))
        return {"ms": ms, "host_ms": self._h2d["host_ms"], "bytes": self._h2d["bytes"], "mode": self._h2d["mode"]}

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
        if self.exchange == "rccl" and len(set(devs)) == len(devs) and parts[0].is_cuda:
            try:
                res = _rccl_all_gather(parts)
                self.last_exchange = "rccl all_gather (torch.cuda.nccl, one process)"
                return res
            except RcclUnavailable as e:   # RCCL absent in this process: the copies below are always valid.  Anything else is a bug: it raises.
                self.last_exchange = f"peer copies (rccl asked for but unavailable: {e})"
        elif self.exchange == "rccl":
            self.last_exchange = "peer copies (rccl asked for, but a device is named more than once)"
        else:
            self.last_exchange = "peer copies"
        d0 = parts[0].device
        return torch.cat([p.to(d0, non_blocking=True) for p in parts], 0)


class _nullctx:   # (host-side tests drive the partition / exchange logic with CPU tensors)
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _rccl_all_gather(parts):
    """Single-process RCCL all-gather of unequal row counts: shards padded to the longest, gathered on every device, the first
    device's copy trimmed.  Stream-ordered on every device's current stream.  Raises RcclUnavailable when torch.cuda.nccl is not
    there for these tensors; every other failure propagates."""
    try:
        import torch.cuda.nccl as nccl
    except ImportError as e:
        raise RcclUnavailable(f"torch.cuda.nccl cannot be imported: {e}")
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
        raise RcclUnavailable("torch.cuda.nccl.is_available() is False for these tensors")
    nccl.all_gather(inputs, outputs)
    out = outputs[0]
    if all(p.shape[0] == nmax for p in parts):
        return out
    return torch.cat([out[i * nmax: i * nmax + p.shape[0]] for i, p in enumerate(parts)], 0)
