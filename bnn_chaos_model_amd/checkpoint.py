"""Reader/writer for the reference's SWAG checkpoints (`*_output.pkl`).

Format (spock_reg_model.py:911-920, save_swag): a torch.save zip archive of
    {'hparams': AttributeDict, 'swa_params': dict, 'w_avg': f32[d], 'w2_avg': f32[d], 'pre_D': f32[d,K]}.
The pickle references pytorch_lightning.utilities.parsing.AttributeDict, which this image does not have
and which torch.load(weights_only=True) refuses; this module unpickles with a 4-global whitelist
("Checkpoint unpickling") and never executes anything else from the file.
"""
import collections
import io
import pickle
import zipfile

import numpy as np
import torch


class AttributeDict(dict):
    """dict addressable by attribute -- stand-in for pytorch_lightning.utilities.parsing.AttributeDict."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


_DTYPES = {"FloatStorage": np.float32, "DoubleStorage": np.float64, "LongStorage": np.int64, "IntStorage": np.int32,
           "HalfStorage": np.float16, "BoolStorage": np.bool_}


class _StorageType:
    def __init__(self, name):
        self.name = name
        self.dtype = _DTYPES[name]


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    arr = storage[storage_offset:]
    t = torch.from_numpy(arr)
    return torch.as_strided(t, tuple(size), tuple(stride)).clone()


class _Unpickler(pickle.Unpickler):
    ALLOWED = {
        ("pytorch_lightning.utilities.parsing", "AttributeDict"): AttributeDict,
        ("collections", "OrderedDict"): collections.OrderedDict,
        ("torch._utils", "_rebuild_tensor_v2"): _rebuild_tensor_v2,
    }

    def __init__(self, f, zf, prefix):
        super().__init__(f)
        self._zf, self._prefix, self._cache = zf, prefix, {}

    def find_class(self, module, name):
        if (module, name) in self.ALLOWED:
            return self.ALLOWED[(module, name)]
        if module == "torch" and name in _DTYPES:
            return _StorageType(name)
        raise pickle.UnpicklingError(f"checkpoint references {module}.{name}, which is not on the whitelist")

    def persistent_load(self, pid):
        # ('storage', storage_type, key, location, numel)
        if not (isinstance(pid, tuple) and pid and pid[0] == "storage"):
            raise pickle.UnpicklingError("unexpected persistent id")
        _, stype, key, _loc, numel = pid
        if key not in self._cache:
            raw = self._zf.read(f"{self._prefix}/data/{key}")
            self._cache[key] = np.frombuffer(raw, dtype=stype.dtype, count=int(numel)).copy()
        return self._cache[key]


def read_swag_file(path):
    """-> dict(hparams, swa_params, w_avg, w2_avg, pre_D) with CPU tensors."""
    with zipfile.ZipFile(path) as zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise ValueError(f"{path}: not a torch zip checkpoint")
        prefix = pkl[0][: -len("/data.pkl")]
        items = _Unpickler(io.BytesIO(zf.read(pkl[0])), zf, prefix).load()
    for k in ("hparams", "swa_params", "w_avg", "w2_avg", "pre_D"):
        if k not in items:
            raise ValueError(f"{path}: missing '{k}'")
    return items


def write_swag_file(path, hparams, swa_params, w_avg, w2_avg, pre_D):
    """save_swag (spock_reg_model.py:911-920): same keys; hparams stored as a plain dict (loadable by the reference)."""
    torch.save({"hparams": dict(hparams), "swa_params": dict(swa_params), "w_avg": w_avg.detach().cpu(),
                "w2_avg": w2_avg.detach().cpu(), "pre_D": pre_D.detach().cpu()}, path)
