import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def tape(z):
    """The reference's RNG draws, in consumption order, as (kind, array) pairs."""
    n = int(z["tape_n"])
    kinds = [str(k) for k in z["tape_kinds"]]
    return [(kinds[i], z[f"tape_{i:03d}"]) for i in range(n)]


def close_report(a, b, rtol=1e-5, atol=0.0):
    """|a-b| <= atol + rtol*|b| with the exceedance count; the default is BASELINE.json's bar: 1e-5 relative, no absolute slack."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    bad = err > (atol + rtol * np.abs(b))
    return int(bad.sum()), float(err.max()) if err.size else 0.0


@pytest.fixture(scope="session")
def swag_states():
    out = {}
    for i in (0, 12):
        z = load_golden(f"swag_v50_{i}.npz")
        out[i] = {k: z[k] for k in ("w_avg", "w2_avg", "pre_D", "ssX_mean", "ssX_scale")}
    return out


@pytest.fixture(scope="session")
def inputs():
    z = load_golden("inputs.npz")
    return {"slow": z["x_slow"], "iid": z["x_iid"], "const4": z["x_const4"]}
