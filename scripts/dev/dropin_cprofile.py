"""cProfile of the scripts' per-call route (model and x on the GPU, 15-row chunks): where the Python time of a call goes (the absolute
numbers carry the profiler's own overhead; the ranking is what matters).  python scripts/dev/dropin_cprofile.py"""
import cProfile
import io
import json
import os
import pstats
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from bnn_chaos_model_amd import checkpoint  # noqa: E402
from bnn_chaos_model_amd.regression import FeatureRegressor  # noqa: E402
import bench  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
d = tempfile.mkdtemp()
for i in (0, 12):
    z = np.load(os.path.join(gold, f"swag_v50_{i}.npz"))
    checkpoint.write_swag_file(os.path.join(d, f"m_v50_{i:02d}_output.pkl"), json.loads(str(z["hparams_json"])), json.loads(str(z["swa_params_json"])),
                               torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
model = FeatureRegressor(cuda=True, filebase=os.path.join(d, "*v50*output.pkl"), sort=True)
X = bench.synthetic_x(15, torch.device("cuda"), 1)
for _ in range(200):
    model.sample_full_swag(X)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    model.sample_full_swag(X)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:170] for l in s.getvalue().splitlines()))
