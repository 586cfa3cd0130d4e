"""Wall time of the literal 5-planet MC loop (figures/multiswag_5_planet.py:295-298) through the drop-in surface, and of the
same loop as one launch.  150 systems (50 sims x 3 trios), 10 chunks, 100 samples = 1000 sample_full_swag calls."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bnn_chaos_model_amd import checkpoint
from bnn_chaos_model_amd.regression import FeatureRegressor
import bench

gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
d = tempfile.mkdtemp()
for i in (0, 12):
    z = np.load(os.path.join(gold, f"swag_v50_{i}.npz"))
    checkpoint.write_swag_file(os.path.join(d, f"m_v50_{i:02d}_output.pkl"), json.loads(str(z["hparams_json"])), json.loads(str(z["swa_params_json"])),
                               torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
for cuda in (False, True):
    model = FeatureRegressor(cuda=cuda, filebase=os.path.join(d, "*v50*output.pkl"), sort=True)
    Xflat = bench.synthetic_x(150, torch.device("cuda"), 1)
    if not cuda:
        Xflat = Xflat.cpu()
    samples = 100
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = torch.cat([torch.cat([model.sample_full_swag(Xp).detach().cpu() for Xp in torch.chunk(Xflat, chunks=10)])[None] for _ in range(samples)], dim=0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        one = model.sample_full_swag_many(Xflat, samples=samples, chunks=10, rng="torch")
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ph = model.sample_full_swag_many(Xflat, samples=samples, chunks=10, rng="philox")
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"cuda={cuda}: literal loop (1000 calls) {t1 - t0:.3f} s = {(t1 - t0):.3f} ms/call;  one launch, torch rng {t2 - t1:.3f} s;  one launch, philox {t3 - t2:.4f} s")
