import cProfile, io, json, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from bnn_chaos_model_amd import checkpoint
from bnn_chaos_model_amd.regression import FeatureRegressor
import bench
gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")
d = tempfile.mkdtemp()
for i in (0, 12):
    z = np.load(os.path.join(gold, f"swag_v50_{i}.npz"))
    checkpoint.write_swag_file(os.path.join(d, f"m_v50_{i:02d}_output.pkl"), json.loads(str(z["hparams_json"])), json.loads(str(z["swa_params_json"])),
                               torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
for cuda in (False, True):
    model = FeatureRegressor(cuda=cuda, filebase=os.path.join(d, "*v50*output.pkl"), sort=True)
    X = bench.synthetic_x(150, torch.device("cuda"), 1)
    if not cuda:
        X = X.cpu()
    parts = torch.chunk(X, 10)
    for _ in range(50):
        model.sample_full_swag(parts[0])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(50):
        for Xp in parts:
            model.sample_full_swag(Xp).detach().cpu()
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"cuda={cuda}: {dt / 500 * 1e3:.3f} ms per call")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
