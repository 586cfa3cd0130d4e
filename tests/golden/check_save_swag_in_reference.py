#!/usr/bin/env python3
"""save_swag compatibility seen from the REFERENCE's side (build container only).

    python tests/golden/check_save_swag_in_reference.py <file written by bnn_chaos_model_amd.spock_reg_model.save_swag> [member]

Loads the file with the UNMODIFIED reference's load_swag (spock_reg_model.py:922-967; make_golden.import_reference supplies the two
absent modules), compares what it finds with the converted state of pretrained member `member` (default 12) in tests/golden --
w_avg / w2_avg / pre_D bit for bit, hparams, swa_params, K, c, the 'v50' scaler -- and runs ONE reference forward_swag_fast on it under
the seed of case_swagfast_v50_<member>_slow.npz: the reference, reading OUR file, must reproduce its own fixture bit for bit (the HIP
surface is held to that same fixture by tests/test_surface_gpu.py).  Prints one JSON line; exit code 0 = compatible.
tests/test_host_cpu.py runs this in a subprocess when /root/reference is present.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    path = sys.argv[1]
    member = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    srm = import_reference()
    torch.set_num_threads(1)      # the fixtures were generated single-threaded (fixed summation order inside MKL)
    m = srm.load_swag(path).cpu()
    m.eval()
    z = np.load(os.path.join(HERE, f"swag_v50_{member}.npz"))
    checks = {
        "class": type(m).__name__ == "SWAGModel" and type(m).__module__ == "spock_reg_model",
        "w_avg": np.array_equal(m.w_avg.numpy(), z["w_avg"]), "w2_avg": np.array_equal(m.w2_avg.numpy(), z["w2_avg"]),
        "pre_D": np.array_equal(m.pre_D.numpy(), z["pre_D"]) and m.pre_D.is_contiguous(),
        "dtypes": all(t.dtype == torch.float32 for t in (m.w_avg, m.w2_avg, m.pre_D)),
    }
    hp_want = json.loads(str(z["hparams_json"]))
    hp_got = {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in dict(m.hparams).items()}
    checks["hparams"] = hp_got == hp_want
    swa_want = json.loads(str(z["swa_params_json"]))
    checks["swa_params"] = dict(m.swa_params) == swa_want
    checks["K_c"] = (m.K, m.c) == (swa_want["K"], swa_want["c"])
    checks["ssX"] = ("v50" not in path) or (np.array_equal(m.ssX.mean_, z["ssX_mean"]) and np.array_equal(m.ssX.scale_, z["ssX_scale"]))
    c = np.load(os.path.join(HERE, f"case_swagfast_v50_{member}_slow.npz"))
    x = torch.tensor(np.load(os.path.join(HERE, "inputs.npz"))["x_slow"])
    torch.manual_seed(int(c["torch_seed"]))
    out = m.forward_swag_fast(x, scale=0.5).detach().numpy()
    checks["forward_swag_fast_bit_identical_to_fixture"] = np.array_equal(out, c["out"])
    checks["sampled_weights_bit_identical_to_fixture"] = np.array_equal(m.flatten().detach().numpy(), c["w"])
    ok = all(checks.values())
    print(json.dumps({"ok": ok, "checks": checks, "torch": torch.__version__}))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
