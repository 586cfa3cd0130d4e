"""Roofline evidence for the small kernels either side of the path (rounds 4-5), at one GPU's share of BASELINE configs[4]
(125 000 five-planet simulations = 375 000 rows x 100 timesteps; 100 samples per slab):
  bnn_feature_pack_kernel   data_setup_kernel + StandardScaler + .float()   (figures/spock/regression.py:183-213, multiswag_5_planet.py:280-287)
  bnn_sketch_update_kernel  min over trios + histogram update               (multiswag_5_planet.py:428, 484-489)
  bnn_stats_draw_kernel     truncated-normal draw + prior resampling        (:388-422)
  bnn_moments_kernel        [R,B,2] -> float64 [B,4]
  bnn_quantiles_kernel      exact per-system order statistics (one workgroup sorts one column)
  bnn_swag_draw_kernel      SWAGModel.sample_weights for 1 000 draws
  bnn_nonfinite_scan_kernel one streaming read of x for NaN / +-inf (round 5; spock_reg_model.py:452-478 is why it exists)
  bnn_nonfinite_fixup_kernel the exact re-evaluation of the listed systems: an empty list, 0.1 % certain, 0.1 % needing the evaluation
Prints one JSON line per kernel: algorithmic bytes, time (HIP events, median of 5), GB/s, fraction of the 6.3 TB/s a streaming copy
achieves on this chip (MI355X_MICROARCH.md).  Run it under rocprofv3 --kernel-trace --stats for the per-kernel averages."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bnn_chaos_model_amd import ops  # noqa: E402

STREAM_GBS = 6300.0


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def report(name, nbytes, ms, note=""):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(json.dumps({"kernel": name, "algorithmic_bytes": int(nbytes), "ms": round(ms, 4), "GBs": round(gbs, 1),
                      "frac_of_stream": round(gbs / STREAM_GBS, 3), "note": note}), flush=True)


def main():
    N, T, R = int(os.environ.get("SK_ROWS", 375_000)), 100, 100
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    # ---- feature packing: tseries [N,T,26] + mass [N,3] float64 -> x [N,T,41] float32 (standardised)
    from bnn_chaos_model_amd.spock_reg_model import v50_scaler
    ssX = v50_scaler()
    ts = torch.randn(N, T, 26, generator=g, device=dev, dtype=torch.float64)
    mass = torch.rand(N, 3, generator=g, device=dev, dtype=torch.float64) * 1e-4
    ms = timed(lambda: ops.feature_pack(ts, mass, mean=ssX.mean_, scale=ssX.scale_))
    report("bnn_feature_pack_kernel (-> x32)", N * T * (26 * 8 + 41 * 4) + N * 24, ms, f"{N} x {T} rows: 208 B read + 164 B written per row")
    ms = timed(lambda: ops.feature_pack(ts, mass, mean=ssX.mean_, scale=ssX.scale_, want_x64=True))
    report("bnn_feature_pack_kernel (-> x32 + X64)", N * T * (26 * 8 + 41 * 4 + 41 * 8) + N * 24, ms, "also data_setup_kernel's float64 return value")
    del ts
    # ---- statistics epilogue on materialised pairs, sketch update, moments, exact quantiles
    musd = torch.empty(R, N, 2, device=dev)
    musd[..., 0] = 4 + 8 * torch.rand(R, N, generator=g, device=dev)
    musd[..., 1] = 0.5 + 2 * torch.rand(R, N, generator=g, device=dev)
    st = ops.stats_params(device=dev)
    tt = ops.stats_draw(musd, st=st, philox_seed=1)
    ms = timed(lambda: ops.stats_draw(musd, st=st, philox_seed=1))
    report("bnn_stats_draw_kernel", R * N * 12, ms, "8 B read + 4 B written per evaluation; 1 Philox block (+ a second and a 13-step bisection past the threshold)")
    sk = ops.QuantileSketch(N, group=3, device=dev)
    ms = timed(lambda: sk.update(tt))
    report("bnn_sketch_update_kernel", R * N * 4 + R * (N // 3) * 8, ms, "4 B read per evaluation + one 4-byte atomic (read-modify-write) per simulation and draw")
    ms = timed(lambda: sk.percentiles((2.5, 16.0, 50.0, 84.0, 97.5)))
    report("bnn_sketch_quantiles_kernel", sk.nbins * (N // 3) * 4, ms, "reads the whole histogram once")
    ms = timed(lambda: ops.moments(musd))
    report("bnn_moments_kernel", R * N * 8 + N * 32, ms)
    nq = min(N, 100_000)
    ms = timed(lambda: ops.quantiles(musd[:, :nq].contiguous(), (50.0,)))
    report("bnn_quantiles_kernel", R * nq * 8 + nq * 8, ms, f"{nq} systems x 2 channels, bitonic sort of {R} values per workgroup in LDS")
    # ---- SWAG draw: 1 000 draws of the 30-member ensemble
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ensemble_v50.npz"))
    wa, w2, pd = (torch.as_tensor(z[k]).to(dev) for k in ("w_avg", "w2_avg", "pre_D"))
    idx = (torch.arange(1000, dtype=torch.int32) % 30).to(dev)
    ms = timed(lambda: ops.swag_draw(wa, w2, pd, idx, philox_seed=3))
    report("bnn_swag_draw_kernel", 1000 * 7583 * 4 * (32 + 1), ms, "970 KB of SWAG state read (L2 / Infinity Cache resident: 29 MB for 30 members) + 30 KB written per draw")
    # ---- non-finite inputs: the scan (one pass over x), and a whole call with an empty list / 0.1 % listed systems
    del musd, tt
    x = torch.randn(N, T, 41, generator=g, device=dev)
    ms = timed(lambda: ops.nonfinite_scan(x))
    report("bnn_nonfinite_scan_kernel", N * T * 41 * 4, ms, f"{N} systems x {T} x 41 floats read once; the record [4 + B] int32 written only for listed systems")
    idx100 = (torch.arange(100, dtype=torch.int32) % 30).to(dev)
    out = torch.empty(100, N, 2, device=dev)
    base = timed(lambda: ops.multiswag(x, wa, w2, pd, idx100, philox_seed=3, out=out, assume_finite=True), reps=3)
    withscan = timed(lambda: ops.multiswag(x, wa, w2, pd, idx100, philox_seed=3, out=out), reps=3)
    print(json.dumps({"kernel": "whole call, 100 draws: assume_finite vs scan + (empty) fix-up", "ms_assume_finite": round(base, 3), "ms_with_scan": round(withscan, 3),
                      "overhead_frac": round(withscan / base - 1.0, 5)}), flush=True)
    hurt = torch.randperm(N, generator=torch.Generator().manual_seed(5))[: max(1, N // 1000)].to(dev)
    xc = x.clone(); xc[hurt, 7, 3] = float("nan")        # a masked column: certain NaN, answered without the evaluation
    t_c = timed(lambda: ops.multiswag(xc, wa, w2, pd, idx100, philox_seed=3, out=out), reps=3)
    xe = x.clone(); xe[hurt, 7, 12] = float("inf")       # a live column: the exact IEEE evaluation per (row, system)
    t_e = timed(lambda: ops.multiswag(xe, wa, w2, pd, idx100, philox_seed=3, out=out), reps=3)
    n_items = 100 * int(hurt.numel())
    print(json.dumps({"kernel": "bnn_nonfinite_fixup_kernel", "listed_systems": int(hurt.numel()), "rows": 100, "ms_extra_certain": round(t_c - withscan, 3),
                      "ms_extra_exact": round(t_e - withscan, 3), "exact_items_per_s": round(n_items / max(t_e - withscan, 1e-6) * 1e3, 0),
                      "note": "0.1 % of the systems damaged: NaN in a masked column (direct answer) vs +inf in a live column (full re-evaluation of "
                              "100 rows x listed systems); differences of whole-call times"}), flush=True)


if __name__ == "__main__":
    main()
