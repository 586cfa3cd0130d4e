#!/usr/bin/env python3
"""MultiSWAG ensemble-inference benchmark (BASELINE.json metric: system x MC-sample forward evals/s).

One "step" = one pass of the hot path over one batch: every system of the batch under every weight
draw of the MultiSWAG grid (30 seeds x 100 MC samples) -> (mu, std) per eval, then the predictive
moments per system.  x, the ensemble and the outputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|tiny]

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL): systems are sharded over ranks
(weak scaling: --systems is per GPU), every rank evaluates the same draws on its shard, and the one
exchange of the path -- an all-gather of the per-system predictive moments -- is inside the step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_EVAL = 16408   # 100*41*4 B of x read + 8 B written
ALG_FLOP_PER_EVAL = 814560   # 407 280 MAC: 100*(41*40 + 40*40 + 40*20) + (40*40 + 40*40 + 40*2)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA (= fp32 vector) peak
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # BASELINE.json configs[1]: 30-seed MultiSWAG, 10k systems x 100 MC samples, fp32
    "c2": dict(systems=10_000, seeds=30, samples=100, name="configs[1]: 30-seed MultiSWAG, 10k systems x 100 MC samples, fp32"),
    # configs[2]: 1M systems x 100 samples (x = 16.4 GB, far beyond the 256 MiB Infinity Cache)
    "c3": dict(systems=1_000_000, seeds=30, samples=100, draws=100, name="configs[2]: 1M systems x 100 samples (seeds cycled), fp32"),
    "tiny": dict(systems=512, seeds=30, samples=2, name="smoke-sized grid"),
}


def synthetic_x(B, device, seed):
    """SURVEY.md section 8(d) 'slow' inputs: per-system base + 0.1 noise, column 0 = standardised time."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(B, 1, 41, generator=g, device=device) + 0.1 * torch.randn(B, 100, 41, generator=g, device=device)
    x[:, :, 0] = torch.linspace(-1.71, 1.74, 100, device=device)[None]
    return x.contiguous()


def synthetic_ensemble(S, device):
    """S SWAG states: the two converted pretrained seeds of tests/golden, perturbed per member (seeded)."""
    import numpy as np
    import torch
    gold = os.path.join(ROOT, "tests", "golden")
    base = [np.load(os.path.join(gold, f"swag_v50_{i}.npz")) for i in (0, 12)]
    rng = np.random.default_rng(2024)
    wa, w2, pd = [], [], []
    for s in range(S):
        z = base[s % 2]
        jit = (1.0 + 0.01 * rng.standard_normal(z["w_avg"].shape)).astype(np.float32)
        wa.append(z["w_avg"] * jit)
        w2.append(z["w2_avg"] * jit * jit)
        pd.append(z["pre_D"] * jit[:, None])
    t = lambda a: torch.as_tensor(np.stack(a)).to(device)
    return t(wa), t(w2), t(pd)


def host_threads():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes give a
    16-CPU share of a 256-thread host: running 256 threads there only measures oversubscription)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(round(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(x_cpu, wa, w2, pd, budget_s=15.0):
    """The oracle (oracle/bnn_oracle.c: fp32 C restatement, OpenMP over systems) on a bounded sample of the same
    workload: same synthetic inputs, same ensemble, a few draws, sized to about `budget_s` seconds of CPU work."""
    import numpy as np
    from oracle import oracle as orc
    cores = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library is first loaded
    rng = np.random.default_rng(0)

    def run(Bs, Js):
        seed_idx = (np.arange(Js) % wa.shape[0]).astype(np.int32)
        z1 = rng.standard_normal((Js, wa.shape[1]), dtype=np.float32)
        z2 = rng.standard_normal((Js, pd.shape[2]), dtype=np.float32)
        eps = rng.standard_normal((Js, Bs, 2, 20), dtype=np.float32)
        t0 = time.perf_counter()
        orc.multiswag(x_cpu[:Bs], wa, w2, pd, seed_idx, z1, z2, eps)
        return time.perf_counter() - t0

    Bmax = x_cpu.shape[0]
    run(Bmax, 1)                                        # warm the thread pool
    t_probe = run(Bmax, 8)
    Js = max(8, int(min(1000, 8 * budget_s / t_probe)))  # whole sample = Bmax systems x Js draws, about budget_s seconds
    t = run(Bmax, Js)
    return {"value": Bmax * Js / t, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{Bmax} systems x {Js} draws = {Bmax * Js} evals in {t:.1f} s; same synthetic inputs and ensemble; "
                      f"oracle/bnn_oracle.c (fp32, fmaf chains), OpenMP threads = {cores}"}


def torch_cpu_baseline(x_cpu, wa, w2, pd, budget_s=8.0):
    """The reference's own op sequence (spock_reg_model.py:815-838 elementwise, :884-907) written with torch CPU ops --
    i.e. what the reference's PyTorch-CPU path costs on this host.  Not the oracle, not
    the reference's code: a few lines of eager torch, all host threads."""
    import numpy as np
    import torch
    import torch.nn.functional as Fn
    cores = host_threads()
    torch.set_num_threads(cores)
    x = torch.as_tensor(x_cpu)
    wa_t, w2_t, pd_t = (torch.as_tensor(a) for a in (wa, w2, pd))
    dead = [1, 2, 3, 4, 5, 6, 7, 38, 39, 40]
    K = pd_t.shape[2]

    def one_draw(s):
        with torch.no_grad():
            a, a2, D = wa_t[s], w2_t[s], pd_t[s] - wa_t[s][:, None]
            w = a + 0.5 / np.sqrt(2.0) * torch.randn(a.shape) * torch.sqrt(torch.abs(a2 - a ** 2))
            w = w + 0.5 * (D @ torch.randn(K, 1))[:, 0] / np.sqrt(2 * (K - 1))
            o = 81
            def lin(h, n_out, n_in):
                nonlocal o
                W = w[o:o + n_out * n_in].reshape(n_out, n_in); o += n_out * n_in
                b = w[o:o + n_out]; o += n_out
                return Fn.linear(h, W, b)
            xm = x.clone(); xm[..., dead] = 0
            h = lin(torch.relu(lin(torch.relu(lin(xm, 40, 41)), 40, 40)), 20, 40)
            mu_, var_ = h.mean(1), h.std(1) ** 2
            n = h.shape[1]
            m = torch.randn_like(mu_) * torch.sqrt(var_ / n) + mu_
            v = torch.randn_like(var_) * torch.sqrt(2 * var_ ** 2 / (n - 1)) + var_
            sstat = torch.cat((m, torch.sqrt(torch.abs(v) + 1e-5)), 1)
            r = lin(torch.relu(lin(torch.relu(lin(sstat, 40, 40)), 40, 40)), 2, 40)
            return torch.cat((0.5 * (torch.tanh(r[:, [0]]) + 1) * 8 + 4, 0.5 * (torch.tanh(r[:, [1]]) + 1) * 5.5 + 0.5), 1)

    one_draw(0)
    t0 = time.perf_counter(); one_draw(1); t1 = time.perf_counter() - t0
    Js = max(2, int(min(200, budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    for j in range(Js):
        one_draw(j % wa_t.shape[0])
    t = time.perf_counter() - t0
    B = x.shape[0]
    return {"value": B * Js / t, "unit": "evals/s", "cores": cores, "kind": "port (eager torch CPU ops in the reference's order)",
            "sample": f"{B} systems x {Js} draws in {t:.1f} s, torch {torch.__version__}, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--systems", type=int, default=0, help="systems per GPU (overrides the workload)")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--unfused", action="store_true", help="separate ops.swag_draw + ops.forward calls")
    ap.add_argument("--single-launch", action="store_true", help="in-kernel draw in every workgroup prologue (no workspace)")
    ap.add_argument("--spb", type=int, default=0, help="systems per workgroup (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bnn_chaos_model_amd import ops
    from bnn_chaos_model_amd.distributed import all_gather_moments

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched by torch.distributed.run with N ranks")
    # BNN_BENCH_REHEARSE=1: every rank on cuda:0 with gloo -- exercises the N>1 code path on a one-GPU box
    rehearse = os.environ.get("BNN_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            try:
                dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
                probe = torch.zeros(1, dtype=torch.float64, device=dev)
                dist.all_reduce(probe)                            # fail here, not inside the timed region
                torch.cuda.synchronize()
            except Exception as e:  # the one exchange of the path is 160 KB per rank: gloo via host memory still measures the job
                print(f"[bench] RCCL unavailable ({type(e).__name__}: {e}); falling back to gloo for the moments gather", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                dist.init_process_group("gloo")

    wl = dict(WORKLOADS[args.workload])
    if args.systems:
        wl["systems"] = args.systems
    if args.samples:
        wl["samples"] = args.samples
    B, S, M = wl["systems"], wl["seeds"], wl["samples"]
    J = wl.get("draws", S * M)

    x = synthetic_x(B, dev, seed=123 + rank)          # this rank's shard: global systems [rank*B, (rank+1)*B)
    wa, w2, pd = synthetic_ensemble(S, dev)           # replicated ensemble (29 MB)
    seed_idx = (torch.arange(J, dtype=torch.int32) % S).to(dev)  # dense grid: every seed x every sample
    out = torch.empty((J, B, 2), dtype=torch.float32, device=dev)
    plan = ops.get_plan()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i, timed):
        if timed:
            ev0[i].record()
        if args.unfused:
            W = ops.swag_draw(wa, w2, pd, seed_idx, philox_seed=99, draw_id0=0, plan=plan)
            o = ops.forward(x, W, philox_seed=99, draw_id0=0, system_id0=rank * B, plan=plan, systems_per_block=args.spb)
        else:
            o = ops.multiswag(x, wa, w2, pd, seed_idx, philox_seed=99, draw_id0=0, system_id0=rank * B, plan=plan, out=out,
                              systems_per_block=args.spb, single_launch=args.single_launch)
        if timed:
            ev1[i].record()
        mom = ops.moments(o)
        if world > 1:
            mom = all_gather_moments(mom, world * B)  # the path's one exchange: [B,4] float64 per rank
        return mom

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(0, False)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        mom = step(i, True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps

    evals_per_step = world * B * J
    value = evals_per_step * args.steps / dt
    if rank == 0:
        evals_per_launch = B * J
        ach_tflops = evals_per_launch * ALG_FLOP_PER_EVAL / (kern_ms * 1e-3) / 1e12
        ach_gbs = evals_per_launch * ALG_BYTES_PER_EVAL / (kern_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if rec.get("workload") == args.workload and rec.get("systems") == B and rec.get("draws") == J:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "system x MC-sample forward evals/sec", "value": value, "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "systems_per_gpu": B, "seeds": S, "mc_samples": M, "draws": J, "timesteps": 100,
                       "features": 41, "noise": "in-kernel Philox4x32-10", "kernel": "ops.swag_draw + ops.forward" if args.unfused else ("multiswag, in-kernel draw per workgroup" if args.single_launch else "multiswag, draw-once workspace + forward"),
                       "sharding": f"systems over {world} rank(s), all-gather of moments",
                       "collective": (dist.get_backend() if world > 1 else "none")},
            "roofline": {"bound": "mfma", "achieved": ach_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "kernel_ms": kern_ms, "flop_per_eval": ALG_FLOP_PER_EVAL,
                         "hbm_algorithmic_GBs": ach_gbs, "hbm_frac_of_8TBs": ach_gbs / PEAK_HBM_GBS},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(x.cpu().numpy(), wa.cpu().numpy(), w2.cpu().numpy(), pd.cpu().numpy())
            except Exception as e:  # the oracle is a checker, never a dependency of the measured path
                res["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
            try:
                res["cpu_baseline_torch"] = torch_cpu_baseline(x.cpu().numpy(), wa.cpu().numpy(), w2.cpu().numpy(), pd.cpu().numpy())
            except Exception as e:
                res["cpu_baseline_torch"] = {"value": None, "sample": f"failed: {e}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
