#!/usr/bin/env python3
"""MultiSWAG ensemble-inference benchmark (BASELINE.json metric: system x MC-sample forward evals/s).

One "step" = one pass of the hot path over one batch: every system of the batch under every weight draw of the grid
-> (mu, std) per eval, then the predictive moments per system.  x, the ensemble and the outputs are resident in HBM before
the timed region.  Default workload = BASELINE.json configs[2], the largest single-GPU configuration:
1 000 000 systems x 100 MC samples (x = 16.4 GB, far beyond the 256 MiB Infinity Cache).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c4q|c5|noisy|tiny] [--engine generic] [--net H,L,IN,OUT[,F]]

  c3 (default)  configs[2]: 1M systems x 100 samples per GPU, samples kept, moments gathered
  c4            configs[3], one GPU's share: 1.25M systems x 3000 draws (30 seeds x 100) through the native slab driver
                (bnn_multiswag_moments_f64: 250 draws per launch, float64 moments), then the ONE all-gather of moments
  c4q           the same share streamed to what the scripts consume (bnn_multiswag_bands_f32: fused statistics tail + quantile sketch),
                then the all-gather of the per-system bands
  c5            configs[4], one GPU's share: 125 000 five-planet systems = 375 000 rows x 100 samples x 10 chunks; under --gpus N the
                shards are WHOLE simulations (3 trios each), every rank reduces its samples to per-simulation bands and the bands
                are gathered

--engine generic runs the same workload through the generic forward engine (weights streamed from LDS: what every network other than
the pretrained ensemble's, and every series length T % 4 != 0, runs on); --net 64,16,1,1 benches another hparams-built network
(hidden, latent, depth in, depth out[, features 41|82]) on a seeded synthetic ensemble.  Neither is a BASELINE configuration.

--single-process --gpus N drives the N GPUs from THIS process instead (multidevice.DeviceSet: one shard of the systems resident on
every device, launches enqueued device after device, ONE exchange of the per-system moments onto the first device) -- the route an
unchanged single-process evaluation script takes with devices="all"; same JSON schema, config.launcher = "single-process".

--assume-finite skips the once-per-step scan of x for NaN / +-inf (the product default scans: ops.nonfinite_scan); the line says which.

--gpus N > 1 from a plain invocation launches N rank processes itself (a torch.distributed.run child, started BEFORE this
process touches the GPU); under torch.distributed.run (WORLD_SIZE set) it is a rank.  One rank per GPU over RCCL: systems
are sharded over ranks (weak scaling: the workload's systems are per GPU), every rank evaluates the same draws on its shard,
and the one exchange of the path -- an all-gather of the per-system predictive moments -- is inside the step.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_EVAL = 16408   # 100*41*4 B of x read + 8 B written
ALG_FLOP_PER_EVAL = 814560   # 407 280 MAC: 100*(41*40 + 40*40 + 40*20) + (40*40 + 40*40 + 40*2)
EXEC_FLOP_PER_EVAL = {31: 734560, 41: 814560}  # MACs the kernel issues: the v50 mask multiplies 31 of the 41 input columns
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA (= fp32 vector) peak
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # BASELINE.json configs[2]: 1M systems x 100 samples (one ensemble member per sample, members cycled)
    "c3": dict(systems=1_000_000, seeds=30, samples=100, draws=100, name="configs[2]: 1M systems x 100 MC samples (30-seed ensemble cycled), fp32, fused draw+f1+pool+f2"),
    # configs[1]: 30-seed MultiSWAG, 10k systems x 100 MC samples per seed
    "c2": dict(systems=10_000, seeds=30, samples=100, name="configs[1]: 30-seed MultiSWAG, 10k systems x 100 MC samples, fp32"),
    # forward(noisy_val=True) with every normal generated in-kernel; not a BASELINE config
    "noisy": dict(systems=10_000, seeds=30, samples=10, noisy=True, name="f4: forward(noisy_val=True), 10k systems x 300 draws, in-kernel Philox noise"),
    # BASELINE.json configs[4], one GPU's share: 125 000 five-planet systems = 375 000 rows, 100 samples x 10 chunks, one random
    # ensemble member + one weight draw per chunk per sample (figures/multiswag_5_planet.py:295-298).  fp32 by default;
    # --precision bf16 | bf16x3 | bf16x6 | f16 | f16x3 runs the OPT-IN reduced-precision forward (never the default, never the headline).
    "c5": dict(systems=375_000, seeds=30, samples=100, chunks=10, trios=3, steps=10, warmup=2,
               name="configs[4] share: 5-planet shapes, 125k simulations x 3 trios = 375k rows x 100 samples x 10 chunks -> bands per simulation"),
    # BASELINE.json configs[3], one GPU's share: 10M systems / 8 = 1.25M systems (x = 20.5 GB) x 30 seeds x 100 samples = 3000 draws,
    # reduced to float64 predictive moments slab by slab (distributed.MultiSwagSharded.local_moments -> bnn_multiswag_moments_f64)
    "c4": dict(systems=1_250_000, seeds=30, samples=100, slab=250, steps=3, warmup=1,
               name="configs[3] share: 1.25M systems x 30 seeds x 100 samples (3000 draws in slabs of 250) -> float64 moments, all-gather"),
    # the same share reduced to what the evaluation scripts consume instead of moments: statistics epilogue fused in the forward tail,
    # per-system quantile sketch (distributed.MultiSwagSharded.local_bands -> bnn_multiswag_bands_f32), bands + mean gathered
    "c4q": dict(systems=1_250_000, seeds=30, samples=100, slab=250, bands=True, steps=3, warmup=1,
                name="configs[3] share, streamed to bands: 1.25M systems x 3000 draws -> truncated-normal draw + prior + quantile sketch -> "
                     "2.5/16/50/84/97.5 % bands + mean per system, all-gather"),
    "tiny": dict(systems=512, seeds=30, samples=2, name="smoke-sized grid"),
}
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
DTYPE_OF = {"f32": "f32", "bf16": "bf16 operands, f32 accumulate (feature_nn); f32 elsewhere",
            "bf16x3": "split-bf16 x3 (16 significant bits), f32 accumulate (feature_nn); f32 elsewhere",
            "bf16x6": "split-bf16 x6 (24 significant bits), f32 accumulate (feature_nn); f32 elsewhere",
            "f16": "f16 operands, f32 accumulate (feature_nn); f32 elsewhere",
            "f16x3": "split-f16 x3 (22 significant bits), f32 accumulate (feature_nn); f32 elsewhere"}
PRODUCTS_OF = {"f32": 1, "bf16": 1, "bf16x3": 3, "bf16x6": 6, "f16": 1, "f16x3": 3}


def net_flop_per_eval(F, H, L, din, dout, T=100, live_cols=None):
    """2 x MACs of one evaluation of mlp(F, L, H, din) per timestep + mlp(2L, 2, H, dout) once (spock_reg_model.py:301-321, 359-360)."""
    def mlp_macs(i, o, h, layers):
        return i * o if layers == 0 else i * h + layers * h * h + h * o
    f = mlp_macs(F, L, H, din) - ((F - live_cols) * (L if din == 0 else H) if live_cols else 0)
    return 2 * (T * f + mlp_macs(2 * L, 2, H, dout))


def synthetic_net_ensemble(S, d, K, device):
    """A seeded synthetic SWAG ensemble for a network no pretrained checkpoint has (throughput does not depend on the values)."""
    import torch
    g = torch.Generator(device=device).manual_seed(2025)
    wa = 0.15 * torch.randn(S, d, generator=g, device=device)
    w2 = wa ** 2 + (0.02 * torch.rand(S, d, generator=g, device=device)) ** 2
    pd = wa[:, :, None] + 0.05 * torch.randn(S, d, K, generator=g, device=device)
    return wa, w2, pd


def synthetic_x(B, device, seed, F=41):
    """SURVEY.md section 8(d) 'slow' inputs: per-system base + 0.1 noise, column 0 = standardised time.  Built in slabs of
    systems so that the temporaries stay small next to the 16.4 GB result at B = 1e6."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.empty((B, 100, F), dtype=torch.float32, device=device)
    t0 = torch.linspace(-1.71, 1.74, 100, device=device)[None]
    slab = 65536
    for b0 in range(0, B, slab):
        n = min(slab, B - b0)
        xs = x[b0:b0 + n]
        xs.normal_(generator=g).mul_(0.1)
        xs.add_(torch.randn(n, 1, F, generator=g, device=device))
        xs[:, :, 0] = t0
    return x


def synthetic_ensemble(S, device):
    """S SWAG states.  With tests/golden/ensemble_v50.npz present (the 30 converted pretrained seeds) the real ensemble is used;
    otherwise the two converted seeds of tests/golden, perturbed per member (seeded): throughput does not depend on the values."""
    import numpy as np
    import torch
    gold = os.path.join(ROOT, "tests", "golden")
    real = os.path.join(gold, "ensemble_v50.npz")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
    if os.path.exists(real):
        z = np.load(real)
        if z["w_avg"].shape[0] >= S:
            return t(z["w_avg"][:S]), t(z["w2_avg"][:S]), t(z["pre_D"][:S])
    base = [np.load(os.path.join(gold, f"swag_v50_{i}.npz")) for i in (0, 12)]
    rng = np.random.default_rng(2024)
    wa, w2, pd = [], [], []
    for s in range(S):
        z = base[s % 2]
        jit = (1.0 + 0.01 * rng.standard_normal(z["w_avg"].shape)).astype(np.float32)
        wa.append(z["w_avg"] * jit)
        w2.append(z["w2_avg"] * jit * jit)
        pd.append(z["pre_D"] * jit[:, None])
    return t(np.stack(wa)), t(np.stack(w2)), t(np.stack(pd))


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


def cpu_baseline(x_cpu, wa, w2, pd, budget_s=15.0, net=None):
    """The oracle (oracle/bnn_oracle.c: fp32 C restatement, OpenMP over systems) on a BOUNDED sample of the same workload:
    the first systems of the same synthetic batch, the same ensemble, as many draws as fit in about `budget_s` seconds."""
    import numpy as np
    from oracle import oracle as orc
    cores = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library is first loaded
    rng = np.random.default_rng(0)
    arch = orc.make_arch(T=x_cpu.shape[1], **net) if net else None
    latent = net["latent"] if net else 20

    def run(Bs, Js):
        seed_idx = (np.arange(Js) % wa.shape[0]).astype(np.int32)
        z1 = rng.standard_normal((Js, wa.shape[1]), dtype=np.float32)
        z2 = rng.standard_normal((Js, pd.shape[2]), dtype=np.float32)
        eps = rng.standard_normal((Js, Bs, 2, latent), dtype=np.float32)
        t0 = time.perf_counter()
        orc.multiswag(x_cpu[:Bs], wa, w2, pd, seed_idx, z1, z2, eps, arch=arch)
        return time.perf_counter() - t0

    Bmax = x_cpu.shape[0]
    run(Bmax, 1)                                        # warm the thread pool
    t_probe = run(Bmax, 4)
    Js = max(4, int(min(1000, 4 * budget_s / t_probe)))  # whole sample = Bmax systems x Js draws, about budget_s seconds
    t = run(Bmax, Js)
    return {"value": Bmax * Js / t, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"first {Bmax} systems of the batch x {Js} draws = {Bmax * Js} evals in {t:.1f} s; same synthetic inputs and ensemble; "
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
            "sample": f"first {B} systems of the batch x {Js} draws in {t:.1f} s, torch {torch.__version__}, {cores} threads"}


def library_id():
    """Which native library produced the numbers: the default in-tree build reports {"variant": false}; an A/B or ablation build
    (BNN_CHAOS_SO, extra compile-time switches) is named so that its line cannot pass for a headline number."""
    from bnn_chaos_model_amd import _native as N
    flags = N.lib().bnn_build_flags().decode()
    override = os.environ.get("BNN_CHAOS_SO")
    return {"variant": bool(flags or override), "build_flags": flags, "path": os.path.relpath(N.SO_PATH, ROOT) if override else "default"}


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Plain `python bench.py --gpus N`: start N ranks as a torch.distributed.run child and return its exit code.  This
    process has not touched the GPU (nothing but `import` has run), so the children are fresh processes, not re-execs."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env, cwd=ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 10; 3 for c4, whose step is ~23 s)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 2; 1 for c4)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default c3 (configs[2]); with --gpus N > 1 and no --workload the line "
                    "also carries `named_config`: one step of configs[3]'s per-GPU share (workload c4), the BASELINE 8-GPU configuration")
    ap.add_argument("--systems", type=int, default=0, help="systems per GPU (overrides the workload)")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--unfused", action="store_true", help="separate ops.swag_draw + ops.forward calls")
    ap.add_argument("--single-launch", action="store_true", help="in-kernel draw in every workgroup prologue (no workspace)")
    ap.add_argument("--spb", type=int, default=0, help="systems per workgroup (0 = auto)")
    ap.add_argument("--precision", default="f32", choices=sorted(DTYPE_OF), help="opt-in reduced-precision forward (workload c5 / c3 / c2)")
    ap.add_argument("--engine", default="auto", choices=("auto", "generic", "spec"),
                    help="generic: force the generic forward engine (LDS-streamed weights); spec: its run-time-compiled form for this network (specialize.py)")
    ap.add_argument("--spec-w8", default="auto", choices=("auto", "0", "1", "2"),
                    help="--engine spec: eight waves at 256 registers (1), four at 512 (0), sixteen at 128 (2: small networks), builder's choice")
    ap.add_argument("--timesteps", type=int, default=100, help="series length T (100 = every BASELINE config; others: the ragged lengths of the reference's "
                    "`augment`, which the generic engine / its specialised forms take)")
    ap.add_argument("--net", default="", help="hidden,latent,in,out[,features]: another hparams-built network on a synthetic ensemble (generic engine)")
    ap.add_argument("--assume-finite", action="store_true", help="skip the once-per-step scan of x for NaN / +-inf (default: scan, as the module surface does)")
    ap.add_argument("--single-process", action="store_true", help="drive the --gpus N devices from this one process (multidevice.DeviceSet) instead of N ranks")
    ap.add_argument("--h2d-probe-gb", type=float, default=1.0, help="--single-process: GB per device staged from host memory (pinned, then pageable) "
                    "outside the timed region, to report the PCIe-inclusive rate; 0 = skip")
    ap.add_argument("--no-clock-sample", action="store_true", help="skip the sclk / power sample taken under load after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-systems", type=int, default=8192, help="systems of the batch the CPU baselines are timed on")
    ap.add_argument("--allow-gloo", action="store_true", help="accept a gloo moments gather when RCCL cannot start (the line says degraded)")
    ap.add_argument("--launcher-selftest", action="store_true", help="ranks only rendezvous (gloo, CPU) and report; no GPU work")
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0, help="seconds a rank waits for the others in init_process_group")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gather even at world size 1 (under a launcher): "
                    "exercises the N > 1 code path -- init, collective, event timing -- on a one-GPU box")
    args = ap.parse_args(argv)
    args.workload_defaulted = args.workload is None
    if args.workload is None:
        args.workload = "c3"
    wl = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = wl.get("steps", 10)
    if args.warmup is None:
        args.warmup = wl.get("warmup", 2)
    return args


def clock_sample_under_load(dev_index, enqueue_steps, n_steps=2, period_s=0.02):
    """sclk / socket power of the device WHILE it runs `n_steps` more (untimed) steps of the same workload, read through amdsmi -- taken
    right after the timed region, never inside it.  The same library measures 549-599 ms per configs[2] step from box to box (the chip is
    power-limited under this kernel and holds 2.26-2.34 GHz): with the held clock in the line a swing reads as "box", not "regression".
    Returns None when amdsmi is not there or the device cannot be matched."""
    import torch
    read = None
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        p = torch.cuda.get_device_properties(dev_index)
        want = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", -1), getattr(p, "pci_device_id", 0))
        h = None
        for cand in handles:
            if str(amdsmi.amdsmi_get_gpu_device_bdf(cand)).lower().startswith(want):
                h = cand
        if h is None and len(handles) == 1:
            h = handles[0]
        if h is not None:
            read = lambda: (float(amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)["clk"]),
                            float(amdsmi.amdsmi_get_power_info(h)["current_socket_power"]))
            read()
    except Exception:
        read = None
    # EVERY caller enqueues the extra steps, whether or not it can read the counters: under torch.distributed the steps hold a collective,
    # and a rank that skipped them (amdsmi missing on it alone) would leave the others waiting
    done = torch.cuda.Event()
    enqueue_steps(n_steps)           # asynchronous: the host is free to poll while the GPU works
    done.record()
    clk, pw = [], []
    t_end = time.perf_counter() + 60.0
    while read is not None and not done.query() and time.perf_counter() < t_end:
        try:
            c, w = read()
            clk.append(c); pw.append(w)
        except Exception:
            break
        time.sleep(period_s)
    torch.cuda.synchronize()
    if len(clk) < 3:
        return None
    clk, pw = clk[1:], pw[1:]        # (the first read may still show the idle clock)
    return {"sclk_mhz_mean": sum(clk) / len(clk), "sclk_mhz_min": min(clk), "sclk_mhz_max": max(clk), "power_w_mean": sum(pw) / len(pw),
            "power_w_max": max(pw), "reads": len(clk),
            "source": f"amdsmi (gfx clock, current socket power) polled every {int(period_s * 1e3)} ms during {n_steps} extra untimed steps right after the timed region"}


def main_single_process(args):
    """--single-process --gpus N: the N devices driven from this one process (multidevice.DeviceSet), one shard of B systems resident
    per device (weak scaling, like the rank-per-GPU form), every device evaluating the same draws, ONE exchange of [B, 4] float64
    moments onto the first device.  Dense fp32 workloads with kept samples (c3, c2, tiny)."""
    import torch
    from bnn_chaos_model_amd import ops
    from bnn_chaos_model_amd.multidevice import DeviceSet
    wl = dict(WORKLOADS[args.workload])
    if wl.get("chunks", 1) != 1 or wl.get("slab") or wl.get("noisy") or args.net or args.precision != "f32" or args.engine != "auto":
        sys.exit("--single-process runs the dense fp32 workloads (c3, c2, tiny)")
    if args.systems:
        wl["systems"] = args.systems
    if args.samples:
        wl["samples"] = args.samples
    B, S, M = wl["systems"], wl["seeds"], wl["samples"]
    J = wl.get("draws", S * M)
    n = args.gpus
    # BNN_BENCH_REHEARSE=1: every logical shard on cuda:0 -- exercises this code path on a one-GPU box
    devs = [0] * n if os.environ.get("BNN_BENCH_REHEARSE") == "1" else list(range(n))
    ds = DeviceSet(devs)
    plans = [ops.get_plan(device=d) for d in ds.devices]
    xs, outs, seeds = [], [], []
    ens_host = synthetic_ensemble(S, "cpu")
    state = ds.replicate("ensemble", ens_host)
    for i, d in enumerate(ds.devices):
        with torch.cuda.device(d):
            xs.append(synthetic_x(B, d, seed=123 + i))
            outs.append(torch.empty((J, B, 2), dtype=torch.float32, device=d))
            seeds.append((torch.arange(J, dtype=torch.int32) % S).to(d))
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ds.devices] for _ in range(args.steps)]

    def step(k, timed):
        def shard(i, d, lo, hi):
            if timed:
                ev[k][i][0].record()
            wa, w2, pd = state[i]
            o = ops.multiswag(xs[i], wa, w2, pd, seeds[i], philox_seed=99, draw_id0=0, system_id0=lo, plan=plans[i], out=outs[i],
                              assume_finite=args.assume_finite)
            m = ops.moments(o)
            if timed:
                ev[k][i][1].record()
            return m
        return ds.gather_rows(ds.run(n * B, shard))

    def fence():
        for d in ds.devices:
            torch.cuda.synchronize(d)

    for _ in range(args.warmup):
        step(0, False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        res = step(k, True)
    fence()
    dt = time.perf_counter() - t0
    assert res.shape == (n * B, 4)
    kern = [sum(ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(args.steps)) / args.steps for i in range(len(ds))]
    value = n * B * J * args.steps / dt
    ach = B * J * ALG_FLOP_PER_EVAL / (max(kern) * 1e-3) / 1e12
    exe = B * J * EXEC_FLOP_PER_EVAL[31] / (max(kern) * 1e-3) / 1e12
    h2d = None
    if args.h2d_probe_gb > 0:   # the PCIe-inclusive side of the route: rows of a host-resident batch onto every device at once (DeviceSet.stage)
        rows = max(n, int(args.h2d_probe_gb * 1e9 / (100 * 41 * 4))) * n
        h2d = {}
        for mode in ("pinned", "pageable"):
            hx = torch.empty((rows, 100, 41), dtype=torch.float32, pin_memory=(mode == "pinned"))
            hx.normal_()
            fence()
            t1 = time.perf_counter()
            shards = ds.stage(hx)
            fence()
            wall = (time.perf_counter() - t1) * 1e3
            info = ds.h2d_ms()
            h2d[mode] = {"bytes_per_device": hx.numel() * 4 // n, "wall_ms": wall, "slowest_device_ms": info["ms"], "mode": info["mode"],
                         "aggregate_GBs": hx.numel() * 4 / wall / 1e6}
            del shards, hx
    clock = None if args.no_clock_sample else clock_sample_under_load(ds.devices[0].index, lambda k: [step(0, False) for _ in range(k)])
    res = {
        "metric": "system x MC-sample forward evals/sec", "value": value, "unit": "evals/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["name"], "launcher": "single-process", "systems_per_gpu": B, "seeds": S, "mc_samples": M, "draws": J, "chunks": 1,
                   "timesteps": 100, "features": 41, "kernel": "multiswag, draw-once workspace + forward, one shard per device from one process",
                   "noise": "in-kernel Philox4x32-10", "library": library_id(), "devices": [str(d) for d in ds.devices],
                   "finite_check": "assumed finite (--assume-finite)" if args.assume_finite else "x scanned once per step per device (ops.nonfinite_scan)",
                   "sharding": f"systems over {n} device(s) of one process, {ds.last_exchange} of moments [systems, 4] float64 onto the first device",
                   "exchange": ds.last_exchange, "gather_bytes_per_rank": B * 4 * 8, "kernel_ms_min": min(kern), "kernel_ms_max": max(kern),
                   "h2d_probe": h2d,
                   "timing_note": "ms_per_step = wall clock of the whole step (launches on every device + the exchange), fenced by a synchronize of "
                                  "every device on both sides; kernel_ms_* = HIP events around a device's launches (min / max over devices)"},
        "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                     "traffic_measured_in_run": False, "achieved_executed": exe, "frac_executed": exe / PEAK_F32_MFMA_TFLOPS, "kernel_ms": max(kern),
                     "note": "per device (the slowest one's launches); see the rank-per-GPU line for the counter-based figures"},
    }
    if clock:
        res["clock"] = clock
        res["roofline"]["frac_at_held_clock"] = exe * 1e12 / (clock["sclk_mhz_mean"] * 1e6 * 1024 * 64)
    print(json.dumps(res), flush=True)


def main():
    args = parse()
    if args.single_process:
        return main_single_process(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        sys.exit(f"bench.py --gpus {args.gpus} is running under a launcher with WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import datetime

    import torch
    import torch.distributed as dist
    pg_timeout = datetime.timedelta(seconds=args.rendezvous_timeout)   # a missing rank fails the job in seconds, not in 10-30 minutes

    if args.launcher_selftest:
        dist.init_process_group("gloo", timeout=pg_timeout)
        ones = torch.ones(1)
        dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": int(ones.item()), "collective": "gloo"}), flush=True)
        dist.destroy_process_group()
        return

    from bnn_chaos_model_amd import ops
    from bnn_chaos_model_amd.distributed import all_gather_moments, shard_bounds

    # BNN_BENCH_REHEARSE=1: every rank on cuda:0 with gloo -- exercises the N>1 code path on a one-GPU box
    rehearse = os.environ.get("BNN_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    degraded = False
    ranks_seen = 1
    use_dist = world > 1 or (args.force_dist and "WORLD_SIZE" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo", timeout=pg_timeout)
            degraded = True
        else:
            try:
                dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)   # RCCL over xGMI
                probe = torch.ones(1, dtype=torch.float64, device=dev)
                dist.all_reduce(probe)                            # fail here, not inside the timed region
                torch.cuda.synchronize()
            except Exception as e:
                if not args.allow_gloo:   # a gloo number must never pass for an xGMI measurement
                    print(f"[bench] rank {rank}: RCCL could not start ({type(e).__name__}: {e}); rerun with --allow-gloo to time the "
                          "job with the moments gather on gloo (marked degraded)", file=sys.stderr, flush=True)
                    sys.exit(3)
                print(f"[bench] RCCL unavailable ({type(e).__name__}: {e}); gloo for the moments gather (degraded)", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                dist.init_process_group("gloo", timeout=pg_timeout)
                degraded = True
        ones = torch.ones(1, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    on_nccl = use_dist and dist.get_backend() == "nccl"

    wl = dict(WORKLOADS[args.workload])
    if args.systems:
        wl["systems"] = args.systems
    if args.samples:
        wl["samples"] = args.samples
    B, S, M = wl["systems"], wl["seeds"], wl["samples"]
    nch = wl.get("chunks", 1)
    trios = wl.get("trios", 1)
    slab = wl.get("slab", 0)                      # > 0: the native slab driver reduces the draws on the fly (c4: moments, c4q: bands)
    stream_bands = bool(wl.get("bands"))
    if B % trios:
        sys.exit(f"--systems must be a multiple of {trios} for workload {args.workload} (whole simulations per rank)")
    J = wl.get("draws", S * M) if nch == 1 else M * nch
    R = J // nch                                  # output rows = samples per system
    noisy = bool(wl.get("noisy"))
    lowp = args.precision != "f32"
    if lowp and (noisy or args.unfused or args.single_launch or slab):
        sys.exit("--precision applies to the fused quiet forward with kept samples only (workloads c3, c2, c5)")

    # Weak scaling: every rank holds B systems (B / trios whole simulations); this rank's global systems are [lo, hi).
    lo, hi = shard_bounds(world * B, world, trios)[rank]
    assert hi - lo == B
    net = None
    if args.net:
        v = [int(t) for t in args.net.split(",")]
        net = dict(hidden=v[0], latent=v[1], depth_in=v[2], depth_out=v[3], n_features=v[4] if len(v) > 4 else 41)
        if lowp or slab or trios > 1 or args.single_launch:
            sys.exit("--net applies to the dense fp32 workloads (c3, c2, noisy, tiny)")
    NF = net["n_features"] if net else 41
    plan = ops.get_plan(**net) if net else ops.get_plan()
    if args.engine == "spec":   # compile this network's own form of the generic engine (cached on disk) before anything is timed
        t0 = time.time()
        ops.specialize(plan, noisy=(noisy,), w8={"auto": None, "0": False, "1": True, "2": 2}[args.spec_w8])
        spec_compile_s = time.time() - t0
    x = synthetic_x(B, dev, seed=123 + rank, F=NF)    # this rank's shard
    T_ = args.timesteps
    if T_ != 100:
        if not 2 <= T_ <= 100:
            raise SystemExit("--timesteps must be in [2, 100]")
        x = x[:, :T_].contiguous()
    wa, w2, pd = synthetic_net_ensemble(S, plan.d, 30, dev) if net else synthetic_ensemble(S, dev)   # replicated ensemble (29 MB)
    if nch == 1:
        seed_idx = (torch.arange(J, dtype=torch.int32) % S).to(dev)  # dense grid: every seed x every sample
    else:                                                            # one random member per chunk per sample (regression.py:78)
        import numpy as np
        seed_idx = torch.as_tensor(np.random.default_rng(7).integers(0, S, J).astype(np.int32)).to(dev)
    out = None if slab else torch.empty((R, B, 2), dtype=torch.float32, device=dev)
    W_noisy = ops.swag_draw(wa, w2, pd, seed_idx, philox_seed=99, plan=plan) if noisy else None
    sketch = ops.QuantileSketch(B, group=trios, device=dev) if (trios > 1 or stream_bands) else None
    stats = ops.stats_params(device=dev) if (trios > 1 or stream_bands) else None
    BANDS_Q = (2.5, 16.0, 50.0, 84.0, 97.5)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]   # behind the non-finite scan of a step, in front of its compute launches
    fin = dict(assume_finite=True) if args.assume_finite else None            # else: nonfinite=<this step's record>
    gv0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    gv1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    gather_host_ms = []

    def gather(local, n_total, i, timed):
        """The path's one exchange.  RCCL: HIP events on the current stream bracket it (the collective's stream is joined to the
        current stream on both sides); gloo (rehearsal / degraded): staged through the host, so a host clock around it."""
        if not use_dist:
            return local
        if on_nccl:
            if timed:
                gv0[i].record()
            res = all_gather_moments(local, n_total, force=args.force_dist)
            if timed:
                gv1[i].record()
            return res
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = all_gather_moments(local, n_total)
        if timed:
            gather_host_ms.append((time.perf_counter() - t) * 1e3)
        return res

    def step(i, timed):
        if timed:
            ev0[i].record()          # on torch's current stream = the stream the ops launch on (ops.N.stream_ptr())
        # the product default: x is scanned for NaN / +-inf once per step (one streaming read), and every forward launch below is followed
        # by the (empty-handed, here) fix-up launch; timed on its own so that roofline.kernel_ms stays the compute launches'
        fk = fin or dict(nonfinite=ops.nonfinite_scan(x, plan=plan))
        if timed:
            evs[i].record()
        if slab and stream_bands:    # c4q: slabs of draws -> statistics epilogue in the forward tail -> quantile sketch, ONE native call
            sketch.hist.zero_(); sketch.mom.zero_(); sketch.count = 0
            ops.multiswag_bands(x, wa, w2, pd, seed_idx, sketch, st=stats, philox_seed=99, draw_id0=0, system_id0=lo, draws_per_launch=slab, plan=plan, **fk)
            bands = torch.cat([sketch.percentiles(BANDS_Q), sketch.mean().float()[:, None]], 1)   # [B, 6]
            if timed:
                ev1[i].record()
            return gather(bands, world * B, i, timed)
        if slab:                     # c4: slabs of draws -> float64 moments inside ONE native call
            mom = ops.multiswag_moments(x, wa, w2, pd, seed_idx, philox_seed=99, draw_id0=0, system_id0=lo, draws_per_launch=slab, plan=plan, **fk)
            if timed:
                ev1[i].record()
            return gather(mom, world * B, i, timed)
        if noisy:
            o = ops.forward(x, W_noisy, philox_seed=99, draw_id0=0, system_id0=lo, plan=plan, noisy=True, systems_per_block=args.spb, engine=args.engine, **fk)
        elif args.unfused:
            W = ops.swag_draw(wa, w2, pd, seed_idx, philox_seed=99, draw_id0=0, plan=plan)
            o = ops.forward(x, W, philox_seed=99, draw_id0=0, system_id0=lo, plan=plan, systems_per_block=args.spb, engine=args.engine, **fk)
        else:
            o = ops.multiswag(x, wa, w2, pd, seed_idx, nchunks=nch, philox_seed=99, draw_id0=0, system_id0=lo, plan=plan,
                              out=None if lowp else out, systems_per_block=args.spb, single_launch=args.single_launch or None,
                              precision=args.precision, engine=args.engine, **fk)
        if timed:
            ev1[i].record()
        if trios > 1:   # c5: what the 5-planet script does with the samples (multiswag_5_planet.py:388-428, 484-489), per simulation
            sketch.hist.zero_(); sketch.mom.zero_(); sketch.count = 0
            sketch.update(ops.stats_draw(o, st=stats, philox_seed=99, row_id0=0, system_id0=lo))   # truncnorm + prior, min over trios
            bands = torch.cat([sketch.percentiles(BANDS_Q), sketch.mean().float()[:, None]], 1)   # [sims, 6]
            return gather(bands, world * B // trios, i, timed)
        return gather(ops.moments(o), world * B, i, timed)  # [B,4] float64 per rank

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(0, False)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res_last = step(i, True)
    fence()
    dt = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(evs, ev1)) / args.steps    # the compute launches of a step (draw, forward, fix-up, reductions inside it)
    scan_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, evs)) / args.steps   # the non-finite scan of a step (0 with --assume-finite)
    clock = None
    if not args.no_clock_sample and not rehearse:
        clock = clock_sample_under_load(local_rank, lambda k: [step(0, False) for _ in range(k)], n_steps=2 if not slab else 1)
        fence()
    if use_dist:
        gather_ms = (sum(a.elapsed_time(b) for a, b in zip(gv0, gv1)) if on_nccl else sum(gather_host_ms)) / args.steps
        cdev = dev if on_nccl else "cpu"
        t = torch.tensor([dt, gather_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, gather_ms = float(t[0].item()), float(t[1].item())
        km = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(km, torch.tensor([kern_ms], dtype=torch.float64, device=cdev))
        kern_ms_ranks = [float(v.item()) for v in km]
    else:
        gather_ms, kern_ms_ranks = 0.0, [kern_ms]
    expect_rows = world * B // trios
    if res_last.shape[0] != expect_rows:
        sys.exit(f"[bench] rank {rank}: gathered {res_last.shape[0]} rows, expected {expect_rows}")

    # A plain `bench.py --gpus N` (what a driver runs) measures configs[2] weak-scaled.  BASELINE.json's own 8-GPU configuration is
    # configs[3]: one step of its per-GPU share (workload c4: 1.25M systems x 3000 draws -> float64 moments, all-gather) rides along
    # as `named_config`, timed the same way (barrier + synchronize on both sides, max over ranks), outside the headline's timed region.
    gather_bytes_per_rank = int(res_last.shape[0] // world * res_last.shape[1] * res_last.element_size())
    named = None
    if world > 1 and args.workload_defaulted and not net and os.environ.get("BNN_BENCH_NO_NAMED") != "1":
        c4 = WORKLOADS["c4"]
        B4 = int(os.environ.get("BNN_BENCH_NAMED_SYSTEMS", c4["systems"]))   # (the rehearsal on a shared card shrinks it)
        del x, out
        res_last = None
        torch.cuda.empty_cache()
        lo4, hi4 = shard_bounds(world * B4, world)[rank]
        x4 = synthetic_x(B4, dev, seed=123 + rank)
        J4 = c4["seeds"] * c4["samples"]
        seed4 = (torch.arange(J4, dtype=torch.int32) % c4["seeds"]).to(dev)
        ops.multiswag_moments(x4[:4096], wa, w2, pd, seed4[:c4["slab"]], philox_seed=99, system_id0=lo4, draws_per_launch=c4["slab"], plan=plan)   # warm
        fence()
        t4 = time.perf_counter()
        mom4 = ops.multiswag_moments(x4, wa, w2, pd, seed4, philox_seed=99, draw_id0=0, system_id0=lo4, draws_per_launch=c4["slab"], plan=plan)
        g4 = all_gather_moments(mom4, world * B4)
        fence()
        dt4 = time.perf_counter() - t4
        tt = torch.tensor([dt4], dtype=torch.float64, device=dev if on_nccl else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt4 = float(tt.item())
        assert g4.shape[0] == world * B4
        named = {"workload": c4["name"], "systems_per_gpu": B4, "draws": J4, "steps": 1, "warmup": 0, "ms_per_step": dt4 * 1e3,
                 "value": world * B4 * J4 / dt4, "unit": "evals/s", "scaling": "weak",
                 "frac": world * B4 * J4 * ALG_FLOP_PER_EVAL / dt4 / 1e12 / (world * PEAK_F32_MFMA_TFLOPS),
                 "note": "BASELINE.json configs[3] (10M systems x 30 seeds x 100 samples over 8 GPUs): ONE step of the per-GPU share through "
                         "the native slab driver + the all-gather of [systems, 4] float64 moments; whole-job evals/s, max over ranks"}

    evals_per_step = world * B * R   # every system under every sample (a chunked draw covers 1/nch of the systems)
    value = evals_per_step * args.steps / dt
    if rank == 0:
        evals_per_launch = B * R
        kin = 41 if noisy else 31
        T_ = args.timesteps
        ragged = T_ % 4 != 0 or T_ < 8     # the pretrained network's own kernels take whole tiles of four timesteps: other lengths go to the generic route
        generic = bool(net) or args.engine != "auto" or ragged
        spec_form = args.engine == "spec" or (args.engine == "auto" and ragged and not net)   # (the embedded forms of the pretrained network)
        alg_flop, exe_flop, alg_bytes = ALG_FLOP_PER_EVAL, EXEC_FLOP_PER_EVAL[kin], ALG_BYTES_PER_EVAL
        pad4 = lambda n: 4 * ((n + 3) // 4)
        if net:   # another network: the same two counts from its shapes (the v50 mask leaves 31 of the first 41 columns live)
            a5 = (NF, net["hidden"], net["latent"], net["depth_in"], net["depth_out"])
            alg_flop = net_flop_per_eval(*a5, T=T_)
            alg_bytes = T_ * NF * 4 + 8
        else:
            a5 = (41, 40, 20, 1, 1)
            if T_ != 100:
                alg_flop, alg_bytes = net_flop_per_eval(*a5, T=T_), T_ * 41 * 4 + 8
        if generic:   # columns layer 0 multiplies: whole input quads; the specialised quiet form drops the masked columns first; whole tiles of 4 timesteps
            exe_flop = net_flop_per_eval(*a5, T=pad4(T_), live_cols=pad4(a5[0] - 10) if (spec_form and not noisy) else pad4(a5[0]))
        ach_tflops = evals_per_launch * alg_flop / (kern_ms * 1e-3) / 1e12
        exe_tflops = evals_per_launch * exe_flop / (kern_ms * 1e-3) / 1e12
        ach_gbs = evals_per_launch * alg_bytes / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch: NOT measured in this run.  The PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, gfx950
        # correction of MI355X_MICROARCH.md) run the same command under the profiler (scripts/profile_r03.sh) and leave the
        # per-launch figure in profiles/pmc_traffic.json; it is quoted here with its source so that the line is self-describing.
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf)).get(args.workload)
                if rec and rec.get("systems") == B and rec.get("draws") == J and not lowp:
                    traffic, traffic_src = rec.get("hbm_bytes_per_launch"), rec.get("source")
            except Exception:
                traffic = None
        kernel = ("bnn_forward_kernel<41,noisy> (ops.forward, noisy_val=True, in-kernel Philox)" if noisy else
                  "ops.swag_draw + ops.forward" if args.unfused else
                  f"bnn_multiswag_bands_f32: {slab} draws per launch (draw + forward with fused statistics tail + sketch update), {J // slab} launches per step" if (slab and stream_bands) else
                  f"bnn_multiswag_moments_f64: {slab} draws per launch (draw + forward + moments kernels), {J // slab} launches per step" if slab else
                  ("multiswag, in-kernel draw per workgroup" if args.single_launch else "multiswag, draw-once workspace + forward"))
        if generic:
            kernel = (("bnn_spec_forward: the generic engine compiled at run time for this network (specialize.py, %.1f s incl. cache lookup; w8 %s)" % (spec_compile_s, args.spec_w8)
                       if args.engine == "spec" else "bnn_spec_forward_v50q/n: the pretrained network's specialised forms compiled into the library (ragged series length)" if spec_form else "bnn_forward_generic_kernel (weight registers streamed from an LDS image; draw-once workspace)") +
                      (f", network hidden={net['hidden']} latent={net['latent']} in={net['depth_in']} out={net['depth_out']} features={NF}" if net else
                       ", the pretrained network forced onto the generic engine"))
        if lowp:
            kernel = f"bnn_forward_lowp_kernel ({args.precision}): exact fp32 draw + feature_nn on the bf16 matrix pipe, {PRODUCTS_OF[args.precision]} product(s) per layer"
        payload = ("bands [sims, 5 percentiles + mean] float32" if (trios > 1 or stream_bands) else "moments [systems, 4] float64")
        res = {
            "metric": "system x MC-sample forward evals/sec", "value": value, "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_OF[args.precision], "data": "synthetic",
            "config": {"workload": wl["name"], "systems_per_gpu": B, "seeds": S, "mc_samples": M, "draws": J, "chunks": nch, "timesteps": args.timesteps,
                       "features": NF, "kernel": kernel,
                       "noise": ("in-kernel Philox: Philox4x32-7 for the input-noise stream (4 100 of the 4 180 normals of a noisy evaluation), "
                                 "Philox4x32-10 for every other stream" if noisy else "in-kernel Philox4x32-10"),
                       "library": library_id(), "launcher": "torch.distributed ranks" if use_dist else "one process, one GPU",
                       "finite_check": ("assumed finite (--assume-finite): no scan" if args.assume_finite else
                                        "x scanned for NaN / +-inf once per step (bnn_nonfinite_scan_f32: one streaming read), every forward launch "
                                        "followed by the fix-up launch (the reference's x - mask / NaN-propagating ReLU for the listed systems)"),
                       "finite_check_ms": scan_ms,
                       "sharding": (f"whole simulations ({trios} trios each) over {world} rank(s), all-gather of {payload}" if trios > 1 else
                                    f"systems over {world} rank(s), all-gather of {payload}"),
                       "collective": (dist.get_backend() if use_dist else "none"), "degraded": degraded, "ranks_seen": ranks_seen,
                       "gather_ms": gather_ms, "gather_bytes_per_rank": gather_bytes_per_rank,
                       "kernel_ms_min": min(kern_ms_ranks), "kernel_ms_max": max(kern_ms_ranks),
                       "timing_note": "ms_per_step = wall clock of the whole step, max over ranks; kernel_ms_* = HIP events around the compute "
                                      "launches per rank (min / max over ranks); gather_ms = the all-gather alone, max over ranks"},
            "roofline": {"bound": "mfma", "achieved": ach_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_measured_in_run": False,
                         "traffic_source": traffic_src,
                         "achieved_executed": exe_tflops, "frac_executed": exe_tflops / PEAK_F32_MFMA_TFLOPS,
                         "kernel_ms": kern_ms, "flop_per_eval": alg_flop, "flop_per_eval_executed": exe_flop,
                         "note": "frac counts the algorithm's 814 560 flop/eval (SURVEY 8d); frac_executed counts the MACs the kernel issues "
                                 "(the v50 mask drops 10 of 41 input columns); kernel_ms = HIP events around the compute launches of a step "
                                 "(rank 0; the non-finite scan in front of them is config.finite_check_ms and inside ms_per_step); "
                                 "frac_at_held_clock = executed flop/s over (the sclk held under this load x 1024 SIMDs x 64 flop/clk): what "
                                 "separates a slow box from a slow kernel; traffic = HBM bytes per launch from separate rocprofv3 PMC passes of this command, null when "
                                 "no such pass is on file for this workload",
                         "hbm_algorithmic_GBs": ach_gbs, "hbm_frac_of_8TBs": ach_gbs / PEAK_HBM_GBS},
        }
        if clock:
            res["clock"] = clock
            res["roofline"]["frac_at_held_clock"] = exe_tflops * 1e12 / (clock["sclk_mhz_mean"] * 1e6 * 1024 * 64)
        if named:
            res["named_config"] = named
        if lowp:   # priced against the bf16 matrix pipe; issued flops = algorithmic x products; these forms are vector-issue bound
            peak = PEAK_BF16_MFMA_TFLOPS
            iss = exe_tflops * PRODUCTS_OF[args.precision]
            res["roofline"].update({"peak": peak, "frac": ach_tflops / peak, "achieved_executed": iss, "frac_executed": iss / peak, "traffic": None,
                                    "note": "opt-in reduced precision (DESIGN.md 4.6): outputs are NOT within the 1e-5 parity bar; frac = algorithmic "
                                            "flop/eval over the dense bf16 MFMA peak, frac_executed counts the split products; the kernels are "
                                            "bound by vector issue (convert / ReLU / split / pool), not by the matrix pipe or HBM"})
        if not args.no_cpu_baseline and world == 1:
            ns = min(B, args.cpu_sample_systems)
            xs = x[:ns].cpu().numpy()
            try:
                res["cpu_baseline"] = cpu_baseline(xs, wa.cpu().numpy(), w2.cpu().numpy(), pd.cpu().numpy(), net=net)
            except Exception as e:  # the oracle is a checker, never a dependency of the measured path
                res["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
            if not net:   # (the eager-torch port below is written for the pretrained network's shapes)
                try:
                    res["cpu_baseline_torch"] = torch_cpu_baseline(xs, wa.cpu().numpy(), w2.cpu().numpy(), pd.cpu().numpy())
                except Exception as e:
                    res["cpu_baseline_torch"] = {"value": None, "sample": f"failed: {e}"}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
