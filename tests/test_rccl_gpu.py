"""RCCL on the one-GPU box: the calls the N > 1 bench makes (init with device_id, barrier, all_gather_into_tensor of the float64
moments, all_reduce(MAX), per-rank all_gather, HIP-event timing of the gather) executed for real at world size 1, as a child process
under torch.distributed.run exactly as the driver launches `bench.py --gpus N`.  Needs an MI355X."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_under_launcher(*bench_args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--force-dist",
           "--rendezvous-timeout", "60"] + list(bench_args)
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, (out.returncode, out.stderr[-2000:])
    return json.loads(lines[-1])


@pytest.mark.parametrize("workload,extra", (("tiny", ()), ("c5", ("--systems", "3000", "--samples", "4")), ("c4q", ("--systems", "2048", "--samples", "2"))))
def test_bench_gather_runs_on_rccl(workload, extra):
    """moments gather (tiny), the whole-simulation bands gather (c5 shape, 1000 simulations x 3 trios) and the streamed-bands gather
    (c4q: slab driver with the fused statistics tail) through RCCL."""
    r = run_under_launcher("--workload", workload, "--steps", "2", "--warmup", "1", *extra)
    c = r["config"]
    assert c["collective"] == "nccl" and c["degraded"] is False and c["ranks_seen"] == 1 and r["n_gpus"] == 1
    assert c["gather_ms"] > 0.0 and c["kernel_ms_min"] == c["kernel_ms_max"] > 0.0
    assert c["gather_bytes_per_rank"] == {"c5": 3000 // 3 * 6 * 4, "c4q": 2048 * 6 * 4, "tiny": 512 * 4 * 8}[workload]
    assert r["value"] > 0 and r["roofline"]["traffic_measured_in_run"] is False


def test_two_rank_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` launching its own two ranks (both on cuda:0, gather staged through gloo: BNN_BENCH_REHEARSE=1): the N > 1
    code path end to end -- self-launch, sharding by whole simulations, per-rank kernel times, gather timing, one JSON line."""
    env = dict(os.environ, BNN_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5", "--systems", "3000", "--samples", "4", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--rendezvous-timeout", "60"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stderr[-2000:])
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["collective"] == "gloo" and c["degraded"] is True
    assert c["gather_ms"] > 0 and 0 < c["kernel_ms_min"] <= c["kernel_ms_max"]
    assert c["gather_bytes_per_rank"] == 1000 * 6 * 4 and "whole simulations" in c["sharding"]
    assert r["value"] == pytest.approx(2 * 3000 * 4 * 2 / (r["ms_per_step"] * 2e-3), rel=1e-6)   # evals of BOTH ranks over the max-rank time


def test_plain_multi_gpu_invocation_carries_the_named_config():
    """`bench.py --gpus 2` with NO --workload (what a driver runs): the headline is configs[2] weak-scaled, and the line also carries
    `named_config` = one step of BASELINE configs[3]'s per-GPU share (rehearsed here at 8 192 systems per rank, both ranks on cuda:0)."""
    env = dict(os.environ, BNN_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", BNN_BENCH_NAMED_SYSTEMS="8192")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--systems", "4096", "--samples", "4", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--rendezvous-timeout", "60"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stderr[-2000:])
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and "configs[2]" in r["config"]["workload"]
    nc = r["named_config"]
    assert "configs[3]" in nc["workload"] and nc["draws"] == 3000 and nc["systems_per_gpu"] == 8192 and nc["steps"] == 1
    assert nc["value"] == pytest.approx(2 * 8192 * 3000 / (nc["ms_per_step"] * 1e-3), rel=1e-6)


def test_single_process_route_and_the_new_line_fields():
    """`bench.py --single-process --gpus 3` (three logical shards on the one card): the one-process route of multidevice.DeviceSet with the
    rank-per-GPU line's schema -- launcher, exchange, per-device kernel times, the staging probe -- and, on the one-GPU default line, the
    fields round 5 added: the non-finite scan timed on its own, the clock / power held under load, frac_at_held_clock."""
    env = dict(os.environ, BNN_BENCH_REHEARSE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "3", "--workload", "c2", "--systems", "6000", "--samples", "2",
           "--steps", "2", "--warmup", "1", "--h2d-probe-gb", "0.05"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stderr[-2000:])
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == 3 and c["launcher"] == "single-process" and c["devices"] == ["cuda:0"] * 3 and c["exchange"] == "peer copies"
    assert r["scaling"] == "weak" and r["value"] == pytest.approx(3 * 6000 * 60 / (r["ms_per_step"] * 1e-3), rel=1e-6)
    assert 0 < c["kernel_ms_min"] <= c["kernel_ms_max"] and "scanned" in c["finite_check"]
    assert set(c["h2d_probe"]) == {"pinned", "pageable"} and c["h2d_probe"]["pinned"]["aggregate_GBs"] > 0 and "3 host thread" in c["h2d_probe"]["pageable"]["mode"]
    # the default one-GPU line (shrunk): scan on, timed on its own; --assume-finite switches it off
    for flag, want_scan in ((), True), (("--assume-finite",), False):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--systems", "50000", "--samples", "20", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *flag]
        out = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="4"), cwd=ROOT, capture_output=True, text=True, timeout=600)
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert (r["config"]["finite_check_ms"] > 0.03) == want_scan   # (an empty event pair reads a few microseconds; 0.8 GB of x cannot be scanned in 30) and ("scanned" in r["config"]["finite_check"]) == want_scan
        assert r["config"]["launcher"] == "one process, one GPU" and r["roofline"]["kernel_ms"] > 0
        if "clock" in r:   # (amdsmi is there on the GPU boxes; the field is optional by design)
            assert r["clock"]["sclk_mhz_mean"] > 0 and r["roofline"]["frac_at_held_clock"] > 0   # (present and sane; rates are box-dependent: not asserted)
