"""Round 6: what a captured HIP graph does with the non-finite record's header, from the record itself.

Round 5's first form of the scan cleared the header with hipMemsetAsync(rec, 0, 16); test_hip_graph_capture_and_replay then failed on
the SECOND replay (system 0 re-evaluated by the fix-up in 14 of 40 rows: the fix-up saw count >= 1 with all-zero entries although x
was finite).  A skipped memset cannot produce that (the header was 0 after replay 1).  This probe reads the header the graph's own
kernels saw, replay by replay:

  micro   : hipMemsetAsync(rec, 0, 16) + ONE elementwise kernel that copies rec[:8] into a log, captured with torch.cuda.graph --
            no product code; rec is refilled with 33 before every replay.  Expected log: [0,0,0,0,33,33,33,33].
  product : [scan(x -> rec); multiswag(nonfinite=rec); log <- rec[:8]] captured once, replayed clean / damaged / clean / damaged on
            the same buffers, compared with eager calls; with the record caller-owned (ordinary memory) and allocated INSIDE the
            capture (the graph's private pool: what ops did per call in round 5).  Run once per library:
              BNN_CHAOS_SO=<csrc/libbnn_nfmemset.so>  (built with -DBNN_NF_HEADER_MEMSET=1: the memset form)   and the product library.
  nodes   : the captured product graph's nodes and edges (hipGraphGetNodes / GetEdges on torch's hipGraph_t), both record forms.

RESULT (round 6, gpurun_out/r06_probe_*.log -> profiles/r06_graph_memset_probe.txt): `micro` alone reproduces it.  A captured
hipMemsetAsync(ptr, 0, n) writes zeros on the FIRST launch of the graph exec and a POINTER-LIKE 64-bit garbage pattern on every later one
(0x78a3_d2e0_0000 ...), for n = 16, 64 and 4096; with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 every replay writes zeros.  The memset node is
not skipped and not reordered: its fill VALUE is wrong from the second replay on -- a defect of the HIP runtime's graph packet capture,
independent of where the memory comes from.  NEVER run `product` with the memset-header library and a damaged x again: the garbage
header sends the scan's append out of bounds (a GPU memory access fault, seen once).

    python scripts/dev/graph_nf_probe3.py micro|product|nodes [tag]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]


def capture(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()                                   # warm-up on a side stream, as torch's recipe prescribes
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        keep = fn()
    return g, keep


if mode == "micro":
    for nbytes in (16, 64, 4096):
        rec = torch.full((2048,), 33, dtype=torch.int32, device="cuda")
        log = torch.zeros(8, dtype=torch.int32, device="cuda")

        def body():
            rc = hip.hipMemsetAsync(rec.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.add(rec[:8], 0, out=log)
        g, _ = capture(body)
        for it in range(5):
            rec.fill_(33 + it)
            log.fill_(-1)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            print(f"micro[{tag}] memset {nbytes:5d} B replay {it + 1}: log {log.tolist()}  rec head after {rec[:6].tolist()}", flush=True)
    # what torch's own zeroing does under capture (a caller might write out.zero_() inside a captured region)
    for name, zero in (("Tensor.zero_() on 4 ints", lambda t: t[:4].zero_()), ("Tensor.zero_() on 2048 ints", lambda t: t.zero_()),
                       ("Tensor.fill_(0)", lambda t: t[:4].fill_(0))):
        rec = torch.full((2048,), 33, dtype=torch.int32, device="cuda")
        log = torch.zeros(8, dtype=torch.int32, device="cuda")

        def body():
            zero(rec)
            torch.add(rec[:8], 0, out=log)
        g, _ = capture(body)
        for it in range(3):
            rec.fill_(33 + it)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            print(f"micro[{tag}] {name} replay {it + 1}: log {log.tolist()}", flush=True)
    sys.exit(0)

from bnn_chaos_model_amd import ops, _native as N  # noqa: E402

z = np.load("tests/golden/swag_v50_0.npz")
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
wa, w2, pd = dev(z["w_avg"][None]), dev(z["w2_avg"][None]), dev(z["pre_D"][None])
gen = torch.Generator().manual_seed(21)
B, J = 700, 40
x_clean = (torch.randn(B, 1, 41, generator=gen) + 0.1 * torch.randn(B, 100, 41, generator=gen)).cuda()
x_bad = x_clean.clone()
x_bad[5, 3, 3] = float("nan")       # masked column (v50 mask): certainly NaN
x_bad[9, 7, 12] = float("inf")      # live column: exact re-evaluation
x = x_clean.clone()
idx = torch.zeros(J, dtype=torch.int32, device="cuda")
out = torch.empty((J, B, 2), device="cuda")
log = torch.zeros(8, dtype=torch.int32, device="cuda")
kw = dict(philox_seed=9, single_launch=False)
plan = ops.get_plan()
ver = C.c_int(0)
hip.hipRuntimeGetVersion(C.byref(ver))
print(f"library: {N.SO_PATH}  build flags: {N.lib().bnn_build_flags().decode()!r}  HIP runtime {ver.value}  torch {torch.__version__}", flush=True)
eager = {"clean": ops.multiswag(x_clean, wa, w2, pd, idx, **kw).clone(), "bad": ops.multiswag(x_bad, wa, w2, pd, idx, **kw).clone()}
torch.cuda.synchronize()

if mode == "nodes":
    # The captured graph of the CURRENT default route, node by node: hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType /
    # hipGraphKernelNodeGetParams + hipKernelNameRefByPtr on torch's own hipGraph_t (CUDAGraph(keep_graph=True).raw_cuda_graph()).
    class Dim3(C.Structure):
        _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]

    class KernelNodeParams(C.Structure):
        _fields_ = [("blockDim", Dim3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", Dim3), ("kernelParams", C.c_void_p),
                    ("sharedMemBytes", C.c_uint)]
    TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait-event", 7: "event-record", 10: "mem-alloc", 11: "mem-free"}
    hip.hipKernelNameRefByPtr.restype = C.c_char_p
    hip.hipKernelNameRefByPtr.argtypes = [C.c_void_p, C.c_void_p]
    import subprocess

    def demangle(name):
        try:
            return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:110]
        except OSError:
            return name[:110]
    for label, rec in (("default route (record from the graph's pool)", None), ("caller-owned record", torch.zeros(4 + B, dtype=torch.int32, device="cuda"))):
        g = torch.cuda.CUDAGraph(keep_graph=True)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        call = (lambda: ops.multiswag(x, wa, w2, pd, idx, out=out, **kw)) if rec is None else \
               (lambda: ops.multiswag(x, wa, w2, pd, idx, out=out, nonfinite=ops.nonfinite_scan(x, out=rec), **kw))
        with torch.cuda.stream(s):
            call()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            call()
        graph = C.c_void_p(g.raw_cuda_graph())
        n = C.c_size_t(0)
        assert hip.hipGraphGetNodes(graph, None, C.byref(n)) == 0
        nodes = (C.c_void_p * n.value)()
        assert hip.hipGraphGetNodes(graph, nodes, C.byref(n)) == 0
        ne = C.c_size_t(0)
        assert hip.hipGraphGetEdges(graph, None, None, C.byref(ne)) == 0
        frm, to = (C.c_void_p * max(ne.value, 1))(), (C.c_void_p * max(ne.value, 1))()
        if ne.value:
            assert hip.hipGraphGetEdges(graph, frm, to, C.byref(ne)) == 0
        ids = {nodes[i]: i for i in range(n.value)}
        print(f"graph[{label}]: {n.value} nodes, {ne.value} edges", flush=True)
        for i in range(n.value):
            t = C.c_int(-1)
            hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t))
            desc = TYPES.get(t.value, str(t.value))
            if t.value == 0:
                kp = KernelNodeParams()
                if hip.hipGraphKernelNodeGetParams(C.c_void_p(nodes[i]), C.byref(kp)) == 0:
                    nm = hip.hipKernelNameRefByPtr(kp.func, None)
                    desc += f" {demangle(nm.decode()) if nm else hex(kp.func or 0)} grid ({kp.gridDim.x},{kp.gridDim.y},{kp.gridDim.z}) block {kp.blockDim.x} lds {kp.sharedMemBytes}"
            print(f"  node {i}: {desc}", flush=True)
        for e in range(ne.value):
            print(f"  edge {ids.get(frm[e])} -> {ids.get(to[e])}", flush=True)
        indeg = {i: 0 for i in range(n.value)}
        outdeg = {i: 0 for i in range(n.value)}
        for e in range(ne.value):
            outdeg[ids[frm[e]]] += 1
            indeg[ids[to[e]]] += 1
        chain = ne.value == n.value - 1 and all(v <= 1 for v in indeg.values()) and all(v <= 1 for v in outdeg.values())
        print(f"  => a single dependency chain: {chain}", flush=True)
        del g
    sys.exit(0)


def scan_into(rec):
    N.check(N.lib().bnn_nonfinite_scan_f32(plan.handle, N.ptr(x), B, 100, N.ptr(rec), N.stream_ptr()))


for where in ("caller-owned record", "record allocated inside the capture"):
    own = torch.full((4 + B,), 77, dtype=torch.int32, device="cuda") if where.startswith("caller") else None

    def body():
        rec = own if own is not None else torch.empty((4 + B,), dtype=torch.int32, device="cuda")
        scan_into(rec)
        ops.multiswag(x, wa, w2, pd, idx, out=out, nonfinite=rec, **kw)
        torch.add(rec[:8], 0, out=log)
        return rec
    x.copy_(x_clean)
    g, rec = capture(body)
    unsafe = "BNN_NF_HEADER_MEMSET" in N.lib().bnn_build_flags().decode() and os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0"
    # (memset header + packet capture: the header is garbage from replay 2 on; with a damaged x the scan's append then goes out of bounds)
    for it, kind in enumerate(("clean",) * 4 if unsafe else ("clean", "bad", "clean", "bad", "clean", "clean")):
        x.copy_(x_clean if kind == "clean" else x_bad)
        out.zero_()
        log.fill_(-1)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(out.nan_to_num(nan=-7.0), eager[kind].nan_to_num(nan=-7.0))
        diff = (out.nan_to_num(nan=-7.0) != eager[kind].nan_to_num(nan=-7.0)).any(-1)
        print(f"product[{tag}] {where}: replay {it + 1} ({kind:5s}) == eager: {same}  header after the chain {log[:4].tolist()} entries {log[4:].tolist()}"
              f"  differing evals {int(diff.sum())} in systems {diff.any(0).nonzero().flatten()[:8].tolist()}", flush=True)
    del g
