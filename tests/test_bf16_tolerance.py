"""BASELINE.json configs[4] asks for a bf16-vs-fp32 tolerance sweep.  The GPU side of it is tests/test_lowp.py (the opt-in
bf16 / split-bf16 kernels of bnn_lowp.hip.h); this CPU test keeps the oracle-side record: what rounding the inputs and/or the
weights to bf16 (fp32 accumulation, fp32 activations) does to the outputs of the pinned oracle."""
import numpy as np

from conftest import load_golden, tape
from oracle import oracle as orc


def to_bf16(a):
    """Round-to-nearest-even to bfloat16, returned as float32."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def test_bf16_rounding_error_budget(inputs, capsys):
    z = load_golden("case_swagfast_v50_0_slow.npz")
    tp = tape(z)
    x, w = inputs["slow"], z["w"]
    e1, e2 = tp[2][1], tp[3][1]
    ref = orc.forward(x, w, e1, e2).astype(np.float64)
    rows = []
    for name, xx, ww in (("x bf16", to_bf16(x), w), ("w bf16", x, to_bf16(w)), ("x and w bf16", to_bf16(x), to_bf16(w))):
        out = orc.forward(xx, ww, e1, e2).astype(np.float64)
        err = np.abs(out - ref)
        rows.append((name, err[:, 0].max(), err[:, 1].max(), np.median(err[:, 0])))
    with capsys.disabled():
        for name, emu, esd, med in rows:
            print(f"\n  bf16 sweep [{name:13s}] max |d mu| = {emu:.3e}   max |d std| = {esd:.3e}   median |d mu| = {med:.3e}", end="")
    # three to four orders of magnitude above the 1e-5 parity bar: a bf16-operand kernel can only be an opt-in approximate mode
    assert rows[2][1] > 1e-3 and rows[2][1] < 2.0


def test_split_emulation_converges_to_the_fp32_oracle(inputs):
    """oracle/lowp.py (the checker of the GPU's reduced-precision kernels) against the pinned fp32 oracle: with three bf16 parts
    per operand (24 significant bits, six products) the pooled summary equals the fp32 oracle's to fp32 rounding; with two parts
    to ~2^-16; with one (plain bf16) to ~2^-8 of the row scale."""
    from oracle import lowp
    z = load_golden("case_swagfast_v50_0_slow.npz")
    x, w = inputs["slow"], z["w"]
    B = x.shape[0]
    zero = np.zeros((B, 20), np.float32)
    _, ex = orc.forward(x, w, zero, zero, extras=True)
    ref = ex["summary"].astype(np.float64)
    scale = np.abs(ref).max(1, keepdims=True)
    errs = {}
    for ns in (1, 2, 3):
        got = lowp.pooled_summary(lowp.feature_nn(x, w, ns))
        errs[ns] = (np.abs(got - ref) / scale).max()
    assert errs[3] < 2e-6 and errs[2] < 1e-4 and 1e-4 < errs[1] < 5e-2, errs
    assert errs[3] < errs[2] < errs[1]
