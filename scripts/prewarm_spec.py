#!/usr/bin/env python3
"""Cross-compile (no GPU needed) the specialised forms the tests and the bench scripts use into the in-tree cache
(bnn_chaos_model_amd/csrc/_spec, git-ignored, travels with the tree like the built .so): a GPU box then only loads them."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bnn_chaos_model_amd import specialize as S   # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
TEST_NETS = [(40, 20, 1, 1, 41, False), (64, 16, 1, 1, 41, False), (33, 7, 1, 1, 41, False), (40, 20, 2, 2, 41, False), (30, 12, 0, 0, 41, False),
             (40, 20, 1, 1, 82, False), (48, 24, 1, 1, 41, True), (72, 20, 1, 1, 41, False)]       # tests/test_hip_spec.py NETS
BENCH_NETS = [(20, 10, 1, 1, 41, False), (128, 32, 1, 1, 41, False)]                                    # scripts/spec_bench_r04.sh
t0 = time.time()
full = "--bench" in sys.argv
rows = S.prewarm(TEST_NETS + (BENCH_NETS if full else []), noisy=(False, True), w8=(None, False, True) if full else (None,))
rows += S.prewarm([(56, 14, 1, 1, 41, False)], noisy=(False,), w8=(False, True))
rows += S.prewarm([(64, 16, 1, 1, 41, False, 0), (64, 16, 1, 1, 41, False, 1 << 3)], noisy=(False, True), w8=(None,))
import test_hip_spec   # noqa: E402  (tests/: the fixture networks with their own masks)
rows += S.prewarm(test_hip_spec.fixture_archs(), noisy=(False, True), w8=(None,))
rows += S.prewarm([(24, 6, 1, 1, 41, False)], noisy=(False,), w8=(None,))
for a in test_hip_spec.random_archs():      # (some lie outside the LDS budget: the test expects the error)
    try:
        rows += S.prewarm([a[:7]], noisy=(False, True), w8=(None,))
    except Exception as e:
        print("skipped", a[:6], str(e)[:60])
for net, nz, w, info in rows:
    print(net, "noisy" if nz else "quiet", "w8=%s" % w, info)
print(f"{len(rows)} forms in {S.cache_dir()} ({time.time() - t0:.0f} s)")
