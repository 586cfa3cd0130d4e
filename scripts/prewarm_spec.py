#!/usr/bin/env python3
"""Cross-compile (no GPU needed) the specialised forms the tests and the bench scripts use into the in-tree cache
(bnn_chaos_model_amd/csrc/_spec, git-ignored, travels with the tree like the built .so): a GPU box then only loads them."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bnn_chaos_model_amd import specialize as S   # noqa: E402

TEST_NETS = [(40, 20, 1, 1, 41, False), (64, 16, 1, 1, 41, False), (20, 10, 1, 1, 41, False), (33, 7, 1, 1, 41, False), (40, 20, 2, 2, 41, False),
             (30, 12, 0, 0, 41, False), (40, 20, 1, 1, 82, False), (48, 24, 1, 1, 41, True), (100, 30, 1, 1, 41, False), (56, 14, 1, 1, 41, False),
             (128, 32, 1, 1, 41, False)]
t0 = time.time()
rows = S.prewarm(TEST_NETS, noisy=(False, True), w8=(None, False, True))
for net, nz, w, info in rows:
    print(net, "noisy" if nz else "quiet", "w8=%s" % w, info)
print(f"{len(rows)} forms in {S.cache_dir()} ({time.time() - t0:.0f} s)")
