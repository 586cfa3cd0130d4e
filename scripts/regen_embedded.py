#!/usr/bin/env python3
"""Rewrites bnn_chaos_model_amd/csrc/bnn_fwd_v50spec.hip -- the pretrained network's two specialised forms compiled into the library --
from the generator inside the library (bnn_spec_embedded_source).  Run after changing gen_build / the source generator in
bnn_generic.cpp (tests/test_spec_cpu.py fails when the committed file and the generator disagree); needs no GPU.
The generator is plain host C++: it is compiled on its own here, so this works even when the library itself is stale."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "bnn_chaos_model_amd", "csrc")
MAIN = '#include "bnn_generic.h"\n#include <cstdio>\n#include <vector>\nint main() { std::vector<char> b(1 << 16); ' \
       'if (bnn::gen_spec_embedded_source(b.data(), b.size()) < 0) return 1; fputs(b.data(), stdout); return 0; }\n'
with tempfile.TemporaryDirectory() as td:
    with open(os.path.join(td, "m.cpp"), "w") as f:
        f.write(MAIN)
    exe = os.path.join(td, "gen")
    subprocess.check_call(["g++", "-std=c++17", "-I", CSRC, os.path.join(td, "m.cpp"), os.path.join(CSRC, "bnn_generic.cpp"), "-o", exe])
    text = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
path = os.path.join(CSRC, "bnn_fwd_v50spec.hip")
old = open(path).read() if os.path.exists(path) else ""
if "--check" in sys.argv:
    sys.exit(0 if old == text else 1)
if old != text:
    with open(path, "w") as f:
        f.write(text)
    print("rewrote", path)
else:
    print("up to date:", path)
