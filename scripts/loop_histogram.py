#!/usr/bin/env python3
"""Instruction histogram of a kernel's hottest loop (the backward branch spanning the most 4x4x1 MFMAs), read from the BUILT library:
the loop body executes once per 64-row tile, so its static counts are the per-tile dynamic counts the PMC totals divide into.
usage: python scripts/loop_histogram.py <mangled-name substring> [lib.so]"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "..", "bnn_chaos_model_amd", "csrc", "libbnn_chaos_hip.so")
target = sys.argv[1]
data = open(so, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pos, dis = 0, None
while dis is None:
    pos = data.find(MAGIC, pos)
    if pos < 0:
        sys.exit("kernel not found")
    n = struct.unpack_from("<Q", data, pos + 24)[0]
    p = pos + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        blob = data[pos + off:pos + off + size]
        if "gfx950" in triple and size and target.encode() in blob:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob); f.flush()
                dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    pos += 24
lines = dis.split("\n")
start = [i for i, l in enumerate(lines) if target in l and l.rstrip().endswith(">:")][0]
sym0 = int(lines[start].split()[0], 16)
end = start + 1
while end < len(lines) and not re.match(r"^[0-9a-f]+ <", lines[end]):
    end += 1
body = [l for l in lines[start + 1:end] if "//" in l]
offs = [int(re.search(r"// ([0-9A-F]+):", l).group(1), 16) - sym0 for l in body]
index = {o: i for i, o in enumerate(offs)}
best = None
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w* .*\+0x([0-9a-f]+)>", l)
    if m and int(m.group(1), 16) < offs[i] and int(m.group(1), 16) in index:
        j = index[int(m.group(1), 16)]
        nm = sum("v_mfma_f32_4x4x1" in x for x in body[j:i + 1])
        inner = any(re.search(r"s_c?branch\w* .*\+0x([0-9a-f]+)>", x) and int(re.search(r"\+0x([0-9a-f]+)>", x).group(1), 16) < offs[j + k]
                    and int(re.search(r"\+0x([0-9a-f]+)>", x).group(1), 16) >= offs[j] for k, x in enumerate(body[j:i]))
        if not inner and (best is None or nm > best[0]):
            best = (nm, j, i)
nm, j, i = best
cls = collections.Counter()
for l in body[j:i + 1]:
    op = l.strip().split()[0]
    if op.startswith("v_mfma"): c = "MFMA (4x4x1)"
    elif re.match(r"v_(sin|cos|log|sqrt|rcp|exp)_f32", op): c = "transcendental (v_sin / v_cos / v_log / v_sqrt): 2 issue slots"
    elif op.startswith("v_mad_u64_u32"): c = "v_mad_u64_u32 (Philox products): 2 issue slots"
    elif op.startswith("v_bitop3"): c = "v_bitop3_b32 (Philox xors)"
    elif op.startswith("v_pk_"): c = "packed fp32 (noise application, Welford, r * (cos, sin))"
    elif op.startswith("v_max_i32"): c = "v_max_i32 (ReLU)"
    elif op.startswith("v_"): c = "other vector (field alignment, and / and-or, sub, mul, moves)"
    elif op.startswith("ds_"): c = "LDS"
    elif op.startswith(("global_", "buffer_", "flat_")): c = "VMEM"
    elif op.startswith("s_waitcnt"): c = "s_waitcnt"
    elif op.startswith("s_nop"): c = "s_nop"
    else: c = "scalar"
    cls[c] += 1
print(f"tile loop of {target}: {i - j + 1} instructions, offsets +0x{offs[j]:x} .. +0x{offs[i]:x}")
for k, v in cls.most_common():
    print(f"  {v:5d}  {k}")
