#!/usr/bin/env python3
"""Run-length picture of the engine-B main loop in a -save-temps .s file: M = MFMA, L = ds_read_b128, w = s_waitcnt, r = v_max_i32 (ReLU),
n = s_nop, F/A/S = fp32 VALU (pk_fma / pk_add / sub), G/g = global loads.  Every non-M instruction costs the matrix pipe an issue slot."""
import collections
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/bk/bnn_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
name = sys.argv[2] if len(sys.argv) > 2 else "_Z21bnn_multiswag4_kernelILi31ELb0ELb0EEv9FwdParams"
s = open(path).read()
a = s.index(name + ":")
body = s[a:s.index(".Lfunc_end", a)]
blocks, cur, lab = [], [], "entry"
for line in body.split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        blocks.append((lab, cur)); cur = []; lab = m.group(1)
    else:
        t = line.strip()
        if t and not t.startswith((".", ";")):
            cur.append(t.split()[0])
blocks.append((lab, cur))
lab, ins = max(blocks, key=lambda b: sum("mfma" in i for i in b[1]))
code = {"v_mfma_f32_4x4x1_16b_f32": "M", "ds_read_b128": "L", "s_waitcnt": "w", "v_max_i32_e32": "r", "s_nop": "n", "v_pk_fma_f32": "F",
        "v_sub_f32_e32": "S", "v_pk_add_f32": "A", "global_load_dwordx4": "G", "global_load_dword": "g", "global_load_dwordx2": "g"}
st = "".join(code.get(x, "?") for x in ins)
print(lab, len(ins), "instructions;", collections.Counter(ins).most_common(14))
print(" ".join(m.group(1) + (str(len(m.group(0))) if len(m.group(0)) > 1 else "") for m in re.finditer(r"(.)\1*", st)))
