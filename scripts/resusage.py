#!/usr/bin/env python3
"""Per-kernel register / spill / LDS numbers of the BUILT library, read from the code-object notes of every gfx950 image in it.
usage: python scripts/resusage.py [lib.so] [name filter]"""
import os, re, struct, subprocess, sys, tempfile

so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bnn_chaos_model_amd", "csrc", "libbnn_chaos_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
data = open(so, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pos = 0
rows = []
while True:
    pos = data.find(MAGIC, pos)
    if pos < 0:
        break
    n = struct.unpack_from("<Q", data, pos + 24)[0]
    p = pos + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple and size:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[pos + off:pos + off + size]); f.flush()
                txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
                rows.append((g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    pos += 24
for r in rows:
    name = subprocess.run(["c++filt", r[0]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void ", "", name).split("(")[0]
    if flt in name:
        print("%-64s vgpr %3s agpr %3s sgpr %3s  spills v%s s%s  scratch %s B  static lds %s B" % ((name[:64],) + r[1:]))
