#!/usr/bin/env python3
"""Per-kernel register / spill / LDS numbers of the BUILT library, read from the code-object notes of every gfx950 image in it.
usage: python scripts/resusage.py [lib.so] [name filter]      (tests/test_host_cpu.py imports `kernels()` for its zero-scratch check)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

DEFAULT_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bnn_chaos_model_amd", "csrc", "libbnn_chaos_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(so=DEFAULT_SO):
    """[{name, vgpr, agpr, sgpr, vgpr_spills, sgpr_spills, scratch, lds}] for every gfx950 kernel in the library (demangled names)."""
    data = open(so, "rb").read()
    pos, rows = 0, []
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
                    rows.append((g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                                 g("private_segment_fixed_size"), g("group_segment_fixed_size")))
        pos += 24
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    out = []
    for r, name in zip(rows, names):
        name = re.sub(r"^void ", "", name.strip()).split("(")[0]
        num = lambda v: int(v) if str(v).isdigit() else -1
        out.append(dict(name=name, vgpr=num(r[1]), agpr=num(r[2]), sgpr=num(r[3]), vgpr_spills=num(r[4]), sgpr_spills=num(r[5]),
                        scratch=num(r[6]), lds=num(r[7])))
    return out


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else DEFAULT_SO
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(so):
        if flt in k["name"]:
            print("%-64s vgpr %3d agpr %3d sgpr %3d  spills v%d s%d  scratch %d B  static lds %d B" %
                  (k["name"][:64], k["vgpr"], k["agpr"], k["sgpr"], k["vgpr_spills"], k["sgpr_spills"], k["scratch"], k["lds"]))
