#!/usr/bin/env python3
"""VGPR / SGPR / spill counts of every kernel in a hipcc object (reads the gfx950 code object out of the offload bundle).
    python tools/kernel_resources.py smart-vocoder_amd/csrc/conv_wino4.o [filter]"""
import re, struct, subprocess, sys, tempfile
d = open(sys.argv[1], "rb").read()
i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", d, i + 24)[0]
off = i + 32
co = None
for _ in range(n):
    o, s, ts = struct.unpack_from("<QQQ", d, off); off += 24
    t = d[off:off + ts].decode(); off += ts
    if "gfx950" in t:
        co = d[i + o:i + o + s]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co); f.flush()
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], stdout=subprocess.PIPE, text=True).stdout
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], stdout=subprocess.PIPE, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void svoc::", "")
    if flt in name:
        print(f"{name:60s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size')}")
