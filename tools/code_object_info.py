#!/usr/bin/env python3
"""Per-kernel resource usage of a built library / object (VGPRs, AGPRs, SGPRs, LDS, scratch = spills) from the code object metadata:
    python tools/code_object_info.py [diffusion-rs_amd/libflux_mi355x.so | build/attention.o] [name filter]
Extracts every gfx950 code object from the clang offload bundles in the file and prints llvm-readelf's notes, one line per kernel."""
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
path = sys.argv[1] if len(sys.argv) > 1 else "diffusion-rs_amd/libflux_mi355x.so"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = open(path, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos, objs = 0, []
while True:
    i = d.find(magic, pos)
    if i < 0:
        break
    cnt = struct.unpack_from("<Q", d, i + 24)[0]
    off = i + 32
    for _ in range(cnt):
        o, sz, tl = struct.unpack_from("<QQQ", d, off)
        off += 24
        trip = d[off:off + tl].decode()
        off += tl
        if "gfx950" in trip:
            objs.append(d[i + o:i + o + sz])
    pos = i + 24
print(f"{path}: {len(objs)} gfx950 code object(s)")
print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>8}  kernel")
for blob in objs:
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(blob)
        f.flush()
        out = subprocess.run([LLVM + "llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    for blk in out.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if flt in name:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{g('vgpr_count'):>5} {blk.split()[0]:>5} {g('sgpr_count'):>5} {g('group_segment_fixed_size'):>7} {g('private_segment_fixed_size'):>8}  {dem[:110]}")
