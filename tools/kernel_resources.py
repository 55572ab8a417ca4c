#!/usr/bin/env python
"""Registers, LDS, scratch (spills) of every kernel in libnerf_hip.so, read from the code objects' metadata notes
(no GPU needed):  python tools/kernel_resources.py [substring]"""
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "project-nerf_amd", "libnerf_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + 24)
        p = pos + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + 1)


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    blob = open(LIB, "rb").read()
    for i, co in enumerate(code_objects(blob)):
        tmp = f"/tmp/nerf_co_{i}.o"
        open(tmp, "wb").write(co)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", tmp], capture_output=True, text=True).stdout
        for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            get = lambda key: (re.search(r"\." + key + r":\s+(\S+)", block) or [None, "?"])[1]
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
            if want in name:
                print(f"{name[:100]:100s} vgpr {get('vgpr_count'):>4s} agpr {block.split()[0]:>3s} sgpr {get('sgpr_count'):>4s} "
                      f"lds {get('group_segment_fixed_size'):>6s} scratch {get('private_segment_fixed_size'):>5s} "
                      f"spills v{get('vgpr_spill_count')} s{get('sgpr_spill_count')}")
        os.remove(tmp)


if __name__ == "__main__":
    main()
