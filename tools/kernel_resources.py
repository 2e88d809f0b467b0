#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a hipcc object or shared library (no GPU needed).

    python3 tools/kernel_resources.py kmerind_amd/libkmerind_hip.so [name-filter ...]

The gfx950 code objects are cut out of the clang offload bundle(s) inside the file and their metadata notes are read with
llvm-readelf; one line per kernel: VGPRs (AGPRs included), SGPRs, LDS bytes, scratch bytes, waves per SIMD the registers allow.
"""
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx" in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def main():
    path = sys.argv[1]
    filters = sys.argv[2:]
    blob = open(path, "rb").read()
    rows = []
    for triple, co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, re.S):
            blk = m.group(0)
            def g(key):
                mm = re.search(r"\.%s:\s+(\S+)" % key, blk)
                return mm.group(1) if mm else "?"
            name = g("name")
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            except OSError:
                pass
            name = re.sub(r"^void kmi::", "", name).split("(")[0]
            if filters and not any(x in name for x in filters):
                continue
            v = int(g("vgpr_count"))
            alloc = (v + 7) // 8 * 8
            rows.append((name, v, int(g("sgpr_count")), int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size")),
                         min(8, 512 // max(alloc, 1))))
    for r in sorted(set(rows)):
        print("%-70s vgpr %3d sgpr %3d lds %6d scratch %4d waves/simd %d" % r)


if __name__ == "__main__":
    main()
