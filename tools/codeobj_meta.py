#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every kernel in libcpt.so (the gfx950 code objects' AMDGPU metadata notes).
   python tools/codeobj_meta.py [lib] [substring]   ->  one line per kernel
Used for the spill targets of the perturbation kernels."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "classpp_public_amd", "csrc", "libcpt.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else (sys.argv[1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[1]) else "")
    notes = ""
    with tempfile.TemporaryDirectory() as td:
        # the fat binary sits in section .hip_fatbin: one clang offload bundle per translation unit, back to back
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        pos = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for n, p0 in enumerate(pos):
            nent = struct.unpack_from("<Q", blob, p0 + 24)[0]
            q = p0 + 32
            for _ in range(nent):
                off, size, idsz = struct.unpack_from("<QQQ", blob, q)
                ident = blob[q + 24:q + 24 + idsz].decode()
                q += 24 + idsz
                if "gfx950" in ident and size:
                    co = os.path.join(td, "co%d.o" % n)
                    open(co, "wb").write(blob[p0 + off:p0 + off + size])
                    notes += subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    cur = {}
    rows = []
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.?(\w+):\s*(.*)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "agpr_count" and cur.get("name"):
            rows.append(cur)
            cur = {}
        if key in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                   "group_segment_fixed_size", "max_flat_workgroup_size") and key not in cur:
            cur[key] = val.strip("'")
    if cur.get("name"):
        rows.append(cur)
    for r in rows:
        name = r.get("name", "?")
        try:
            dem = subprocess.check_output([os.path.join(LLVM, "llvm-cxxfilt"), name], text=True).strip()
        except Exception:
            dem = name
        if pat and pat not in dem:
            continue
        print("%-70s vgpr %3s agpr %3s sgpr %3s | spill v %4s s %4s | scratch %5s B | lds %6s B | wg %s" % (
            dem[:70], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"), r.get("vgpr_spill_count"), r.get("sgpr_spill_count"),
            r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size"), r.get("max_flat_workgroup_size")))


if __name__ == "__main__":
    main()
