"""Registers, spills, LDS and scratch of the gfx950 kernels inside a shared library or object built by hipcc.

    python tools/kernel_resources.py l3ster_amd/lib/libl3k.so [name-substring ...]

Reads the clang offload bundle out of the file, takes the gfx950 code object and prints the AMDGPU metadata notes
(llvm-readelf --notes) per kernel: VGPRs (+AGPRs), SGPRs, spilled registers, static LDS, scratch bytes.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and size:
                yield data[pos + o:pos + o + size]
        pos += len(MAGIC)


def kernels(path):
    out = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        try:
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        finally:
            os.unlink(f.name)
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                            vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"), lds=g("group_segment_fixed_size"),
                            scratch=g("private_segment_fixed_size")))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    ks = kernels(path)
    for k, dn in zip(ks, demangle([k["name"] for k in ks])):
        if pats and not all(p in dn for p in pats):
            continue
        print(f"vgpr {k['vgpr']:>3} agpr {k['agpr']:>3} sgpr {k['sgpr']:>3} vspill {k['vspill']:>3} sspill {k['sspill']:>3} "
              f"lds {k['lds']:>6} scratch {k['scratch']:>5}  {dn[:200]}")


if __name__ == "__main__":
    main()
