"""ISA checks of a built library (used by tools/check_dpp_hazards.py, the CPU tests and the run-time plugin build).

The hand-written DPP instructions (inline asm v_fmac_f64_dpp ... row_newbcast, csrc/device/assemble.hpp) are opaque to the compiler's
hazard recognizer.  check_dpp_hazards scans the gfx950 ISA for the two hazards the ISA manual lists for DPP reads:
  * a VALU instruction that writes the DPP source register (src0) needs 2 wait states before the DPP instruction,
  * a VALU instruction that writes EXEC (v_cmpx_*) needs 5.
A wait state is one issued instruction; s_nop N counts N + 1."""
import os
import re
import struct
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
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



_REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def _regs(tok):
    m = _REG.fullmatch(tok.strip().lstrip("-|").rstrip("|,"))
    if not m:
        return None
    if m.group(3) is not None:
        return int(m.group(3)), int(m.group(3))
    return int(m.group(1)), int(m.group(2))


def check_dpp_hazards(path):
    """-> (number of DPP row_newbcast instructions, [(function, instruction, reason), ...])"""
    n_dpp, bad = 0, []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        try:
            text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        func, window = "", []  # window: the last instructions as (mnemonic, written vgpr range or None, wait states)
        for line in text.splitlines():
            if line.endswith(">:"):
                func, window = line.split("<", 1)[1][:-2], []
                continue
            body = line.split("//")[0].strip()
            if not body or body.startswith(("/", ".")) or ":" in body.split()[0]:
                continue
            parts = body.split(None, 1)
            mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
            if "_dpp" in mn and "row_newbcast" in ops:
                n_dpp += 1
                toks = ops.split(",")
                src0 = _regs(toks[1].split()[0]) if len(toks) > 1 else None
                states = 0
                for pmn, pw, pstates in reversed(window):
                    if pmn.startswith("v_") and src0 and pw and not (pw[1] < src0[0] or pw[0] > src0[1]) and states < 2:
                        bad.append((func, body, f"{pmn} writes the DPP source {states} wait state(s) earlier"))
                    if pmn.startswith("v_cmpx") and states < 5:
                        bad.append((func, body, f"{pmn} writes EXEC {states} wait state(s) earlier"))
                    states += pstates
                    if states >= 5:
                        break
            wr = None
            if mn.startswith("v_") and ops:
                wr = _regs(ops.split(",")[0].split()[0])
            st = 1
            if mn == "s_nop":
                try:
                    st = int(ops.strip(), 0) + 1
                except ValueError:
                    st = 1
            window.append((mn, wr, st))
            if len(window) > 8:
                window.pop(0)
    return n_dpp, bad
