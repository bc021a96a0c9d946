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
_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")


class IsaScanError(RuntimeError):
    """The scan could not be carried out (no gfx950 code object found, a compressed bundle, llvm-objdump missing or failing)."""


def _regs(tok):
    m = _REG.fullmatch(tok.strip().lstrip("-|").rstrip("|,"))
    if not m:
        return None
    if m.group(3) is not None:
        return int(m.group(3)), int(m.group(3))
    return int(m.group(1)), int(m.group(2))


def _disassemble(co):
    if not os.path.exists(OBJDUMP):
        raise IsaScanError(f"{OBJDUMP} not found")
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(co)
    try:
        r = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True)
    finally:
        os.unlink(f.name)
    if r.returncode != 0:
        raise IsaScanError(f"llvm-objdump failed ({r.returncode}): {r.stderr[-500:]}")
    return r.stdout


def _functions(text):
    """-> {function: [(address, mnemonic, operands, written vgpr range or None, wait states, text), ...]}"""
    funcs, cur = {}, None
    for line in text.splitlines():
        if line.endswith(">:"):
            cur = funcs.setdefault(line.split("<", 1)[1][:-2], [])
            continue
        if cur is None:
            continue
        m = _ADDR.search(line)
        body = line.split("//")[0].strip()
        if not body or body.startswith(("/", ".")) or ":" in body.split()[0]:
            continue
        parts = body.split(None, 1)
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        wr = _regs(ops.split(",")[0].split()[0]) if mn.startswith("v_") and ops else None
        st = 1
        if mn == "s_nop":
            try:
                st = int(ops.strip(), 0) + 1
            except ValueError:
                st = 1
        cur.append((int(m.group(1), 16) if m else -1, mn, ops, wr, st, body))
    return funcs


def _is_dpp(ins):
    return "_dpp" in ins[1] and "row_newbcast" in ins[2]


def _hazards_before(ins, preceding, func, via=""):
    """Hazards of the DPP instruction `ins` against the instructions `preceding` (program order, nearest last)."""
    bad = []
    toks = ins[2].split(",")
    src0 = _regs(toks[1].split()[0]) if len(toks) > 1 else None
    states = 0
    for _, pmn, _, pw, pstates, _ in reversed(preceding):
        if pmn.startswith("v_") and src0 and pw and not (pw[1] < src0[0] or pw[0] > src0[1]) and states < 2:
            bad.append((func, ins[5], f"{pmn} writes the DPP source {states} wait state(s) earlier{via}"))
        if pmn.startswith("v_cmpx") and states < 5:
            bad.append((func, ins[5], f"{pmn} writes EXEC {states} wait state(s) earlier{via}"))
        states += pstates
        if states >= 5:
            break
    return bad


def scan(path):
    """-> dict(n_code_objects, n_dpp, hazards=[(function, instruction, reason), ...]).  Straight-line order AND every branch edge:
    the instructions in front of a taken s_branch / s_cbranch_* precede the instructions at its target (a VALU write of a DPP source at
    a loop's tail against a DPP read at its head)."""
    data = open(path, "rb").read()
    cos = list(code_objects(path))
    if not cos:
        why = "a compressed offload bundle (CCOB): not supported by this scan" if b"CCOB" in data else "no __CLANG_OFFLOAD_BUNDLE__ with a gfx950 entry"
        raise IsaScanError(f"{os.path.basename(path)}: no gfx950 code object found ({why})")
    n_dpp, bad = 0, []
    for co in cos:
        for func, ins in _functions(_disassemble(co)).items():
            index_of = {a: i for i, (a, *_rest) in enumerate(ins) if a >= 0}
            for i, x in enumerate(ins):
                if _is_dpp(x):
                    n_dpp += 1
                    bad += _hazards_before(x, ins[max(0, i - 8):i], func)
                if x[1] == "s_branch" or x[1].startswith("s_cbranch"):
                    try:
                        off = int(x[2].split()[0], 0)
                    except (ValueError, IndexError):
                        continue
                    off = off - 0x10000 if off >= 0x8000 else off  # simm16, in dwords, relative to the next instruction
                    j = index_of.get(x[0] + 4 + 4 * off)
                    if j is None:
                        continue
                    pred, states = ins[max(0, i - 7):i + 1], 0
                    for k in range(j, min(j + 8, len(ins))):  # the DPP reads within 5 wait states of the target
                        if _is_dpp(ins[k]):
                            bad += _hazards_before(ins[k], pred + ins[j:k], func, via=f" (across the branch at {x[0]:#x})")
                        states += ins[k][4]
                        if states >= 5:
                            break
    return dict(n_code_objects=len(cos), n_dpp=n_dpp, hazards=bad)


def check_dpp_hazards(path):
    """-> (number of DPP row_newbcast instructions, [(function, instruction, reason), ...]); raises IsaScanError if the library
    could not be scanned."""
    r = scan(path)
    return r["n_dpp"], r["hazards"]
