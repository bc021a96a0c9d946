"""Scan a built library for hazards of the hand-written DPP instructions (l3ster_amd/isa_check.py has the rules).
    python tools/check_dpp_hazards.py [l3ster_amd/lib/libl3k.so]
Exit code 1 and a listing if a hazard is found.  (Also run by tests/test_cabi_cpu.py on the CPU and by the plugin build.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l3ster_amd.isa_check import check_dpp_hazards as check  # noqa: E402

if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "l3ster_amd", "lib", "libl3k.so")
    n, bad = check(lib)
    print(f"{n} DPP row_newbcast instructions in {lib}: {len(bad)} hazards")
    for func, ins, why in bad[:40]:
        print(f"  {func[:90]}: {ins}   <- {why}")
    sys.exit(1 if bad else 0)
