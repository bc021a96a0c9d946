"""Turns gpurun_out/refresh/ (tools/refresh_profiles.sh on the GPU box) into the committed files under profiles/."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
sys.path.insert(0, ROOT)
import bench as bench_module  # noqa: E402  (kernel_source_hash: the traffic is keyed to the kernel sources it was measured on)


def is_p6(name):
    """the headline kernel: sumfactFastKernel<Diffusion3D, 6, 7, ...> (the bench line also runs the order-4 instance)"""
    return "sumfactFastKernel" in name and "Diffusion3D, 6, 7" in name


def last_json_line(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


bench = last_json_line(os.path.join(SRC, "bench.json"))
json.dump(bench, open(os.path.join(DST, f"{tag}_bench.json"), "w"), indent=1)
json.dump(last_json_line(os.path.join(SRC, "bench_order4.json")), open(os.path.join(DST, f"{tag}_bench_order4.json"), "w"), indent=1)
with open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline  (the default run: 15 warm-up + 20 timed"
              " applies; MI355X, 64^3 order-6 Diffusion3D apply)\n")
    trace = [r for r in csv.DictReader(open(os.path.join(SRC, "stats", "stats_kernel_trace.csv")))
             if is_p6(r["Kernel_Name"])]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in trace]
    W = 15
    timed = sorted(dur[W:W + 20])
    out.write(f"# order-6 sumfactFastKernel launches in order (ms): {' '.join(f'{d:.3f}' for d in dur)}; the 20 timed ones (after {W} warm-up "
              f"launches): average {sum(timed) / max(1, len(timed)):.3f} ms, median {timed[len(timed) // 2] if timed else 0:.3f} ms (the Average column "
              f"below includes the warm-up launches, during which the clock of the idle GPU ramps)\n")
    for row in csv.reader(open(os.path.join(SRC, "stats", "stats_kernel_stats.csv"))):
        out.write(",".join('"' + c[:90] + '"' if i == 0 and row[0] != "Name" else c for i, c in enumerate(row)) + "\n")


def counter_mean(name, counter):
    vals = []
    for row in csv.DictReader(open(os.path.join(SRC, name, f"{name}_counter_collection.csv"))):
        if is_p6(row["Kernel_Name"]) and row["Counter_Name"] == counter:
            vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch, nf = counter_mean("fetch", "FETCH_SIZE")
write, nw = counter_mean("write", "WRITE_SIZE")
traffic = (2.0 * fetch + write) * 1024.0  # MI355X_MICROARCH.md: counters in KiB; gfx950 fetch counts 64 B per 128-B read
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
           "kernel": "sumfactFastKernel<Diffusion3D,6,7>", "workload": "64x64x64 order 6", "launches_averaged": [nf, nw],
           "kernel_source_sha256": bench_module.kernel_source_hash(),
           "FETCH_SIZE_per_launch": fetch, "WRITE_SIZE_per_launch": write, "unit": "KiB (counter units)",
           "correction": "fetch x2 (gfx950 wide-read under-count), write x1", "traffic_bytes_per_launch": traffic,
           "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]},
          open(os.path.join(DST, f"{tag}_hbm_traffic.json"), "w"), indent=1)
import glob
acc = {}
for f in glob.glob(os.path.join(SRC, "sq*", "sq*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if is_p6(row["Kernel_Name"]):
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
if acc:
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    with open(os.path.join(DST, f"{tag}_pmc_fast_kernel_64cubed.txt"), "w") as out:
        out.write("# rocprofv3 --pmc <SQ set> --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (two passes), sumfactFastKernel<Diffusion3D,6,7>,\n"
                  "# 64^3 elements, mean per launch, chip-wide sums\n")
        cyc = m.get("SQ_BUSY_CU_CYCLES", 0.0) / 256
        if cyc:
            out.write(f"#   cycles per launch (SQ_BUSY_CU_CYCLES / 256 CUs)              = {cyc:.4g}\n")
            if "SQ_ACTIVE_INST_VALU" in m:
                out.write(f"#   VALU busy = 4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)        = {100 * 4 * m['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.1f} %\n")
            if "SQ_WAVE_CYCLES" in m:
                out.write(f"#   resident waves per CU = 4*SQ_WAVE_CYCLES / (256 * cycles)        = {4 * m['SQ_WAVE_CYCLES'] / (256 * cyc):.2f}\n")
        if "SQ_LDS_IDX_ACTIVE" in m and cyc:
            out.write(f"#   LDS pipe active = SQ_LDS_IDX_ACTIVE / (256 * cycles)             = {100 * m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.1f} %, "
                      f"of which bank conflicts {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.1f} %\n")
        if "SQ_INSTS_VALU" in m:
            out.write(f"#   instructions per element: VALU {m['SQ_INSTS_VALU'] / 262144:.0f}, LDS {m.get('SQ_INSTS_LDS', 0) / 262144:.0f}, SALU "
                      f"{m.get('SQ_INSTS_SALU', 0) / 262144:.0f}, SMEM {m.get('SQ_INSTS_SMEM', 0) / 262144:.0f}\n")
        for k in sorted(m):
            out.write(f"{k:28s} n={len(acc[k])} mean={m[k]:.4g}\n")


# ---- round 3: counters of the order-4 instance, memory-side requests, stored assembly, config 5
def is_p4(name):
    return "sumfactFastKernel" in name and "Diffusion3D, 4, 5" in name


def counters_of(dirs, pred):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(SRC, d, f"{d}_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if pred(row["Kernel_Name"]):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def write_counters(path, header, m, n, n_elems):
    with open(path, "w") as out:
        out.write(header)
        cyc = m.get("SQ_BUSY_CU_CYCLES", 0.0) / 256
        if cyc:
            out.write(f"#   cycles per launch (SQ_BUSY_CU_CYCLES / 256 CUs)              = {cyc:.4g}\n")
            if "SQ_ACTIVE_INST_VALU" in m:
                out.write(f"#   VALU busy = 4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)        = {100 * 4 * m['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.1f} %\n")
            if "SQ_WAVE_CYCLES" in m:
                out.write(f"#   resident waves per CU = 4*SQ_WAVE_CYCLES / (256 * cycles)        = {4 * m['SQ_WAVE_CYCLES'] / (256 * cyc):.2f}\n")
            if "SQ_LDS_IDX_ACTIVE" in m:
                out.write(f"#   LDS pipe active = SQ_LDS_IDX_ACTIVE / (256 * cycles)             = {100 * m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.1f} %, "
                          f"of which bank conflicts {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.1f} %\n")
        if "SQ_INSTS_VALU" in m and n_elems:
            out.write(f"#   instructions per element: VALU {m['SQ_INSTS_VALU'] / n_elems:.0f}, LDS {m.get('SQ_INSTS_LDS', 0) / n_elems:.0f}, SALU "
                      f"{m.get('SQ_INSTS_SALU', 0) / n_elems:.0f}, SMEM {m.get('SQ_INSTS_SMEM', 0) / n_elems:.0f}, vector memory "
                      f"{(m.get('SQ_INSTS_VMEM_WR', 0) + m.get('SQ_INSTS_VMEM_RD', 0)) / n_elems:.1f}\n")
        if "TCC_EA0_WRREQ" in m and n_elems:
            out.write(f"#   memory-side requests per element: write requests {m['TCC_EA0_WRREQ'] / n_elems:.1f} (64-byte ones "
                      f"{m.get('TCC_EA0_WRREQ_64B', 0) / n_elems:.1f}), of them atomics to DRAM {m.get('TCC_EA0_WRREQ_ATOMIC_DRAM', 0) / n_elems:.1f}; "
                      f"read requests {m.get('TCC_EA0_RDREQ', 0) / n_elems:.1f}\n")
        for k in sorted(m):
            out.write(f"{k:28s} n={n[k]} mean={m[k]:.4g}\n")


if os.path.isdir(os.path.join(SRC, "o4_1")):
    m4, n4 = counters_of(["o4_1", "o4_2", "o4_3"], is_p4)
    write_counters(os.path.join(DST, f"{tag}_pmc_fast_kernel_order4.txt"),
                   "# rocprofv3 --pmc <set> --kernel-trace -- python bench.py --order 4 --steps 3 --warmup 1 --no-cpu-baseline (three passes: two SQ sets, one TCC set),\n"
                   "# sumfactFastKernel<Diffusion3D,4,5>, 64^3 elements (two per wave), mean per launch, chip-wide sums\n", m4, n4, 262144)
if os.path.isdir(os.path.join(SRC, "o6_tcc")):
    m6, n6 = counters_of(["o6_tcc"], is_p6)
    write_counters(os.path.join(DST, f"{tag}_tcc_fast_kernel_64cubed.txt"),
                   "# rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_ATOMIC_DRAM TCC_EA0_RDREQ --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline,\n"
                   "# sumfactFastKernel<Diffusion3D,6,7>, 64^3 elements, mean per launch, chip-wide sums\n", m6, n6, 262144)
if os.path.isdir(os.path.join(SRC, "asm_fetch")):  # round 4: the stored route is three kernels (tools/r04_stored_assembly.py runs all routes)
    alg = 64 * 1372 * 1372 * 8
    with open(os.path.join(DST, f"{tag}_tcc_assembly_stored.txt"), "w") as out:
        out.write("# rocprofv3 --pmc <TCC set | WRITE_SIZE | FETCH_SIZE> --kernel-trace -- python tools/r04_stored_assembly.py --orders 6 --batch 64 --steps 2 --routes direct_store,x_tiled_one_pass_symmetric\n"
                  "# 64 element matrices of 1372 x 1372 doubles (15.06 MB each, 963.7 MB per launch), row-major, per KERNEL of the two routes; mean per launch\n"
                  "# (direct store = assembleSumfactKernel<..., 0, 0> with K; default = assembleSumfactKernel<..., 2, 0> (x-major tiled layout, lower triangle only) + tiledXToRowMajorSymKernel)\n")
        for label, pred, n_mat in (("direct row-major store (assembleSumfactKernel<.., 0, 0>), 64 matrices per launch", lambda nme: "assembleSumfactKernel" in nme and ", 0, 0>" in nme, 64),
                                   ("x-major tiled store of the lower triangle (assembleSumfactKernel<.., 2, 0>), 64 matrices per launch", lambda nme: "assembleSumfactKernel" in nme and ", 2, 0>" in nme, 64),
                                   ("tiledXToRowMajorSymKernel (reads the lower triangle once, writes every entry and its mirror image), 64 matrices per launch", lambda nme: "tiledXToRowMajorSymKernel" in nme, 64)):
            mk, nk = counters_of(["asm_tcc", "asm_write", "asm_fetch"], pred)
            if not mk:
                continue
            alg = n_mat * 1372 * 1372 * 8
            out.write(f"## {label}\n")
            if "WRITE_SIZE" in mk:
                out.write(f"#   WRITE_SIZE = {mk['WRITE_SIZE'] * 1024 / 1e6:.1f} MB per launch = x{mk['WRITE_SIZE'] * 1024 / alg:.2f} of the matrices' bytes\n")
            if "FETCH_SIZE" in mk:
                out.write(f"#   2 x FETCH_SIZE = {2 * mk['FETCH_SIZE'] * 1024 / 1e6:.1f} MB per launch = x{2 * mk['FETCH_SIZE'] * 1024 / alg:.2f}\n")
            if "TCC_EA0_WRREQ" in mk:
                out.write(f"#   write requests to memory: {mk['TCC_EA0_WRREQ']:.4g} per launch = {mk['TCC_EA0_WRREQ'] * 64 / alg:.2f} x (matrix bytes / 64); "
                          f"64-byte ones: {mk.get('TCC_EA0_WRREQ_64B', 0):.4g} ({100 * mk.get('TCC_EA0_WRREQ_64B', 0) / mk['TCC_EA0_WRREQ']:.0f} %)\n")
            for k in sorted(mk):
                out.write(f"{k:28s} n={nk[k]} mean={mk[k]:.4g}\n")
elif os.path.isdir(os.path.join(SRC, "asm_tcc")):
    ma, na = counters_of(["asm_tcc", "asm_write"], lambda name: "assembleSumfactKernel" in name)
    with open(os.path.join(DST, f"{tag}_tcc_assembly_stored.txt"), "w") as out:
        out.write("# rocprofv3 --pmc <TCC set | WRITE_SIZE> --kernel-trace -- python tools/bench_assembly.py --order 6 --store --batch 64: assembleSumfactKernel<Diffusion3D,6,7>\n"
                  "# writing 64 element matrices of 1372 x 1372 doubles (15.06 MB each, 963.7 MB per launch) row-major; mean per launch\n")
        alg = 64 * 1372 * 1372 * 8
        if "WRITE_SIZE" in ma:
            out.write(f"#   WRITE_SIZE = {ma['WRITE_SIZE'] * 1024 / 1e6:.1f} MB per launch vs {alg / 1e6:.1f} MB algorithmic = x{ma['WRITE_SIZE'] * 1024 / alg:.2f}\n")
        if "TCC_EA0_WRREQ" in ma:
            out.write(f"#   write requests to memory: {ma['TCC_EA0_WRREQ']:.4g} per launch = {ma['TCC_EA0_WRREQ'] * 64 / alg:.2f} x (algorithmic bytes / 64); "
                      f"64-byte ones: {ma.get('TCC_EA0_WRREQ_64B', 0):.4g} ({100 * ma.get('TCC_EA0_WRREQ_64B', 0) / ma['TCC_EA0_WRREQ']:.0f} %)\n")
        for k in sorted(ma):
            out.write(f"{k:28s} n={na[k]} mean={ma[k]:.4g}\n")
if os.path.exists(os.path.join(SRC, "config5.json")):
    with open(os.path.join(DST, f"{tag}_config5_pcg.jsonl"), "a") as out:
        out.write(json.dumps(last_json_line(os.path.join(SRC, "config5.json"))) + "\n")
print(json.dumps({"value": bench["value"], "kernel_ms": bench["roofline"]["kernel_ms"], "frac": bench["roofline"]["frac"],
                  "traffic_GB": traffic / 1e9}))
