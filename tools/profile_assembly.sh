#!/bin/bash
# rocprofv3 summaries of the LocalAssembly kernels (tools/bench_assembly.py, order 6, batch 512): kernel stats + MFMA counters
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/asm_prof
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python tools/bench_assembly.py --batch 512 --steps 3 > "$OUT/stats.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc" -o pmc -- python tools/bench_assembly.py --batch 512 --steps 3 > "$OUT/pmc.log" 2>&1 || { tail -5 "$OUT/pmc.log"; exit 1; }
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open("$OUT/pmc/pmc_counter_collection.csv")):
    if "assembleGemmKernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(f"{k:32s} n={len(acc[k])} mean={sum(acc[k])/len(acc[k]):.5g}")
for row in csv.reader(open("$OUT/stats/stats_kernel_stats.csv")):
    if row and ("assemble" in row[0] or row[0] == "Name"):
        print(",".join(c[:70] for c in row))
PY
