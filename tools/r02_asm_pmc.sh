#!/bin/bash
# PMC counters of the sum-factorised assembly kernel (order 6, batch 512, checksum mode)
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/asm_pmc
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p$i -- python tools/bench_assembly.py --batch 512 --steps 3 > "$OUT/p$i.log" 2>&1 || { tail -5 "$OUT/p$i.log"; exit 1; }
done
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/p*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "assembleSumfactKernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in acc.items()}
for k in sorted(m): print(f"{k:28s} n={len(acc[k])} mean={m[k]:.5g}")
cyc = m["SQ_BUSY_CU_CYCLES"]/256
print("cycles/launch", cyc, "VALU busy %", 100*4*m["SQ_ACTIVE_INST_VALU"]/(1024*cyc), "waves/CU", 4*m["SQ_WAVE_CYCLES"]/(256*cyc),
      "LDS active %", 100*m["SQ_LDS_IDX_ACTIVE"]/(256*cyc), "conflict %", 100*m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"])
print("per element: VALU", m["SQ_INSTS_VALU"]/512, "SALU", m["SQ_INSTS_SALU"]/512, "LDS", m["SQ_INSTS_LDS"]/512, "SMEM", m["SQ_INSTS_SMEM"]/512)
PY
