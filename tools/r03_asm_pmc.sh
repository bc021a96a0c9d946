#!/bin/bash
# PMC counters of the two streaming assembly kernels of round 3 (order 6, batch 2048, checksum mode): diagonal and off-diagonal blocks
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/asm_pmc
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p$i -- python tools/bench_assembly.py --batch 2048 --steps 3 > "$OUT/p$i.log" 2>&1 || { tail -5 "$OUT/p$i.log"; exit 1; }
done
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/p*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "assembleSumfactKernel" in n:
            kind = "diagonal blocks (BLOCKS 1)" if ", 1>" in n else ("off-diagonal blocks (BLOCKS 2)" if ", 2>" in n else "other")
            acc[kind][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kind, cs in sorted(acc.items()):
    m = {k: sum(v)/len(v) for k, v in cs.items()}
    cyc = m["SQ_BUSY_CU_CYCLES"]/256
    print(f"== {kind}: mean per launch of {len(cs['SQ_BUSY_CU_CYCLES'])} launches (2048 elements each)")
    print("   cycles/launch %.4g  VALU busy %.1f %%  waves/CU %.2f  LDS pipe active %.1f %% (conflicts %.1f %% of it)  waiting %.1f %%" % (
        cyc, 100*4*m["SQ_ACTIVE_INST_VALU"]/(1024*cyc), 4*m["SQ_WAVE_CYCLES"]/(256*cyc), 100*m["SQ_LDS_IDX_ACTIVE"]/(256*cyc),
        100*m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"], 100*m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"]))
    print("   per element: VALU %.0f  SALU %.0f  LDS %.0f  SMEM %.0f instructions" % (m["SQ_INSTS_VALU"]/2048, m["SQ_INSTS_SALU"]/2048, m["SQ_INSTS_LDS"]/2048, m["SQ_INSTS_SMEM"]/2048))
PY
