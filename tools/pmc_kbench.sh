#!/bin/bash
# PMC passes over the element kernel alone (tools/kbench.py), summarised per dispatch.  Usage: pmc_kbench.sh <order> <ne> <flags>
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf "$OUT" && mkdir -p "$OUT"
export L3K_DEBUG_FLAGS=$3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python tools/kbench.py --order $1 --ne $2 --steps 5 --child > "$OUT/p$i.log" 2>&1 || { tail -5 "$OUT/p$i.log"; exit 1; }
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "sumfactFastKernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
