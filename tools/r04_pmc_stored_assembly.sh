#!/bin/bash
# SQ counters of the two kernels of the stored row-major LocalAssembly route (x-major tiled assembly of the lower triangle, mirroring
# transposition), order 6, 202 matrices in sub-batches of 101 -> profiles/r04_pmc_stored_assembly.txt
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_asm
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/s$i" -o s$i -- python tools/r04_stored_assembly.py --orders 6 --batch 202 --steps 2 --routes x_tiled_one_pass_symmetric > "$OUT/s$i.log" 2>&1 || exit 1
done
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_asm/s*/*counter_collection.csv"):
    by=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        key = "asm2" if ("assembleSumfactKernel" in n and ", 2, 0>" in n) else ("trX" if "tiledXToRowMajorSym" in n else None)
        if key: by[(key,r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
    for (key,_),d in by.items():
        for c,v in d.items(): acc[key][c].append(v)
for key in acc:
    print("##",key)
    m={c:sum(v)/len(v) for c,v in acc[key].items()}
    for c in sorted(m): print(f"{c:28s} n={len(acc[key][c])} mean={m[c]:.4g}")
    if "SQ_BUSY_CU_CYCLES" in m:
        cyc=m["SQ_BUSY_CU_CYCLES"]/256
        print(f"# cycles per launch {cyc:.4g}; VALU busy {4*m['SQ_ACTIVE_INST_VALU']/(1024*cyc)*100:.1f} %; waves/CU {4*m['SQ_WAVE_CYCLES']/(256*cyc):.2f}; LDS active (inst) {4*m['SQ_ACTIVE_INST_LDS']/(1024*cyc)*100:.1f} %; wait any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']*100:.1f} % of wave cycles; wait inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']*100:.1f} %")
PY
