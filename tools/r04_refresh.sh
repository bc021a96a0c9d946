#!/bin/bash
# Round-4 judged measurements (same set as round 3's tools/r03_refresh.sh) on the GPU box: tools/refresh_profiles.sh (bench lines, kernel stats, HBM traffic and SQ counters of
# the order-6 kernel) + the order-4 kernel's counters (SQ sets, memory-side write / atomic requests), the order-6 kernel's
# memory-side requests, the stored LocalAssembly's write requests, config 5 (PCG).  Outputs under gpurun_out/refresh/.
set -o pipefail
export TMPDIR=/tmp
bash tools/refresh_profiles.sh > gpurun_out/refresh_main.log 2>&1 || { tail -5 gpurun_out/refresh_main.log; exit 1; }
OUT=$PWD/gpurun_out/refresh
TCC="TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_ATOMIC_DRAM TCC_EA0_RDREQ"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "$TCC"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/o4_$i" -o o4_$i -- python bench.py --order 4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/o4_$i.log" 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --pmc $TCC --kernel-trace --output-format csv -d "$OUT/o6_tcc" -o o6_tcc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/o6_tcc.log" 2>&1 || exit 1
# stored LocalAssembly (row-major): the kernels of the round-4 route (x-major tiled assembly of the lower triangle, mirroring transposition) beside the direct store
timeout -k 10 300 rocprofv3 --pmc $TCC --kernel-trace --output-format csv -d "$OUT/asm_tcc" -o asm_tcc -- python tools/r04_stored_assembly.py --orders 6 --batch 64 --steps 2 --routes direct_store,x_tiled_one_pass_symmetric > "$OUT/asm_tcc.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/asm_write" -o asm_write -- python tools/r04_stored_assembly.py --orders 6 --batch 64 --steps 2 --routes direct_store,x_tiled_one_pass_symmetric > "$OUT/asm_write.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/asm_fetch" -o asm_fetch -- python tools/r04_stored_assembly.py --orders 6 --batch 64 --steps 2 --routes direct_store,x_tiled_one_pass_symmetric > "$OUT/asm_fetch.log" 2>&1 || exit 1
timeout -k 10 900 python tools/bench_config5.py > "$OUT/config5.json" 2> "$OUT/config5.err" || { tail -3 "$OUT/config5.err"; exit 1; }
tail -1 "$OUT/config5.json" | cut -c1-600
tail -1 "$OUT/bench.json" | cut -c1-1500
