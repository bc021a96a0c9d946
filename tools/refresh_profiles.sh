#!/bin/bash
# Runs on the GPU box (gpurun): the round's judged measurements.  Outputs under gpurun_out/refresh/.
# After a change of the kernel sources: run this, then python tools/collect_profiles.py rNN here (writes the traffic profile
# with the new source hash), then bench.py alone once more into gpurun_out/refresh/bench.json and collect again: the bench line
# of the first run was printed against the previous traffic profile and says so ("stale").
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/refresh
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
timeout -k 10 300 python bench.py --order 4 --no-cpu-baseline > "$OUT/bench_order4.json" 2>> "$OUT/bench.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python bench.py --no-cpu-baseline > "$OUT/stats.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1 || exit 1
# SQ counters of the element kernel in the same bench command (three passes)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/sq$i" -o sq$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/sq$i.log" 2>&1 || exit 1
done
find "$OUT" -name "*.csv" | head -20
cat "$OUT/bench.json"
