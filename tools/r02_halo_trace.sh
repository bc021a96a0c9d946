#!/bin/bash
# kernel / copy timeline of the partitioned apply with the library's RCCL exchange (self exchange on one GPU)
set -o pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r02_halo_trace
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python tests/rccl_native_halo_periodic.py --ne ${NE:-32 32 32} --order ${ORDER:-6} --bench 5 > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
tail -3 $OUT/run.log
find $OUT -name "*.csv" | head
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:70], r.get("Stream_Id", "")))
for f in glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "copy") + " " + r.get("Bytes", ""), r.get("Stream_Id", "")))
rows.sort()
# the last apply: from the last scaleKernel-ish start; print the last 40 events relative to the last-but-one 'dirichletRows'
t0 = rows[-45][0] if len(rows) > 45 else rows[0][0]
for s, e, kind, name, stream in rows[-45:]:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  {kind} stream {stream:>3}  {name}")
PY
