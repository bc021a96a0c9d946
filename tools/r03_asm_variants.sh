#!/bin/bash
# streaming / tiled LocalAssembly with stage-3 blocks of ZB b_z and a register budget for MINW waves per SIMD (3: two workgroups per CU at order 6)
export TMPDIR=/tmp
for v in "" _asm_1_2 _asm_2_3 _asm_2_2; do
  echo "== lib$v  (default lib: ZB 1, MINW 3; _asm_Z_W: ZB Z, MINW W)"
  L3K_LIB=$PWD/l3ster_amd/lib/libl3k$v.so python tools/r03_tiled_rate.py 2>&1 | grep "p=6\|p=4" | grep -v row-major
  for o in 6 4; do L3K_LIB=$PWD/l3ster_amd/lib/libl3k$v.so python tools/bench_assembly.py --order $o --batch 512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_assembly batch 512', d['metric'][-12:], round(d['value']))"; done
done
