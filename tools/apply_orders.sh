#!/bin/bash
# The matrix-free Diffusion3D apply over the orders: python bench.py --order N --ne 64 (48 for orders 7, 8), short runs.
#   bash tools/apply_orders.sh "1 2 3 4 5 6 7 8" > profiles/rNN_apply_orders.log
export TMPDIR=/tmp
for o in ${1:-1 2 3 4 5 6 7 8}; do
  ne=64; [ "$o" -ge 7 ] && ne=48
  timeout -k 10 300 python bench.py --order $o --ne $ne --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null > /tmp/_ao.json || exit 1
  python3 - "$o" "$ne" <<'PY'
import json, sys
d = json.loads(open("/tmp/_ao.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"order {sys.argv[1]} ne {sys.argv[2]}: {d['value']:.3e} dof/s frac {r['frac']:.3f} kernel ms {r['kernel_ms']:.3f}  {r.get('kernel', '')[:70]}", flush=True)
PY
done
