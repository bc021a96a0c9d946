#!/bin/bash
# A/B on one box: <p, A p> accumulated inside the quadrature stage (libl3k_enqp.so) vs as x_e . y_e at the I^T z stage (libl3k_enend.so):
# config 5 (AdvDiff3D order 4) and the order-6 Diffusion3D solve at 48^3, twice each, alternating
export TMPDIR=/tmp
for rep in 1 2; do
  for v in enqp enend; do
    echo "== $v config5 (rep $rep)"; L3K_LIB=$PWD/l3ster_amd/lib/libl3k_$v.so python tools/bench_config5.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['iterations'], round(d['solve_s'],3), 's  apply alone', round(d['ms_per_apply_alone'],4), 'ms')"
    echo "== $v order 6 48^3 (rep $rep)"; L3K_LIB=$PWD/l3ster_amd/lib/libl3k_$v.so python tools/diffusion3d_benchmark.py --ne 48 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['iterations'], round(d['solve_s'],3), 's')"
  done
done
