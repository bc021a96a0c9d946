#!/bin/bash
# kbench over the experiment libraries lib/libl3k_<variant>.so (python -m l3ster_amd.build with L3K_VARIANT / L3K_EXTRA_FLAGS)
for lib in l3ster_amd/lib/libl3k.so "$@"; do
  echo "== $lib"
  L3K_LIB=$PWD/$lib timeout -k 10 200 python tools/kbench.py --order 6 --ne 32 --flags 0 --child || exit 1
  L3K_LIB=$PWD/$lib timeout -k 10 200 python tools/kbench.py --order 4 --ne 48 --flags 0 --child || exit 1
done
