#!/bin/bash
# same-box comparison of library builds: kbench at 64^3; usage: r02_abc.sh variant1 variant2 ...  (ORDERS="6 4")
for rep in 1 2; do
for o in ${ORDERS:-6}; do
  timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --child 2>&1 | grep flags | sed "s/^/default   /"
  for v in "$@"; do
    L3K_LIB=$PWD/l3ster_amd/lib/libl3k_$v.so timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --child 2>&1 | grep flags | sed "s/^/$v   /"
  done
done
done
