#!/bin/bash
# A/B of two builds of the library on the same box: kbench at 64^3, orders 6 and 4, alternating
for rep in 1 2; do
for o in 6 4; do
  timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --child 2>&1 | grep flags | sed "s/^/A (default)  /"
  L3K_LIB=$PWD/l3ster_amd/lib/libl3k_$1.so timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --child 2>&1 | grep flags | sed "s/^/B ($1)  /"
done
done
