#!/bin/bash
# the affine variant of the one-wave kernel on uniform meshes vs the general kernel on the same meshes
set -o pipefail
for o in 6 4; do
  timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --perturb 0 --child 2>&1 | grep flags | sed "s/^/affine variant   /"
  L3K_NO_AFFINE=1 timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --perturb 0 --child 2>&1 | grep flags | sed "s/^/general kernel   /"
  timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --child 2>&1 | grep flags | sed "s/^/perturbed mesh   /"
done
