#!/bin/bash
# the PCG with the p pass inside the element kernel's gather (default) vs L3K_PCG_UNFUSED=1 (9 vector passes), same box.
# The attempt was NOT kept: the switch exists only in a tree with tools/attempts/r03_pcg_p_pass_in_gather.patch applied (to the commit
# that added the patch: git log -- tools/attempts/r03_pcg_p_pass_in_gather.patch) and rebuilt.  On HEAD both legs would run the same code.
grep -rq L3K_PCG_UNFUSED l3ster_amd/csrc || { echo "apply tools/attempts/r03_pcg_p_pass_in_gather.patch and rebuild first: HEAD has no L3K_PCG_UNFUSED switch" >&2; exit 1; }
export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export L3K_PCG_UNFUSED=1; tag=unfused; else unset L3K_PCG_UNFUSED; tag=fused; fi
    echo -n "$tag config5: "; python tools/bench_config5.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['iterations'], 'iterations', round(d['solve_s'],3), 's', '%.3e' % d['dof_per_s_inside_solve'], 'dof/s in the solve; apply alone %.3e' % d['dof_per_s_apply_alone'], 'ratio %.3f' % (d['dof_per_s_inside_solve']/d['dof_per_s_apply_alone']))"
    echo -n "$tag Diffusion3D order 6 64^3: "; python tools/diffusion3d_benchmark.py --ne 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['iterations'], 'iterations', round(d['solve_s'],3), 's', ['%.3e' % e for e in d['l2_error_components']])"
  done
done
