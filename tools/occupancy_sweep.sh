for w in 4 5 6 7 8; do
  echo "waves_per_cu=$w"
  L3K_FAST_WAVES_PER_CU=$w timeout -k 10 200 python tools/kbench.py --order 5 --ne 36 --flags 0,3 || exit 1
done
for w in 4 6 7; do
  echo "p6 waves_per_cu=$w"
  L3K_FAST_WAVES_PER_CU=$w timeout -k 10 200 python tools/kbench.py --order 6 --ne 32 --flags 0,3 || exit 1
done
