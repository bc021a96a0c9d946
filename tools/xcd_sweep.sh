for ne in 48 64; do
  for x in 0 1; do
    if [ $x = 1 ]; then export L3K_FAST_XCD=1; else unset L3K_FAST_XCD; fi
    echo "== p4 ne=$ne xcd=$x"; timeout -k 10 200 python tools/kbench.py --order 4 --ne $ne --steps 8 --flags 0 --child || exit 1
  done
  echo "== p6 ne=$ne xcd default(on) / off"
  timeout -k 10 200 python tools/kbench.py --order 6 --ne $ne --steps 8 --flags 0 --child
  L3K_FAST_NO_XCD=1 timeout -k 10 200 python tools/kbench.py --order 6 --ne $ne --steps 8 --flags 0 --child
done
