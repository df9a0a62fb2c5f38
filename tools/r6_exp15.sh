#!/bin/bash
# r6 experiment 15: one-launch packings + no separate weight-maximum launch while a plan is replayed: full GPU suite, then cfg 2 / UNETR with / without
O=gpurun_out/r6_exp15.log
: > $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $O || { cat $O; exit 1; }
for rep in 1 2 3; do
for np in 0 1; do
  echo "== MI355SEG_NO_PREPACK=$np" >> $O
  export MI355SEG_NO_PREPACK=$np; [ $np = 0 ] && unset MI355SEG_NO_PREPACK
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py cfg2', round(r['ms_per_step'], 3), 'ms/step')" >> $O
  python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
done
done
cat $O
