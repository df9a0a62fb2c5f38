#!/bin/bash
# r6 experiment 26: the cfg-2 headline step, eager launch loop vs HIP-graph replay, same command and box, alternating (no in-library bracketing in either)
O=gpurun_out/r6_exp26.log
: > $O
for rep in 1 2 3; do
  for mode in "" "--hip-graph"; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof $mode 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py cfg2 ${mode:-eager}', round(r['ms_per_step'], 3), 'ms/step')" >> $O
  done
done
MI355SEG_NO_PREPACK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof --hip-graph 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py cfg2 --hip-graph, MI355SEG_NO_PREPACK=1', round(r['ms_per_step'], 3), 'ms/step')" >> $O
MI355SEG_NO_PREPACK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py cfg2 eager, MI355SEG_NO_PREPACK=1', round(r['ms_per_step'], 3), 'ms/step')" >> $O
cat $O
