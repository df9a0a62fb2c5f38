#!/bin/bash
# r6 experiment 31: conv_x3s without its cold epilogue paths (ab/slim.so: no activation store loop, no generic act_grad reduce: 16.8 k -> 5.1 k instructions): 'old' = slim, 'new' = in-tree
O=gpurun_out/r6_exp31.log
: > $O
python tools/_ab.py $PWD/ab/slim.so --math f16x3 --what fwd,dgrad,dgbn -- "2 128 128 128 32 32 3" "2 128 128 128 64 32 3" "2 128 128 128 32 64 3" "2 64 64 64 64 64 3" "2 64 64 64 128 64 3" "2 32 32 32 128 128 3" >> $O 2>&1
for rep in 1 2 3; do
  for lib in "$PWD/ab/slim.so" ""; do
    MI355SEG_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py cfg2 lib=${lib:-in-tree}', round(r['ms_per_step'], 3), 'ms/step')" >> $O
  done
done
cat $O
