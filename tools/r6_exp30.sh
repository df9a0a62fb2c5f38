#!/bin/bash
# r6 experiment 30: bn_head / bn_pool / stem weight-gradient kernels with a ReLU instantiation (no per-element switch) on top of the per-activation norm kernels: tests, cfg 2 in-tree vs ab/base.so (HEAD)
O=gpurun_out/r6_exp30.log
: > $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2 >> $O
for rep in 1 2 3; do
  for lib in "$PWD/ab/base.so" ""; do
    MI355SEG_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg --no-workloads --no-prof 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py cfg2 lib=${lib:-in-tree}', round(r['ms_per_step'], 3), 'ms/step')" >> $O
  done
done
MI355SEG_LIB_PATH=$PWD/ab/base.so python tools/bench_model.py unet 2 1 128 128 128 --steps 10 2>&1 | grep "ms/step\|norm_act\|stem_head" >> $O
python tools/bench_model.py unet 2 1 128 128 128 --steps 10 2>&1 | grep "ms/step\|norm_act\|stem_head" >> $O
cat $O
