#!/bin/bash
# r6 experiment 24: conv_b16s persistent tile walk (k3, >= 1536 tiles): tests, layers and legs, in-tree ('new') vs ab/base.so ('old' = HEAD)
O=gpurun_out/r6_exp24.log
: > $O
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_models.py -x -q 2>&1 | tail -2 >> $O
SH=("1 160 192 160 32 32 3" "1 160 192 160 64 32 3" "1 80 96 80 64 64 3" "1 80 96 80 128 64 3" "1 40 48 40 128 128 3" "1 96 96 96 32 32 3" "1 48 48 48 128 64 3" "1 96 96 96 64 64 3" "1 160 192 160 32 64 3")
python tools/_ab.py $PWD/ab/base.so --dtype bf16 --what fwd,dgrad -- "${SH[@]}" >> $O 2>&1
for rep in 1 2; do
  for lib in "$PWD/ab/base.so" ""; do
    echo "== lib=${lib:-in-tree}" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
    MI355SEG_LIB_PATH=$lib python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step" >> $O
  done
done
cat $O
