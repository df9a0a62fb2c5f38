#!/bin/bash
# r6 experiment 21: store loops of conv_b16s / conv_x3s in two instantiations (activation switch out of the training path): in-tree ('new') vs ab/base.so ('old' = HEAD)
O=gpurun_out/r6_exp21.log
: > $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -2 >> $O
SH=("1 160 192 160 32 32 3" "1 160 192 160 64 32 3" "1 80 96 80 64 64 3" "1 80 96 80 128 64 3" "1 40 48 40 128 128 3" "2 128 128 128 32 32 5" "2 64 64 64 64 64 5" "1 96 96 96 32 32 3" "1 48 48 48 128 64 3")
echo "== bf16" >> $O
python tools/_ab.py $PWD/ab/base.so --dtype bf16 --what fwd,dgrad -- "${SH[@]}" >> $O 2>&1
echo "== f16x3" >> $O
python tools/_ab.py $PWD/ab/base.so --math f16x3 --what fwd,dgrad -- "2 128 128 128 32 32 3" "2 128 128 128 64 32 3" "2 128 128 128 32 64 3" "2 64 64 64 64 64 3" "2 64 64 64 128 64 3" "2 32 32 32 128 128 3" "2 16 16 16 256 256 3" >> $O 2>&1
cat $O
