#!/bin/bash
# r6 experiment 3: conv_b16s -- restructured epilogue (in-tree) and three workgroups per CU for k3 (ab/occ3.so) against ab/base.so (r5)
O=gpurun_out/r6_exp3.log
: > $O
python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3 >> $O
SH=("1 160 192 160 32 32 3" "1 80 96 80 64 64 3" "1 40 48 40 128 128 3" "2 128 128 128 32 32 5" "2 64 64 64 64 64 5" "1 96 96 96 32 32 3")
echo "== new (in-tree) vs base" >> $O
python tools/_ab.py $PWD/ab/base.so --dtype bf16 --what fwd,dgrad -- "${SH[@]}" >> $O 2>&1
echo "== occ3 vs base" >> $O
MI355SEG_LIB_PATH=$PWD/ab/occ3.so python tools/_ab.py $PWD/ab/base.so --dtype bf16 --what fwd,dgrad -- "${SH[@]:0:3}" "${SH[@]:5:1}" >> $O 2>&1
cat $O
