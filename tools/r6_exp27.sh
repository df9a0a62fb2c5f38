#!/bin/bash
# r6 experiment 27: conv_b16s<3, 4> at three workgroups per CU (ab/occ3.so: 168 VGPRs, 60 B of scratch; <3, 2> is already there at 160): 'old' = occ3, 'new' = in-tree (x < 1: occ3 is faster)
O=gpurun_out/r6_exp27.log
: > $O
python tools/_ab.py $PWD/ab/occ3.so --dtype bf16 --what fwd,dgrad -- "1 160 192 160 32 32 3" "1 160 192 160 64 32 3" "1 96 96 96 32 32 3" "1 80 96 80 64 32 3" "1 160 192 160 32 64 3" >> $O 2>&1
cat $O
