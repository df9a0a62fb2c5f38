#!/bin/bash
# r6 experiment 17: conv_b16s weight-fragment prefetch depth (in-tree WD = 2 K-steps; ab/wd4.so, ab/wd6.so): 'old' = the variant, 'new' = in-tree (x < 1: the variant is faster)
O=gpurun_out/${OUT:-r6_exp17}.log
: > $O
SH=("1 160 192 160 32 32 3" "1 160 192 160 64 32 3" "1 80 96 80 64 64 3" "1 80 96 80 128 64 3" "1 40 48 40 128 128 3" "2 128 128 128 32 32 5" "2 64 64 64 64 64 5" "1 96 96 96 32 32 3" "1 48 48 48 128 64 3")
for v in ${VARIANTS:-wd4 wd6}; do
  echo "== $v vs in-tree" >> $O
  python tools/_ab.py $PWD/ab/$v.so --dtype bf16 --what fwd,dgrad -- "${SH[@]}" >> $O 2>&1
done
cat $O
