#!/bin/bash
# r6 experiment 32: the wide weight-gradient kernel wherever the geometry allows (MI355SEG_WGRAD_WIDE=2) vs where the plan rule puts it (1), cfg 2 and the bf16 legs, same box
O=gpurun_out/r6_exp32.log
: > $O
for rep in 1 2; do
  for m in 1 2; do
    echo "== MI355SEG_WGRAD_WIDE=$m" >> $O
    MI355SEG_WGRAD_WIDE=$m python tools/bench_model.py unet 2 1 128 128 128 --steps 10 2>&1 | grep "ms/step\|wgrad" >> $O
    MI355SEG_WGRAD_WIDE=$m python tools/bench_model.py vnet 2 1 128 128 128 --dtype bf16 --steps 10 2>&1 | grep "ms/step\|wgrad" >> $O
    MI355SEG_WGRAD_WIDE=$m python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 10 2>&1 | grep "ms/step\|wgrad" >> $O
  done
done
cat $O
