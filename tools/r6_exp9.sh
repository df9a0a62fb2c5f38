#!/bin/bash
# r6 experiment 9: UNETR token path -- one-launch Linear backward, paired attention-backward GEMMs, merged LayerNorm backward, q/k/v seated in one buffer
O=gpurun_out/r6_exp9.log
: > $O
python -m pytest tests/test_gpu_ops.py -x -q -k "linear or two_batched or split_k" 2>&1 | tail -25 >> $O
python -m pytest tests/test_gpu_models.py -x -q -k "unetr or transformer" 2>&1 | tail -4 >> $O
python -m pytest tests/test_gpu_bf16.py -x -q -k "unetr" 2>&1 | tail -4 >> $O
for v in 1 0; do
  echo "== unetr bf16 leg, MI355SEG_NO_GEMM_PAIRS=$v" >> $O
  if [ $v = 1 ]; then export MI355SEG_NO_GEMM_PAIRS=1; else unset MI355SEG_NO_GEMM_PAIRS; fi
  python tools/bench_model.py unetr 1 1 96 96 96 --dtype bf16 --steps 20 --no-prof 2>&1 | tail -3 >> $O
done
unset MI355SEG_NO_GEMM_PAIRS
bash tools/step_trace_leg.sh r6_unetr bce_bwd_kernel unetr 1 1 96 96 96 --dtype bf16 >> $O 2>&1
head -3 gpurun_out/trace_r6_unetr/last_step.txt >> $O
cat $O
