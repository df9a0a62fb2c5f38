#!/bin/bash
# A/B of an environment knob on the three bf16 legs and cfg 2 (fp32): tools/_ab_legs.sh VAR "v1 v2 ..."
VAR=$1
for v in $2; do
  export $VAR=$v
  echo "$VAR=$v"
  for m in "res_unet 1 4 160 192 160 --classes 4 --dtype bf16" "vnet 2 1 128 128 128 --dtype bf16" "unetr 1 1 96 96 96 --dtype bf16" "unet 2 1 128 128 128"; do
    python tools/bench_model.py $m --steps 10 --no-prof 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['model'], round(d['ms_per_step'],3))"
  done
done
