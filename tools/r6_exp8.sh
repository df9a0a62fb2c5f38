#!/bin/bash
# r6 experiment 8: ticket finalise in norm / bn_head kernels (tests), fma_mix split in conv_x3s + scalar split in the weight gradient (A/B vs ab/base.so)
O=gpurun_out/r6_exp8.log
: > $O
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -4 >> $O
python -m pytest tests/test_gpu_unet.py -x -q 2>&1 | tail -4 >> $O
echo "== A/B (old = ab/base.so)" >> $O
python tools/_ab.py $PWD/ab/base.so --what fwd,dgrad,dgbn,wgrad --math f16x3 -- "2 128 128 128 32 32 3" "2 128 128 128 64 32 3" "2 64 64 64 64 64 3" "2 64 64 64 128 64 3" "2 32 32 32 128 128 3" "2 32 32 32 256 128 3" >> $O 2>&1
cat $O
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_exp8_bench.json 2> gpurun_out/r6_exp8_bench.err
python tools/_print_bench.py gpurun_out/r6_exp8_bench.json
