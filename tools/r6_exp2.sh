#!/bin/bash
# r6 experiment 2: incremental halo offsets + scalarised epilogue addressing (in-tree) against ab/base.so (r5 kernels)
O=gpurun_out/r6_exp2.log
: > $O
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -5 >> $O
echo "== A/B (old = ab/base.so)" >> $O
python tools/_ab.py $PWD/ab/base.so --what fwd,dgbn,dgrad --math f16x3 -- "2 128 128 128 32 32 3" "2 128 128 128 64 32 3" "2 64 64 64 64 64 3" "2 64 64 64 128 64 3" "2 32 32 32 128 128 3" "2 16 16 16 256 256 3" >> $O 2>&1
cat $O
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_exp2_bench.json 2> gpurun_out/r6_exp2_bench.err
python tools/_print_bench.py gpurun_out/r6_exp2_bench.json 2>/dev/null | head -40
