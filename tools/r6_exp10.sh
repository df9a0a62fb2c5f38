#!/bin/bash
# r6 experiment 10: convt_gemm LDS pitch (in-tree BM + 2 vs ab/ctpad8.so BM + 8), x3s staging probes (loads vs split separately), remaining tests
O=gpurun_out/r6_exp10.log
: > $O
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -4 >> $O
echo "== ConvT k2 s2 layers: in-tree (voxel pitch BM + 2) then ab/ctpad8.so (BM + 8)" >> $O
for shp in "1 6 6 6 768 512" "1 12 12 12 512 256" "1 24 24 24 256 128" "2 8 8 8 512 256" "2 16 16 16 256 128"; do
  for lib in "" "$PWD/ab/ctpad8.so"; do
    for dt in bf16 f32; do
      echo "-- $shp $dt lib=${lib:-in-tree}" >> $O
      MI355SEG_LIB_PATH=$lib python tools/bench_convt.py $shp 30 --dtype $dt 2>&1 | grep "fwd\|dgrad" >> $O
    done
  done
done
echo "== x3s staging probes on the Cout = 32 layers (TUNE build): 0 plain, 16 halo loads not requested (split + LDS writes of stale registers), 4 loads only (no split / LDS writes), 1 neither" >> $O
export MI355SEG_LIB_PATH=$PWD/ab/tune.so
export MI355SEG_NO_X3W=1
for shp in "2 128 128 128 32 32" "2 128 128 128 64 32"; do
  for d in 0 16 4 1 0 16 4 1; do
    echo "-- $shp DBG=$d" >> $O
    MI355SEG_DBG=$d python tools/bench_layer.py $shp 3 30 --conv-math f16x3 2>&1 | grep "^fwd\|^dgrad" >> $O
  done
done
cat $O
