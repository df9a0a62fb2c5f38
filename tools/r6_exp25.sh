#!/bin/bash
# r6 experiment 25: conv_b16s persistent walk with the next tile's first weight fragments requested ahead of the stores (in-tree) and the restructured kernel
# without the walk (ab/nopersist.so), each against ab/base.so (HEAD)
O=gpurun_out/r6_exp25.log
: > $O
python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -2 >> $O
SH=("1 160 192 160 32 32 3" "1 160 192 160 64 32 3" "1 80 96 80 64 64 3" "1 80 96 80 128 64 3" "1 96 96 96 64 64 3" "1 160 192 160 32 64 3" "1 40 48 40 128 128 3")
echo "== persistent (in-tree, 'new') vs base ('old')" >> $O
python tools/_ab.py $PWD/ab/base.so --dtype bf16 --what fwd,dgrad -- "${SH[@]}" >> $O 2>&1
echo "== base ('old') vs nopersist ('new': MI355SEG_LIB_PATH of the 'new' arm = ab/nopersist.so)" >> $O
cp -f general-medical-image-segmentation-cnn-framework_amd/libmi355seg.so /tmp/intree_keep.so
cp -f ab/nopersist.so general-medical-image-segmentation-cnn-framework_amd/libmi355seg.so
python tools/_ab.py $PWD/ab/base.so --dtype bf16 --what fwd,dgrad -- "${SH[@]}" >> $O 2>&1
cp -f /tmp/intree_keep.so general-medical-image-segmentation-cnn-framework_amd/libmi355seg.so
cat $O
