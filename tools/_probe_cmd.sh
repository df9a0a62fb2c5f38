mkdir -p gpurun_out/r3y
python tools/_ab.py general-medical-image-segmentation-cnn-framework_amd/libmi355seg_old.so --what wgrad -- "2 128 128 128 32 32 3" "2 128 128 128 64 32 3" "2 64 64 64 64 64 3" "2 32 32 32 128 128 3" 2>&1 | tee gpurun_out/r3y/ab.log
python tools/_ab.py general-medical-image-segmentation-cnn-framework_amd/libmi355seg_old.so --what wgrad --dtype bf16 -- "1 160 192 160 64 64 3" "2 128 128 128 32 32 5" "1 160 192 160 32 64 3 30 2 1" 2>&1 | tee -a gpurun_out/r3y/ab.log
