mkdir -p gpurun_out/r4a
python tools/bench_model.py unet 2 1 128 128 128 --dtype f32 --steps 5 --dump-launches gpurun_out/r4a/unet_f32.csv > gpurun_out/r4a/unet_f32.log 2>&1
head -10 gpurun_out/r4a/unet_f32.log
