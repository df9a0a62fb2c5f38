set -e
mkdir -p gpurun_out/bn
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_custom_ops.py -m gpu -x -q > gpurun_out/bn/t1.log 2>&1 || { tail -40 gpurun_out/bn/t1.log; exit 1; }
tail -1 gpurun_out/bn/t1.log
bash tools/_quick.sh bn
