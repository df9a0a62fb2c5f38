set -e
mkdir -p gpurun_out/ct
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -m gpu -x -q -k "transpose" > gpurun_out/ct/t1.log 2>&1 || { tail -30 gpurun_out/ct/t1.log; exit 1; }
tail -1 gpurun_out/ct/t1.log
python tools/bench_convt.py 2 64 64 64 64 32 20 64
python tools/bench_convt.py 2 32 32 32 128 64 20 128
python tools/bench_convt.py 2 16 16 16 256 128 20 256
python tools/bench_convt.py 2 8 8 8 512 256 20 512
python tools/bench_convt.py 2 64 64 64 64 16 20 32 --dtype bf16
python tools/bench_convt.py 2 32 32 32 128 32 20 64 --dtype bf16
python tools/bench_convt.py 2 16 16 16 256 64 20 128 --dtype bf16
