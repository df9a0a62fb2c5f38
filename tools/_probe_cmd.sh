set -e
mkdir -p gpurun_out/r3t
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3t/pytest.log 2>&1 || (tail -40 gpurun_out/r3t/pytest.log; exit 1)
tail -2 gpurun_out/r3t/pytest.log
python tools/bench_model.py res_unet 1 4 160 192 160 --classes 4 --dtype bf16 --steps 5 --dump-launches gpurun_out/r3t/res.csv > gpurun_out/r3t/res.log 2>&1
head -12 gpurun_out/r3t/res.log
