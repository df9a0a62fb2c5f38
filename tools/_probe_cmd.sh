set -e
mkdir -p gpurun_out/r3v
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3v/pytest.log 2>&1 || (tail -60 gpurun_out/r3v/pytest.log; exit 1)
tail -2 gpurun_out/r3v/pytest.log
for d in f32 bf16; do python tools/bench_predict.py unet --dtype $d >> gpurun_out/r3v/predict.log 2>&1; done
python tools/bench_predict.py vnet --dtype bf16 >> gpurun_out/r3v/predict.log 2>&1
cat gpurun_out/r3v/predict.log
