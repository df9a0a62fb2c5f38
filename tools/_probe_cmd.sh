set -e
mkdir -p gpurun_out/r3w
timeout -k 10 600 python -m pytest tests/test_custom_ops.py -m gpu -x -q > gpurun_out/r3w/pytest.log 2>&1 || (tail -60 gpurun_out/r3w/pytest.log; exit 1)
tail -2 gpurun_out/r3w/pytest.log
