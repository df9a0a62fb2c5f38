set -e
mkdir -p gpurun_out/r3u
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3u/pytest.log 2>&1 || (tail -60 gpurun_out/r3u/pytest.log; exit 1)
tail -2 gpurun_out/r3u/pytest.log
grep -a "reference bf16\|resunet96 grad" gpurun_out/r3u/pytest.log | head -20
