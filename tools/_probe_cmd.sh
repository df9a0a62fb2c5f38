set -e
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/pytest.log 2>&1 || (tail -40 gpurun_out/r3g/pytest.log; exit 1)
tail -3 gpurun_out/r3g/pytest.log
python bench.py --steps 20 --warmup 5 --no-workloads --no-cpu-baseline --dump-launches gpurun_out/r3g/launches.csv > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err
MI355SEG_X3_SHAPE=32 python bench.py --steps 20 --warmup 5 --no-workloads --no-cpu-baseline > gpurun_out/r3g/bench_s32.json 2> gpurun_out/r3g/bench_s32.err
python bench.py --steps 20 --warmup 5 --no-workloads --no-cpu-baseline > gpurun_out/r3g/bench2.json 2> gpurun_out/r3g/bench2.err
python - <<'PY'
import json
for f in ['bench','bench_s32','bench2']:
    r=json.load(open(f'gpurun_out/r3g/{f}.json'))
    print(f, round(r['ms_per_step'],2), {k:(round(v['ms_per_step'],2), round(v['tflops'],1)) for k,v in r['kernel_families'].items()}, round(r['roofline']['frac'],3))
PY
