set -e
mkdir -p gpurun_out/r3p
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3p/pytest.log 2>&1 || (tail -40 gpurun_out/r3p/pytest.log; exit 1)
tail -3 gpurun_out/r3p/pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r3p/bench.json'))
print('cfg2', round(r['ms_per_step'],2), round(r['roofline']['frac'],3), {k:(round(v['ms_per_step'],2), round(v['tflops'],1)) for k,v in r['kernel_families'].items()})
for n,l in r['workloads'].items():
    print(n, round(l.get('ms_per_step',0),2), l.get('error'), {k:(round(v['ms_per_step'],2), round(v['tflops'],1), round(v['gbs'])) for k,v in l.get('kernel_families',{}).items()})
PY
