set -e
O=gpurun_out/r03final; mkdir -p $O
python bench.py --steps 20 --warmup 5 --dump-launches $O/per_launch.csv > $O/bench3.json 2> $O/bench3.err
( for s in "2 64 64 64 64 32 20 64" "2 32 32 32 128 64 20 128" "2 16 16 16 256 128 20 256" "2 8 8 8 512 256 20 512"; do python tools/bench_convt.py $s; done
  for s in "2 64 64 64 64 16 20 32" "2 32 32 32 128 32 20 64" "2 16 16 16 256 64 20 128" "2 8 8 8 256 128 20 256"; do python tools/bench_convt.py $s --dtype bf16; done
  python tools/bench_convt.py 2 64 64 64 64 32 20 64 --conv-math fp32 ) > $O/convt_layers.log 2>/dev/null
python tools/bench_predict.py unet --dtype f32 > $O/predict.jsonl 2>/dev/null; python tools/bench_predict.py unet --dtype bf16 >> $O/predict.jsonl 2>/dev/null; python tools/bench_predict.py vnet --dtype bf16 >> $O/predict.jsonl 2>/dev/null
tail -c 300 $O/bench3.json; cat $O/convt_layers.log | head -8; tail -3 $O/predict.jsonl | cut -c1-200
