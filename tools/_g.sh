for m in "unetr 1 1 96 96 96" "vnet 2 1 128 128 128" "res_unet 1 4 160 192 160 --classes 4"; do
  python tools/bench_model.py $m --dtype bf16 --steps 10 --no-prof 2>&1 | grep "ms/step"
  python tools/bench_model.py $m --dtype bf16 --steps 10 --no-prof --graph 2>&1 | grep "ms/step\|rror" | head -3
done
