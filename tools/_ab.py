#!/usr/bin/env python3
"""A/B two builds of libmi355seg.so on conv layers: interleaved rounds of tools/bench_layer.py in child processes on one device.
usage: _ab.py OLD_SO [--dtype bf16] [--what fwd,dgrad,wgrad] [--math bf16x6] -- "N D H W Cin Cout k" ..."""
import os, re, subprocess, sys, collections
args = sys.argv[1:]
old = args.pop(0)
dtype, what, math = "f32", ("fwd", "dgrad"), "bf16x6"
while args and args[0] != "--":
    a = args.pop(0)
    if a == "--dtype": dtype = args.pop(0)
    elif a == "--what": what = tuple(args.pop(0).split(","))
    elif a == "--math": math = args.pop(0)
args.pop(0)
res = collections.defaultdict(lambda: collections.defaultdict(list))
for rnd in range(2):
    for shp in args:
        for tag, lib in (("old", old), ("new", None)):
            env = dict(os.environ)
            if lib: env["MI355SEG_LIB_PATH"] = lib
            cmd = [sys.executable, "tools/bench_layer.py", *shp.split(), "30"] + (["--dtype", "bf16"] if dtype == "bf16" else ["--conv-math", math])
            out = subprocess.run(cmd, capture_output=True, text=True, env=env).stdout
            for l in out.splitlines():
                m = re.match(r"(\w+)\s+([\d.]+) ms\s+([\d.]+) TFLOP/s", l)
                if m and m.group(1) in what: res[(shp, m.group(1))][tag].append(float(m.group(3)))
for (shp, w), v in res.items():
    o, n = sum(v["old"]) / len(v["old"]), sum(v["new"]) / len(v["new"])
    print(f"{shp:28s} {w:6s} old {o:7.1f} new {n:7.1f}  x{n / o:.3f}")
