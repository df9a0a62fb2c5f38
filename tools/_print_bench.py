#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line: _print_bench.py file [tag]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2] if len(sys.argv) > 2 else "", round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4),
      {k: round(v["ms_per_step"], 3) for k, v in d.get("kernel_families_all", {}).items()})
