"""Condense a bench.py JSON line (stdin) into one RESULT line for the A/B scripts under tools/:
    python bench.py ... | python tools/result_line.py NAME"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
k = r["kernels"]
crt = next(v for name, v in k.items() if name != "k_extprod")   # k_crt_lean, or whichever CRT kernel ran
print("RESULT", sys.argv[1], round(d["value"], 1), "iter_us", round(r["launch_ms"] * 1e3, 1),
      "ext_us", round(k["k_extprod"]["launch_ms"] * 1e3, 1), "crt_us", round(crt["launch_ms"] * 1e3, 1),
      "lanes", d["config"]["lanes"], "chunk", d["config"]["chunk"])
