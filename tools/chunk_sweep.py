"""Throughput against the lock-step chunk size at Params(1024): python tools/chunk_sweep.py 256 384 512 ..."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for chunk in sys.argv[1:]:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "4096", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-host-io", "--chunk", chunk], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("chunk %-5s value %.0f  extprod %.1f us  crt %.1f us  per bootstrap-iteration %.4f us" % (
        chunk, d["value"], d["roofline"]["launch_ms"] * 1e3, d["roofline"]["pair_launch_ms"] * 1e3,
        (d["roofline"]["launch_ms"] + d["roofline"]["pair_launch_ms"]) * 1e3 / int(chunk)), flush=True)
