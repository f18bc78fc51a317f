import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in sys.argv[1:]:
    env = dict(os.environ, SGFHE_HIP_LIB=os.path.join(root, "tools/abl/lib_%s.so" % v))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "512", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-host-io"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("%-10s extprod %.1f us  crt %.1f us  value %.0f" % (v, d["roofline"]["launch_ms"] * 1e3, d["roofline"]["pair_launch_ms"] * 1e3, d["value"]), flush=True)
