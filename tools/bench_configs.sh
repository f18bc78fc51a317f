#!/bin/bash
# Driver-shaped bench lines for every BASELINE.json configuration that fits one GPU, on the GPU box:
#   usage: tools/bench_configs.sh TAG   -> gpurun_out/bench_TAG_<config>.json (copy into profiles/)
TAG=${1:-rXX}
O=gpurun_out
mkdir -p $O
python bench.py --steps 20 --warmup 2 > $O/bench_${TAG}_params1024.json 2> $O/bench_${TAG}_params1024.err
echo "params1024 done"
python bench.py --config rns2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${TAG}_rns2.json 2> $O/bench_${TAG}_rns2.err
echo "rns2 (config 4) done"
python bench.py --config synth64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${TAG}_synth64.json 2> $O/bench_${TAG}_synth64.err
echo "synth64 (config 3) done"
python bench.py --config params512 --batch 1024 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_${TAG}_params512_b1024.json 2> $O/bench_${TAG}_params512_b1024.err
python bench.py --config params512 --batch 4096 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_${TAG}_params512_b4096.json 2> $O/bench_${TAG}_params512_b4096.err
echo "params512 (config 2) done"
python bench.py --config params64 --batch 16384 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_${TAG}_params64.json 2> $O/bench_${TAG}_params64.err
python bench.py --config params2048 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${TAG}_params2048.json 2> $O/bench_${TAG}_params2048.err
echo "params64 / params2048 done"
python tools/latency.py 1 8 16 24 32 64 256 512 > $O/latency_${TAG}.txt 2>&1
cat $O/latency_${TAG}.txt
for f in $O/bench_${TAG}_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 1),
          "whole_job_frac", round(d["roofline"]["whole_job_frac"], 4), "host_io", round(d.get("host_io", {}).get("value", 0), 1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
