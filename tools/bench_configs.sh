#!/bin/bash
# Driver-shaped bench lines for every BASELINE.json configuration that fits one GPU, on the GPU box:
#   usage: tools/bench_configs.sh TAG   -> gpurun_out/bench_TAG_<config>.json (copy into profiles/)
TAG=${1:-rXX}
O=gpurun_out
mkdir -p $O
run() { name=$1; shift; python bench.py "$@" > $O/bench_${TAG}_$name.json 2> $O/bench_${TAG}_$name.err; echo "$name done"; }
run params1024 --steps 20 --warmup 2
run params1024_b8192 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline          # config 5's per-GPU shard
run params1024_one_lane --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-io
run params1024_random --flatten random --steps 3 --warmup 1 --no-cpu-baseline --no-host-io   # six primes
run rns2 --config rns2 --steps 3 --warmup 1 --no-cpu-baseline                     # config 4
run synth64 --config synth64 --steps 3 --warmup 1 --no-cpu-baseline               # config 3
run params512_b1024 --config params512 --batch 1024 --steps 10 --warmup 2 --no-cpu-baseline   # config 2
run params512_b4096 --config params512 --batch 4096 --steps 5 --warmup 1 --no-cpu-baseline
run params64 --config params64 --batch 16384 --steps 10 --warmup 2 --no-cpu-baseline
run params2048 --config params2048 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline
run params2048_random --config params2048 --batch 1024 --flatten random --steps 2 --warmup 1 --no-cpu-baseline --no-host-io
python tools/latency.py 1 2 4 7 8 12 16 24 32 64 128 256 512 > $O/latency_${TAG}.txt 2>&1
cat $O/latency_${TAG}.txt
for f in $O/bench_${TAG}_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d["config"]
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 1),
          "lanes", c["lanes"], "chunk", c["chunk"], "primes", c["rns_primes"],
          "whole_job_frac", round(d["roofline"]["whole_job_frac"], 4), "host_io", round(d.get("host_io", {}).get("value", 0), 1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
