#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of one bench step of the configurations other
# than the headline (the headline's own passes: tools/profile_round.sh).  One summary per configuration lands in
# gpurun_out/prof_cfg_$TAG/<name>_kernel_stats.csv; the ten busiest kernels of each in summary.txt.
#   usage: tools/profile_configs.sh TAG
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_cfg_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
head -c 4 "$PY" | grep -q ELF || { echo "$PY is not an ELF interpreter"; exit 1; }
: > $OUT/summary.txt
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o run -- "$PY" bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name 'run_kernel_stats.csv' | head -1)
  # the engine's kernels only (bench.py's synthetic inputs come from torch kernels with kilobyte-long names)
  { head -1 "$f"; grep 'sgfhe::' "$f"; } > $OUT/${name}_kernel_stats.csv
  { echo "== $name: bench.py $*"; grep '^{"metric"' $OUT/$name.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("   %.1f %s, whole_job_frac %.4f, primes %d, chunk %d" % (d["value"], d["unit"], d["roofline"]["whole_job_frac"], d["config"]["rns_primes"], d["config"]["chunk"]))'; python3 - $OUT/${name}_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    name = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("sgfhe::", ""))
    print("   %-34s calls %6s  average %9.2f us  (min %8.2f, max %8.2f)  %6s %%" % (
        name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
  } >> $OUT/summary.txt
  rm -rf $OUT/$name      # the kernel trace itself is tens of megabytes per configuration
  echo "$name done"
}
run params1024_random --flatten random
run synth64 --config synth64
run rns2 --config rns2
run params512_b4096 --config params512 --batch 4096
run params2048 --config params2048 --batch 1024
run params2048_random --config params2048 --batch 1024 --flatten random
run params64 --config params64 --batch 16384
cat $OUT/summary.txt
