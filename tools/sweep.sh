#!/bin/bash
# Throughput against the lock-step chunk size and the number of lanes at Params(1024), on the GPU box:
#   tools/sweep.sh "1:512 2:192 2:256 2:320"     (lanes:chunk pairs)
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for lc in $1; do
  $B --lanes ${lc%%:*} --chunk ${lc##*:} | python tools/result_line.py l${lc%%:*}_c${lc##*:}
done
