#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the default bench, then separate PMC
# passes (never combined with other trace domains) on a one-chunk batch.  Results land in
# gpurun_out/prof_$TAG; tools/summarize_profile.py turns them into the files kept under profiles/.
#   usage: tools/profile_round.sh TAG [CHUNK]
set -e
TAG=${1:-rXX}
CHUNK=${2:-256}   # one chunk of the engine default size at Params(1024) (two lanes of 256)
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the program after `rocprofv3 ... --` has to be the interpreter binary itself, not a wrapper script or a PATH
# shim: the profiler's preloaded library initialises the GPU before the program starts, and an exec behind that
# is forbidden on this pool (bench.py's own child passes resolve sys.executable the same way)
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
head -c 4 "$PY" | grep -q ELF || { echo "$PY is not an ELF interpreter"; exit 1; }
# the default schedule: two lanes, the kernels of the two chunks overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- "$PY" bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated > $OUT/bench_stats.log 2>&1
echo "stats done"
# the same chunks one after the other: every kernel alone on the device
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_alone -o run -- "$PY" bench.py --lanes 1 --chunk $CHUNK --steps 1 --warmup 0 --no-cpu-baseline --no-host-io > $OUT/bench_stats_alone.log 2>&1
echo "stats (one lane) done"
ONE="--lanes 1 --chunk $CHUNK --batch $CHUNK --steps 1 --warmup 0 --no-cpu-baseline --no-host-io"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o run -- "$PY" bench.py $ONE > $OUT/pmc_$C.log 2>&1
  echo "$C done"
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_SQ -o run -- "$PY" bench.py $ONE > $OUT/pmc_SQ.log 2>&1
echo "SQ done"
# where the wave cycles go (optional passes: a counter this ROCm does not know must not stop the round)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQW -o run -- "$PY" bench.py $ONE > $OUT/pmc_SQW.log 2>&1 || echo "SQW pass failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_SQL -o run -- "$PY" bench.py $ONE > $OUT/pmc_SQL.log 2>&1 || echo "SQL pass failed"
echo "SQ wait/LDS done"
python3 tools/summarize_profile.py $OUT $TAG $CHUNK
