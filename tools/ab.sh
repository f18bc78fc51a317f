#!/bin/bash
# tools/ab.sh -- the same-call A/B experiments of rounds 2-5 behind one driver (the one-off
# tools/exp_*.sh scripts folded into a table).  Boxes of the pool differ by up to 7 % on one binary, so only
# numbers from ONE gpurun call are comparable: every experiment alternates its variants inside one call.
#
#   tools/ab.sh list                  the experiments, their variant builds (-D sets) and the profile that holds the result
#   tools/ab.sh build <experiment>    BUILD CONTAINER: compile the experiment's variant libraries into tools/abl/
#                                     (git-ignored; they travel to the GPU box with the gpurun snapshot)
#   tools/ab.sh run <experiment>      GPU BOX (gpurun -- 'bash tools/ab.sh run X > gpurun_out/X.txt 2>&1'): run it
#
# A variant library is csrc/ compiled with extra -D flags (csrc/Makefile EXTRA=...): the flags become part of
# sgfhe_build_id(), so bench.py never quotes committed profile counters beside such a library.  Timing-only
# variants (SGFHE_ABL_*) give WRONG results by construction.  Variants chosen by environment need no build.
# Result lines: "RESULT <name> <bootstraps/s> iter_us .. ext_us .. crt_us .. lanes .. chunk .." (tools/result_line.py)
# or the "batch N: .. ms" lines of tools/latency.py.
set -u
cd "$(dirname "$0")/.."
ABL=tools/abl
B3="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-live-counters"
R="python tools/result_line.py"

# ---- the table: experiment | variant builds "name:flags" | result file | what it compares ------------------------------
table() { cat <<'EOF'
ab            | (by hand: cp sgfhe.jl_amd/csrc/libsgfhe_hip.so tools/abl/lib_base.so before the change) | (ad hoc)                          | two builds of the library at the default workload, then the GPU suite
ablate        | nobar:-DSGFHE_ABL_NO_BARRIER nolds:-DSGFHE_ABL_NO_LDS notw:-DSGFHE_ABL_NO_TW nokey:-DSGFHE_ABL_NO_KEY nodig:-DSGFHE_ABL_NO_DIG nomem:-DSGFHE_ABL_NO_TW,-DSGFHE_ABL_NO_KEY,-DSGFHE_ABL_NO_DIG valuonly:-DSGFHE_ABL_NO_TW,-DSGFHE_ABL_NO_KEY,-DSGFHE_ABL_NO_DIG,-DSGFHE_ABL_NO_LDS,-DSGFHE_ABL_NO_BARRIER | r02_ablation.txt | where k_extprod's time goes: parts removed (timing only)
crt_ablation  | crt_nosd:-DSGFHE_ABL_CRT_NOSD crt_memonly:-DSGFHE_ABL_CRT_MEMONLY              | r02_exp_crt_ablation.txt          | what bounds the CRT kernel: arithmetic removed (timing only)
crt_ceiling   | crt_memonly:-DSGFHE_ABL_CRT_MEMONLY                                          | r03_exp_crt_ceiling.txt           | the overlap ceiling if the CRT arithmetic were free, one and two lanes
crt_lean      | (env SGFHE_CRT_LEAN=0)                                                       | r03_exp_crt_lean.txt              | k_crt_lean against k_crt_acc2, one and two lanes
crt_lean_rnd  | (env SGFHE_CRT_LEAN=0)                                                       | r03_exp_crt_lean_random.txt       | k_crt_lean_rnd against k_crt_acc, randomised flatten, three rings
lanes_sweep   | prio2:-DSGFHE_EXT_PRIO=2                                                     | r03_exp_lanes_sweep.txt           | two lanes: chunk sweep; k_extprod at issue priority 2
cache_policy  | plain:-DSGFHE_CRT_PLAIN_LOADS yres_plain:-DSGFHE_YRES_AUX=0 yres_nt:-DSGFHE_YRES_AUX=2 | r03_exp_cache_policy.txt | cache policy of the residue hand-off with two lanes
load_stalls   | NO_DIG:-DSGFHE_ABL_NO_DIG NO_KEY:-DSGFHE_ABL_NO_KEY NO_DIG_KEY:-DSGFHE_ABL_NO_DIG,-DSGFHE_ABL_NO_KEY | r03_exp_load_stalls.txt | upper bound on what k_extprod's digit and key loads cost (timing only)
mid_batches   | (none)                                                                       | r03_exp_mid_batches.txt           | batches of 32-640 gates: automatic split over two lanes against one lane
radix4        | (by hand: lib_base.so = the library before ntt.h fwd_step4 / inv_step4)      | r03_exp_radix4.txt                | radix-4 steps with deferred reductions against the radix-2 transforms
radix4_spills | (by hand: lib_r4all.so = EXTRA="-D'SGFHE_FWD_VEC4(L)=1' -D'SGFHE_INV_R4(L)=1'", lib_r4inv.so = EXTRA="-D'SGFHE_INV_R4(L)=1'") | r03_exp_radix4_spills.txt         | the radix-4 steps where they spill (m = 4096, 16384)
key_prefetch  | (one-off patch of round 3, not kept in the tree: key loads of k_fwd_phase after its transform) | r03_exp_key_prefetch.txt          | small-batch form: key rows requested before / after the forward transform
le3           | le3_12:-DSGFHE_EXT_LE3_MAX=12                                                | r02_exp_p512_points.txt           | Params(512): k_extprod with 16 or 8 points per thread
mid_le3       | le3_13:-DSGFHE_EXT_LE3_MAX=13                                                | r04_exp_mid_le3.txt               | calls of 25-256 gates at Params(1024): 16 or 8 points per thread
overlap       | acc32:-DSGFHE_ACC0_32                                                        | r02_exp_overlap_vgpr.txt          | does a 104-VGPR k_extprod leave room for the CRT's waves (two lanes)
sched         | maxilp:-mllvm,-amdgpu-sched-strategy=max-ilp maxmem:-mllvm,-amdgpu-sched-strategy=max-memory-clause bias100:-mllvm,-amdgpu-schedule-metric-bias=100 | r02_exp_sched.txt | compiler scheduling strategies for the whole library
epilogue      | epi_plain:-DSGFHE_EPI_PLAIN                                                  | r04_exp_epilogue.txt              | rotation epilogue in the plain LDS layout against the swizzled one
clock_handoff | no_yres:-DSGFHE_ABL_NO_YRES no_yload:-DSGFHE_ABL_NO_YLOAD no_both:-DSGFHE_ABL_NO_YRES,-DSGFHE_ABL_NO_YLOAD | r04_exp_clock_handoff.txt | what the residue hand-off costs, with clock and power sampled (timing only)
iter_all      | iter_all:-DSGFHE_WITH_ITER_ALL                                               | r04_exp_iter_all.txt              | one launch per iteration (all primes of a bootstrap in one workgroup)
crt1          | (env SGFHE_CRT1_GATES)                                                       | r04_exp_crt1.txt                  | latency form: one coefficient per CRT thread, threshold in gates (default 8)
small_unpadded| (env SGFHE_SMALL_PADDED=1)                                                   | r04_exp_small_unpadded.txt        | latency form: grids sized by the gates in the call against padding to 8
small_args    | (one-off patch of round 4, not kept in the tree: PrimeSet by value in the kernel arguments) | r04_exp_small_args.txt            | latency form: per-prime constants by value against the record pointer
small_lanes   | (env SGFHE_SMALL_LANES=0)                                                    | r04_exp_small_lanes.txt           | a call of a few gates as two halves on two lanes against one stream
quarter       | (env SGFHE_SMALL_SPLIT)                                                      | r04_exp_quarter.txt               | latency form: each transform across four workgroups
fused         | (env SGFHE_SMALL_FUSED, SGFHE_SMALL_SPLIT)                                   | r04_exp_fused.txt                 | quarter form: its two transform kernels as one launch
fused_forms   | (env)                                                                        | r04_exp_fused.txt                 | which latency form for which call size, Params(1024) and (512), 1-24 gates
io            | (env SGFHE_IO_EXP)                                                           | r04_exp_io_variants.txt           | which part of the pipelined host-pointer path costs the k-loop time
launch_env    | (by hand: hipcc --offload-arch=gfx950 -O3 -o tools/abl/ubs_base tools/ubench_split.hip) | r04_exp_launch_env.txt            | runtime environment knobs against the dependent-launch floor
small_trace   | (none)                                                                       | r04_latency_trace.txt             | rocprofv3 kernel-trace averages of the latency-form kernels at GATES per call
wide          | wide_split2:-DSGFHE_WIDE_SPLIT=2 wide_split4:-DSGFHE_WIDE_SPLIT=4 wide_acc32:-DSGFHE_WIDE_ACC32 | r05_exp_wide.txt | WIDE k_extprod (Params(2048), randomised): three ways to take its scratch out
sweep         | (none)                                                                       | r02_chunk_sweep.txt               | throughput against lanes:chunk pairs: tools/ab.sh run sweep "1:512 2:192 2:256"
clock_probe   | (none)                                                                       | r02_clock_probe.txt               | clock / power / temperature sampled under the default workload
EOF
}
# (One-off source patches of rounds 2-3 that are not kept in the tree -- non-temporal old-digit loads, digit-store
#  and digit-load policies, the 16-byte epilogue, ds_add_u32, signed digit planes -- have no entry: what they
#  changed is described in the profiles/r0x_exp_*.txt file that holds their result.)

variants_of() { table | awk -F'|' -v e="$1" '{gsub(/^ +| +$/, "", $1)} $1 == e {print $2}'; }

build() {
    mkdir -p $ABL
    local vs; vs=$(variants_of "$1")
    [ -z "$vs" ] && { echo "unknown experiment '$1' (tools/ab.sh list)"; exit 2; }
    case "$vs" in *\(*) echo "nothing to build: $vs"; return ;; esac      # (env ...), (none), (by hand ...)
    for v in $vs; do
        name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
        echo "-- lib_$name.so: $flags"
        make -s -C sgfhe.jl_amd/csrc -B OUT=../../$ABL/lib_$name.so EXTRA="$flags" ../../$ABL/lib_$name.so || exit 1
    done
}

lib() { if [ "$1" = base ]; then unset SGFHE_HIP_LIB; else export SGFHE_HIP_LIB=$PWD/$ABL/lib_$1.so; fi; }
lat() { python tools/latency.py "$@" 2>&1 | grep batch; }
probe() {   # bench in the background, rocm-smi sampled under it
    $B3 --steps 5 > gpurun_out/clock_probe_$1.json 2>/dev/null &
    local bp=$!
    sleep 9
    for i in 1 2 3; do
        rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/GPU\[0\]\t\t: //; s/clock level: 1: //; s/Current Socket Graphics Package //' | tr '\n' ' '; echo
        sleep 1.5
    done
    wait $bp
    $R $1 < gpurun_out/clock_probe_$1.json
}

run() {
    local e=$1; shift
    case "$e" in
    ab)     for i in 1 2; do lib base; SGFHE_HIP_LIB=$PWD/$ABL/lib_base.so $B3 | $R base_$i; unset SGFHE_HIP_LIB; $B3 | $R new_$i; done
            python -m pytest tests -q -m gpu -x 2>&1 | tail -3 ;;
    ablate) for i in 1 2; do for v in base nobar nolds notw nokey nodig nomem valuonly; do lib $v; $B3 --lanes 1 | $R ${v}_$i; done; done ;;
    crt_ablation) for i in 1 2; do for v in base crt_nosd crt_memonly; do lib $v; $B3 | $R ${v}_$i; done; done ;;
    crt_ceiling) for i in 1 2; do for v in base crt_memonly; do lib $v; $B3 --lanes 1 | $R ${v}_l1_c512_$i; $B3 --lanes 2 --chunk 256 | $R ${v}_l2_c256_$i; done; done ;;
    crt_lean) for i in 1 2; do
            SGFHE_CRT_LEAN=0 $B3 --lanes 1 | $R acc2_l1_c512_$i; $B3 --lanes 1 | $R lean_l1_c512_$i
            SGFHE_CRT_LEAN=0 $B3 --lanes 2 --chunk 256 | $R acc2_l2_c256_$i; $B3 --lanes 2 --chunk 256 | $R lean_l2_c256_$i
            $B3 --lanes 2 --chunk 512 | $R lean_l2_c512_$i; done ;;
    crt_lean_rnd) python -m pytest tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -1
            for cfg in "" "--config params2048 --batch 1024 --steps 2" "--config params512 --batch 4096"; do for i in 1 2; do
            SGFHE_CRT_LEAN=0 $B3 --flatten random $cfg | $R "k_crt_acc_${cfg// /_}_$i"; $B3 --flatten random $cfg | $R "lean_rnd_${cfg// /_}_$i"; done; done ;;
    lanes_sweep) for i in 1 2; do lib base; $B3 --lanes 1 | $R lean_l1_c512_$i
            for c in 128 192 256 320 384; do $B3 --lanes 2 --chunk $c | $R lean_l2_c${c}_$i; done
            lib prio2; $B3 --lanes 2 --chunk 256 | $R prio2_l2_c256_$i; $B3 --lanes 1 | $R prio2_l1_c512_$i; done ;;
    cache_policy) for i in 1 2; do for v in base plain yres_plain yres_nt; do lib $v; $B3 --no-isolated | $R ${v}_$i; done
            lib base; $B3 --no-isolated --chunk 192 | $R base_c192_$i; done ;;
    load_stalls) for i in 1 2; do for v in base NO_DIG NO_KEY NO_DIG_KEY; do lib $v; $B3 --no-isolated | $R ${v}_$i; done; done
            for v in base NO_DIG_KEY; do lib $v; $B3 --no-isolated --lanes 1 | $R ${v}_one_lane; done ;;
    mid_batches) for b in 32 48 64 100 128 256 384 640; do
            $B3 --steps 5 --no-isolated --batch $b --lanes 1 | $R b${b}_one_lane; $B3 --steps 5 --no-isolated --batch $b | $R b${b}_auto; done ;;
    radix4) for i in 1 2 3; do for v in base new; do [ $v = base ] && export SGFHE_HIP_LIB=$PWD/$ABL/lib_base.so || unset SGFHE_HIP_LIB; $B3 --no-isolated | $R ${v}_$i; done; done ;;
    radix4_spills) for i in 1 2 3; do for v in base r4all r4inv; do lib $v; $B3 --steps 5 --no-isolated --config params512 --batch 4096 | $R p512_${v}_$i; done; done
            for v in base r4all; do lib $v; $B3 --steps 2 --no-isolated --config params2048 --batch 1024 | $R p2048_$v; done ;;
    key_prefetch) for i in 1 2; do for v in keylate base; do lib $v; echo "== $v"; lat 1 8 16 24; done; done ;;
    le3)    for i in 1 2; do for v in base le3_12; do lib $v; $B3 --config params512 --batch 4096 | $R p512_${v}_$i; done; done ;;
    mid_le3) for i in 1 2; do for v in base le3_13; do lib $v; echo "== $v"; lat 25 32 40 48 51 56 64 80 96 128 256; done; done ;;
    overlap) for i in 1 2; do for v in base acc32; do lib $v; $B3 --lanes 1 | $R ${v}_lanes1_$i; $B3 --lanes 2 | $R ${v}_lanes2_$i; done; done ;;
    sched)  for i in 1 2; do for v in base maxilp maxmem bias100; do lib $v; $B3 | $R ${v}_$i; done; done ;;
    epilogue) for i in 1 2 3; do for v in base epi_plain; do lib $v; $B3 --no-isolated | $R ${v}_$i; done; done
            for cfg in params512 params2048; do for v in base epi_plain; do lib $v; $B3 --no-isolated --config $cfg --batch 1024 | $R ${v}_$cfg; done; done ;;
    clock_handoff) for i in 1 2; do for v in base no_yres no_yload no_both; do lib $v; echo "== $v"; probe ${v}_$i; done; done ;;
    iter_all) lib iter_all; SGFHE_ITER_ALL=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "params1024_vs_oracle" 2>&1 | tail -1
            for i in 1 2; do $B3 | $R two_kernels_$i; SGFHE_ITER_ALL=1 $B3 | $R iter_all_$i; SGFHE_ITER_ALL=1 $B3 --lanes 1 --chunk 256 | $R iter_all_one_lane_256_$i; done ;;
    crt1)   for i in 1 2; do for g in 0 4 8; do echo "== one coefficient per CRT thread up to $g gates (0 = never; default 8)"; SGFHE_CRT1_GATES=$g lat 1 2 4 8; done; done ;;
    small_unpadded) for i in 1 2; do echo "== padded to 8 (rounds 1-3)"; SGFHE_SMALL_PADDED=1 lat 1 2 4 8 12 16 24; echo "== unpadded"; lat 1 2 4 8 12 16 24; done ;;
    small_args) for i in 1 2; do for v in small_ptr base; do lib $v; echo "== $v"; lat 1 8 16 24; done; done ;;
    small_lanes) for i in 1 2; do echo "== one chunk, one stream"; SGFHE_SMALL_LANES=0 lat 1 2 3 4 6 8 10 12 14 16 20 24 32; echo "== two halves on two lanes"; lat 1 2 3 4 6 8 10 12 14 16 20 24 32; done ;;
    quarter) for i in 1 2; do echo "== one workgroup per transform"; SGFHE_SMALL_SPLIT=0 lat 1 2 3 4 6 8; echo "== quarter form up to 8 gates"; SGFHE_SMALL_SPLIT=8 lat 1 2 3 4 6 8; done ;;
    fused)  for i in 1 2; do echo "== two transform launches"; SGFHE_SMALL_FUSED=0 lat 1 2 3 4 6 7 8 12 14; echo "== fused (k_ext_quarter from one gate)"; SGFHE_SMALL_FUSED=1 lat 1 2 3 4 6 7 8 12 14; done
            for N in 512; do export SGFHE_LATENCY_N=$N; echo "== Params($N) two launches"; SGFHE_SMALL_FUSED=0 lat 1 2 4 7 8; echo "== Params($N) fused"; SGFHE_SMALL_FUSED=1 lat 1 2 4 7 8; done ;;
    fused_forms) S="1 2 4 6 7 8 10 12 13 14 16 20 24"
            for N in 1024 512; do export SGFHE_LATENCY_N=$N; for i in 1 2; do
            echo "== Params($N) before k_ext_quarter"; SGFHE_SMALL_FUSED=0 lat $S; echo "== Params($N) shipped"; lat $S; done; done ;;
    io)     for x in 0 1 2 4 6 7 0; do echo "== SGFHE_IO_EXP=$x"; SGFHE_IO_EXP=$x SGFHE_DEBUG_IO=1 python tools/io_phases.py params1024 4096 2>&1 | grep -v "pin_\|amdgpu.ids"; done ;;
    launch_env) for kv in A=1 ROC_SYSTEM_SCOPE_SIGNAL=0 HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000 HIP_FORCE_DEV_KERNARG=0 GPU_MAX_HW_QUEUES=1 \
                ROC_SIGNAL_POOL_SIZE=256 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 ROC_SKIP_KERNEL_ARG_COPY=1 HSA_FORCE_FINE_GRAIN_PCIE=1 ROC_AQL_QUEUE_SIZE=4096 A=1; do
            echo "== $kv"; env $kv UB_ANATOMY=x timeout -k 5 60 $ABL/ubs_base; done ;;
    small_trace) export TMPDIR=/tmp
            for b in ${GATES:-1 2 4 8 16}; do rm -rf gpurun_out/trace_small_$b
            rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_small_$b -o run -- python3 tools/latency.py $b > gpurun_out/trace_small_$b.log 2>&1
            echo "== $b gates per call: kernel, calls, average ns"
            python3 - gpurun_out/trace_small_$b <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_fwd_phase", "k_inv_column", "k_fwd_quarter", "k_inv_quarter", "k_ext_quarter", "k_crt_lean", "k_init", "k_final")):
        print("   %-60s %8s %10.0f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
            done ;;
    wide)   W="--config params2048 --batch 1024 --flatten random --steps 2"
            for v in base wide_split2 wide_split4 wide_acc32; do lib $v; echo "== $v: parity"; python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "p2048" 2>&1 | tail -1; done
            for i in 1 2; do for v in base wide_split2 wide_split4 wide_acc32; do lib $v; $B3 $W | $R ${v}_$i; done; done ;;
    sweep)  for lc in ${1:-"1:512 2:192 2:256 2:320"}; do $B3 --steps 2 --no-isolated --lanes ${lc%%:*} --chunk ${lc##*:} | $R l${lc%%:*}_c${lc##*:}; done ;;
    clock_probe) python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated --no-live-counters > gpurun_out/clock_probe_bench.json 2>/dev/null &
            bp=$!; sleep 8
            for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)|fclk" | head -8; echo "--"; sleep 2; done
            wait $bp; cut -c1-90 gpurun_out/clock_probe_bench.json ;;
    *)      echo "unknown experiment '$e' (tools/ab.sh list)"; exit 2 ;;
    esac
    unset SGFHE_HIP_LIB
}

case "${1:-list}" in
    list)  table | awk -F'|' '{printf "%-15s %s\n                result: profiles/%s\n                builds: %s\n", $1, $4, $3, $2}' | sed 's/  */ /g; s/^ *result/                result/; s/^ *builds/                builds/' ;;
    build) build "${2:?experiment}" ;;
    run)   shift; run "$@" ;;
    *)     echo "usage: tools/ab.sh list | build <experiment> | run <experiment> [args]"; exit 2 ;;
esac
