# rocprofv3 kernel-trace averages of the three small-batch kernels at 1, 2, 4, 8 and 16 gates per call
export TMPDIR=/tmp
for b in ${GATES:-1 2 4 8 16}; do
  rm -rf gpurun_out/trace_small_$b
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_small_$b -o run -- python3 tools/latency.py $b > gpurun_out/trace_small_$b.log 2>&1
  echo "== $b gates per call: kernel, calls, average ns"
  python3 - gpurun_out/trace_small_$b <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_fwd_phase", "k_inv_column", "k_fwd_quarter", "k_inv_quarter", "k_ext_quarter", "k_crt_lean", "k_init", "k_final")):
        print("   %-60s %8s %10.0f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
done
