# k_crt_lean1 (one coefficient per thread) in the latency form against the four-per-thread kernel
# (SGFHE_CRT1_GATES=0), and the threshold in gates.  Same call, alternating.
for i in 1 2; do
echo "== four coefficients per thread (SGFHE_CRT1_GATES=0)"; SGFHE_CRT1_GATES=0 python tools/latency.py 1 2 4 8 2>&1 | grep batch
echo "== one per thread up to 4 gates (default)"; python tools/latency.py 1 2 4 8 2>&1 | grep batch
echo "== one per thread up to 8 gates"; SGFHE_CRT1_GATES=8 python tools/latency.py 1 2 4 8 2>&1 | grep batch
done
