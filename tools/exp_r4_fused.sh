# Quarter form with its two transform kernels fused into one launch (k_ext_quarter, default) against the two
# launches (SGFHE_SMALL_FUSED=0).  Same call, alternating; then the fused form up to 12 gates per chain.
for i in 1 2; do
echo "== two transform launches (SGFHE_SMALL_FUSED=0)"; SGFHE_SMALL_FUSED=0 python tools/latency.py 1 2 3 4 6 7 8 12 14 2>&1 | grep batch
echo "== fused (k_ext_quarter)"; SGFHE_SMALL_FUSED=1 python tools/latency.py 1 2 3 4 6 7 8 12 14 2>&1 | grep batch
done
echo "== fused, quarter form up to 12 gates per chain (SGFHE_SMALL_SPLIT=12)"; SGFHE_SMALL_FUSED=1 SGFHE_SMALL_SPLIT=12 python tools/latency.py 8 10 12 16 20 24 2>&1 | grep batch
echo "== fused, default split (7)"; SGFHE_SMALL_FUSED=1 python tools/latency.py 8 10 12 16 20 24 2>&1 | grep batch
echo "== Params(512): two launches"; SGFHE_LATENCY_N=512 SGFHE_SMALL_FUSED=0 python tools/latency.py 1 2 4 7 8 2>&1 | grep batch
echo "== Params(512): fused"; SGFHE_LATENCY_N=512 SGFHE_SMALL_FUSED=1 python tools/latency.py 1 2 4 7 8 2>&1 | grep batch
SGFHE_SMALL_FUSED=1 GATES="1 4" bash tools/exp_r4_small_trace.sh
