# Latency form with each transform cut across four workgroups (k_fwd_quarter / k_inv_quarter / k_crt_lean1q, up to
# SGFHE_SMALL_SPLIT gates) against the one-workgroup transforms (SGFHE_SMALL_SPLIT=0).  Same call, alternating.
for i in 1 2; do
echo "== one workgroup per transform (SGFHE_SMALL_SPLIT=0)"; SGFHE_SMALL_SPLIT=0 python tools/latency.py 1 2 3 4 6 8 2>&1 | grep batch
echo "== quarter form up to 8 gates (SGFHE_SMALL_SPLIT=8)"; SGFHE_SMALL_SPLIT=8 python tools/latency.py 1 2 3 4 6 8 2>&1 | grep batch
done
