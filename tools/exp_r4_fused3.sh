# The latency forms as shipped (two-launch quarter form up to 6 gates, k_ext_quarter for chains of 7 to 12, two lanes
# from 13 gates) against the state before k_ext_quarter (SGFHE_SMALL_FUSED=0: quarter form up to 7, two lanes from 8).
S="1 2 4 6 7 8 10 12 13 14 16 20 24"
for N in 1024 512; do
export SGFHE_LATENCY_N=$N
for i in 1 2; do
echo "== Params($N) before"; SGFHE_SMALL_FUSED=0 python tools/latency.py $S 2>&1 | grep batch
echo "== Params($N) shipped"; python tools/latency.py $S 2>&1 | grep batch
done
done
