# Which latency form for which call size: Params(1024) and Params(512), gates 1 ... 24.
#   A  two transform launches (SGFHE_SMALL_FUSED=0), two lanes from 8 gates   (the state before k_ext_quarter)
#   B  fused                                                              C  A on one stream (SGFHE_SMALL_LANES=0)
#   D  fused, one stream, quarter form up to 24 gates                     E  fused, two lanes, quarter form up to 12 per lane
S="1 2 3 4 5 6 7 8 10 12 14 16 20 24"
for N in 1024 512; do
export SGFHE_LATENCY_N=$N
echo "== Params($N) A"; SGFHE_SMALL_FUSED=0 python tools/latency.py $S 2>&1 | grep batch
echo "== Params($N) B"; SGFHE_SMALL_FUSED=1 python tools/latency.py $S 2>&1 | grep batch
echo "== Params($N) C"; SGFHE_SMALL_FUSED=0 SGFHE_SMALL_LANES=0 python tools/latency.py $S 2>&1 | grep batch
echo "== Params($N) D"; SGFHE_SMALL_FUSED=1 SGFHE_SMALL_LANES=0 SGFHE_SMALL_SPLIT=24 python tools/latency.py $S 2>&1 | grep batch
echo "== Params($N) E"; SGFHE_SMALL_FUSED=1 SGFHE_SMALL_SPLIT=12 python tools/latency.py $S 2>&1 | grep batch
done
