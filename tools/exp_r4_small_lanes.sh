# A call of a few gates as two halves on the two lanes (default up to 24 gates) against one chunk on one stream
# (SGFHE_SMALL_LANES=0).  Params(1024), same call, alternating.
for i in 1 2; do
echo "== one chunk, one stream (SGFHE_SMALL_LANES=0)"; SGFHE_SMALL_LANES=0 python tools/latency.py 1 2 3 4 6 8 10 12 14 16 20 24 32 2>&1 | grep batch
echo "== two halves on the two lanes"; python tools/latency.py 1 2 3 4 6 8 10 12 14 16 20 24 32 2>&1 | grep batch
done
