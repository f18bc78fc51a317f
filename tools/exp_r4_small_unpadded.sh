# Small-batch form of the k-loop with its grids sized by the gates actually in the call (default) against
# the padding to a multiple of 8 bootstraps of rounds 1-3 (SGFHE_SMALL_PADDED=1: a one-gate call launched
# eight gates' workgroups).  Same call, alternating.
for i in 1 2; do
echo "== padded to 8 (rounds 1-3)"; SGFHE_SMALL_PADDED=1 python tools/latency.py 1 2 4 8 12 16 24 2>&1 | grep batch
echo "== unpadded"; python tools/latency.py 1 2 4 8 12 16 24 2>&1 | grep batch
done
