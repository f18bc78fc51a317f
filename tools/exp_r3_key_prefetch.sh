# Small-batch form: the key rows of k_fwd_phase requested before the forward transform (default)
# against after it (-DSGFHE_FWD_KEY_LATE), same call.
for i in 1 2; do
echo "key loads after the transform"; SGFHE_HIP_LIB=$PWD/tools/abl/lib_keylate.so python tools/latency.py 1 8 16 24
echo "key loads before the transform"; python tools/latency.py 1 8 16 24
done
