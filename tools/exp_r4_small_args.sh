# Latency form of the k-loop (1-24 gates per call): the per-prime constants of k_fwd_phase / k_inv_column
# passed by value in the kernel arguments (default) against the record pointer of rounds 1-3
# (-DSGFHE_SMALL_PS_PTR): one dependent memory round trip less at the head of each of 2 x n launches.
#   make -C sgfhe.jl_amd/csrc -B EXTRA="-DSGFHE_SMALL_PS_PTR" OUT=../../tools/abl/lib_small_ptr.so
for i in 1 2; do
echo "== record pointer (rounds 1-3)"; SGFHE_HIP_LIB=$PWD/tools/abl/lib_small_ptr.so python tools/latency.py 1 8 16 24 2>&1 | grep batch
echo "== by value"; python tools/latency.py 1 8 16 24 2>&1 | grep batch
done
