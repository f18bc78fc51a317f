# Calls of 25 to 256 gates at Params(1024): k_extprod with 16 points per thread (512 threads, the build) against 8
# (1024 threads per (gate, prime): four waves per SIMD on a compute unit the workgroup has to itself).
# variant library (build here; tools/abl/ is git-ignored):
#   (cd sgfhe.jl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSGFHE_EXT_LE3_MAX=13 -shared -o ../../tools/abl/lib_le3_13.so engine.hip)
S="25 32 40 48 51 56 64 80 96 128 256"
for i in 1 2; do
echo "== 16 points per thread"; python tools/latency.py $S 2>&1 | grep batch
echo "== 8 points per thread"; SGFHE_HIP_LIB=$PWD/tools/abl/lib_le3_13.so python tools/latency.py $S 2>&1 | grep batch
done
