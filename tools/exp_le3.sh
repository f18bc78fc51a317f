# Params(512): k_extprod with 16 or 8 points per thread.   usage (GPU box): bash tools/exp_le3.sh
# variant library (build here, it travels with gpurun; tools/abl/ is git-ignored):
#   mkdir -p tools/abl && (cd sgfhe.jl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSGFHE_EXT_LE3_MAX=12 -shared -o ../../tools/abl/lib_le3_12.so engine.hip)
for i in 1 2; do
python bench.py --config params512 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-host-io | python tools/result_line.py p512_le4_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_le3_12.so python bench.py --config params512 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-host-io | python tools/result_line.py p512_le3_$i
done
