# two-stream overlap of k_crt_acc with k_extprod: does a 104-VGPR k_extprod (SGFHE_ACC0_32) leave
# room for the CRT waves on the same SIMDs?   usage (GPU box): bash tools/exp_overlap.sh
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
# variant library (build here, it travels with gpurun; tools/abl/ is git-ignored):
#   mkdir -p tools/abl && (cd sgfhe.jl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSGFHE_ACC0_32 -shared -o ../../tools/abl/lib_acc32.so engine.hip)
for i in 1 2; do
$B --lanes 1 | python tools/result_line.py acc64_lanes1_$i
$B --lanes 2 | python tools/result_line.py acc64_lanes2_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_acc32.so $B --lanes 1 | python tools/result_line.py acc32_lanes1_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_acc32.so $B --lanes 2 | python tools/result_line.py acc32_lanes2_$i
done
