# The one-launch-per-iteration prototype (kernels.h k_iter_all: all primes of a bootstrap in one workgroup, residues in
# registers / LDS, CRT in the same kernel) against the build.  Variant library (build here; tools/abl/ is git-ignored):
#   (cd sgfhe.jl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSGFHE_WITH_ITER_ALL -shared -o ../../tools/abl/lib_iter_all.so engine.hip)
export SGFHE_HIP_LIB=$PWD/tools/abl/lib_iter_all.so
SGFHE_ITER_ALL=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "params1024_vs_oracle" 2>&1 | tail -1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-live-counters"
for i in 1 2; do
$B | python tools/result_line.py two_kernels_$i
SGFHE_ITER_ALL=1 $B | python tools/result_line.py iter_all_$i
SGFHE_ITER_ALL=1 $B --lanes 1 --chunk 256 | python tools/result_line.py iter_all_one_lane_256_$i
done
