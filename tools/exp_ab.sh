# Same-box A/B of two builds of libsgfhe_hip.so at Params(1024), batch 4096 (3 steps each, twice):
#   cp sgfhe.jl_amd/csrc/libsgfhe_hip.so tools/abl/lib_base.so     # before the change
#   make -C sgfhe.jl_amd/csrc                                        # after the change
#   gpurun -- 'bash tools/exp_ab.sh > gpurun_out/exp_ab.txt 2>&1'
# (boxes differ by 2-3 %: only numbers from one call are comparable; tools/abl/ is git-ignored)
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
for i in 1 2; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_base.so $B | python tools/result_line.py base_$i
$B | python tools/result_line.py new_$i
done
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
