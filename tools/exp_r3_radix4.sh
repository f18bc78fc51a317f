# Forward transform with radix-4 steps and deferred reductions (ntt.h fwd_step4) against the radix-2 build.
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2 3; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_base.so $B | python tools/result_line.py base_$i
$B | python tools/result_line.py radix4_$i
done
SGFHE_HIP_LIB=$PWD/tools/abl/lib_base.so $B --lanes 1 | python tools/result_line.py base_l1
$B --lanes 1 | python tools/result_line.py radix4_l1
