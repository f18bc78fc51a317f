# Rotation epilogue of k_extprod with both product polynomials in the plain LDS layout (no bank
# conflict for any j; one more workgroup barrier) against round 3's swizzled layout
# (-DSGFHE_EPI_SWIZZLED: 2.6 % of the kernel's LDS cycles were bank conflicts).  Same call, alternating.
# (as run: the plain layout was the default build and -DSGFHE_EPI_SWIZZLED the variant; the result made
#  the swizzled layout the default again, the plain one is now -DSGFHE_EPI_PLAIN)
#   make -C sgfhe.jl_amd/csrc -B EXTRA="-DSGFHE_EPI_SWIZZLED" OUT=../../tools/abl/lib_epi_swz.so && make -C sgfhe.jl_amd/csrc
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2 3; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_epi_swz.so $B | python tools/result_line.py swizzled_$i
$B | python tools/result_line.py plain_$i
done
SGFHE_HIP_LIB=$PWD/tools/abl/lib_epi_swz.so $B --lanes 1 | python tools/result_line.py swizzled_l1
$B --lanes 1 | python tools/result_line.py plain_l1
for cfg in params512 params2048; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_epi_swz.so $B --config $cfg --batch 1024 | python tools/result_line.py swizzled_$cfg
$B --config $cfg --batch 1024 | python tools/result_line.py plain_$cfg
done
