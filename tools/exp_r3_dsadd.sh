# k_extprod: column-1 accumulate in LDS through ds_add_u32 (no return) instead of read / add / write.
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2 3; do
for v in base dsadd; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B | python tools/result_line.py ${v}_l2_$i
done
done
for v in base dsadd; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B --lanes 1 | python tools/result_line.py ${v}_l1
done
