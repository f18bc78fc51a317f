# Two lanes with the integer-only CRT: chunk sweep, and k_extprod at issue priority 2.
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
for i in 1 2; do
$B | python tools/result_line.py lean_l1_c512_$i
for c in 128 192 256 320 384; do
$B --lanes 2 --chunk $c | python tools/result_line.py lean_l2_c${c}_$i
done
SGFHE_HIP_LIB=$PWD/tools/abl/lib_prio2.so $B --lanes 2 --chunk 256 | python tools/result_line.py prio2_l2_c256_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_prio2.so $B | python tools/result_line.py prio2_l1_c512_$i
done
./tools/ubench_xchg
