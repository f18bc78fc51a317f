# k_crt_lean (integer-only CRT, 4 coefficients per thread) against k_crt_acc2 (SGFHE_CRT_LEAN=0 in the environment), one and two lanes.
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
for i in 1 2; do
SGFHE_CRT_LEAN=0 $B | python tools/result_line.py acc2_l1_c512_$i
$B | python tools/result_line.py lean_l1_c512_$i
SGFHE_CRT_LEAN=0 $B --lanes 2 --chunk 256 | python tools/result_line.py acc2_l2_c256_$i
$B --lanes 2 --chunk 256 | python tools/result_line.py lean_l2_c256_$i
$B --lanes 2 --chunk 512 | python tools/result_line.py lean_l2_c512_$i
done
