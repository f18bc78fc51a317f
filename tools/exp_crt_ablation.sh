B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io"
for i in 1 2; do
$B | python tools/result_line.py base_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_crt_nosd.so $B | python tools/result_line.py crt_nosd_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_crt_memonly.so $B | python tools/result_line.py crt_memonly_$i
done
