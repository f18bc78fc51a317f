B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
for i in 1 2; do
for v in base maxilp maxmem bias100; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B | python tools/result_line.py ${v}_$i
done
done
