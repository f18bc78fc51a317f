# Two lanes: non-temporal loads in k_crt_lean (residues; residues + old digits) by chunk size.
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2; do
for c in 256 192 224; do
for v in base ntload ntboth; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B --chunk $c | python tools/result_line.py ${v}_c${c}_$i
done
done
done
