# Two lanes: cache policy of the residue hand-off.  k_crt_lean's residue loads non-temporal (read
# once), k_extprod's residue stores plain / non-temporal instead of write-through (sc1).
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2; do
for v in base ntload yres_plain yres_nt ntload_plain; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B | python tools/result_line.py ${v}_$i
done
SGFHE_HIP_LIB=$PWD/tools/abl/lib_base.so $B --chunk 192 | python tools/result_line.py base_c192_$i
done
