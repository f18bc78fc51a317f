# Two lanes, on top of the non-temporal residue loads: cache policy of k_crt_lean's digit stores and
# of k_extprod's digit loads / residue stores.
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2; do
for v in base crtst_nt crtst_sc1 digld_sc1 digld_nt yres_sc0sc1; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B | python tools/result_line.py ${v}_$i
done
done
