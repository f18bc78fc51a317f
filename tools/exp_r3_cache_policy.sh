# Two lanes: cache policy of the residue hand-off (profiles/r03_exp_cache_policy.txt, call 1).
# Variant libraries (csrc/Makefile, EXTRA flags become part of sgfhe_build_id()):
#   make -C sgfhe.jl_amd/csrc -B OUT=../../tools/abl/lib_plain.so      EXTRA=-DSGFHE_CRT_PLAIN_LOADS   # k_crt_lean with plain residue loads (the build before the change)
#   make -C sgfhe.jl_amd/csrc -B OUT=../../tools/abl/lib_yres_plain.so EXTRA=-DSGFHE_YRES_AUX=0        # k_extprod residue stores plain instead of sc1
#   make -C sgfhe.jl_amd/csrc -B OUT=../../tools/abl/lib_yres_nt.so    EXTRA=-DSGFHE_YRES_AUX=2        # ... non-temporal
# (The other variants of that file -- non-temporal old-digit loads, digit-store and digit-load
# policies, the 16-byte epilogue, ds_add_u32 -- were one-off patches that are not kept in the tree;
# what they changed is described in the profiles/r03_exp_*.txt files.)
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2; do
$B | python tools/result_line.py ntload_$i
for v in plain yres_plain yres_nt; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B | python tools/result_line.py ${v}_$i
done
$B --chunk 192 | python tools/result_line.py ntload_c192_$i
done
