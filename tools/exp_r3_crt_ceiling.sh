# Verdict r2 item 2: the overlap ceiling if the CRT arithmetic were free.
# Same-box A/B: base (lanes 1, chunk 512), base (lanes 2, chunk 256), and the timing-only
# -DSGFHE_ABL_CRT_MEMONLY build in both schedules (wrong results, timing only).
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
for i in 1 2; do
$B | python tools/result_line.py base_l1_c512_$i
$B --lanes 2 --chunk 256 | python tools/result_line.py base_l2_c256_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_crt_memonly.so $B | python tools/result_line.py memonly_l1_c512_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_crt_memonly.so $B --lanes 2 --chunk 256 | python tools/result_line.py memonly_l2_c256_$i
done
