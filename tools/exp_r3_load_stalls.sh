# Upper bound on what the digit and key loads of k_extprod cost (their latency at the head of each
# of the four phases): timing-only builds with the loads replaced by register arithmetic
# (-DSGFHE_ABL_NO_DIG, -DSGFHE_ABL_NO_KEY; wrong results), same call.
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for i in 1 2; do
$B | python tools/result_line.py base_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_NO_DIG.so $B | python tools/result_line.py no_digit_loads_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_NO_KEY.so $B | python tools/result_line.py no_key_loads_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_NO_DIG_KEY.so $B | python tools/result_line.py neither_$i
done
$B --lanes 1 | python tools/result_line.py base_one_lane
SGFHE_HIP_LIB=$PWD/tools/abl/lib_NO_DIG_KEY.so $B --lanes 1 | python tools/result_line.py neither_one_lane
