# Radix-4 steps at m = 4096 (Params(512)) and m = 16384 (Params(2048)) where they spill 16-36 bytes
# per thread: all steps (lib_r4all), the inverse only (lib_r4inv), against the default (none at
# m = 4096 in the inverse, no per-lane forward step at m = 4096 / 16384).  Same call.
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated --config params512 --batch 4096"
for i in 1 2 3; do
$B | python tools/result_line.py p512_default_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_r4all.so $B | python tools/result_line.py p512_all_steps_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_r4inv.so $B | python tools/result_line.py p512_inverse_steps_$i
done
B2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated --config params2048 --batch 1024"
$B2 | python tools/result_line.py p2048_default
SGFHE_HIP_LIB=$PWD/tools/abl/lib_r4all.so $B2 | python tools/result_line.py p2048_all_steps
