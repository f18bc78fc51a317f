# Randomised flatten: k_crt_lean_rnd (integer-only CRT with the draws folded into the limb sums)
# against k_crt_acc (SGFHE_CRT_LEAN=0), same call.  Parity first.
set -e
python -m pytest tests/test_gpu_random.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -k "random" 2>&1 | tail -3
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated --flatten random"
SGFHE_CRT_LEAN=0 $B | python tools/result_line.py p1024_random_k_crt_acc
$B | python tools/result_line.py p1024_random_lean
SGFHE_CRT_LEAN=0 $B | python tools/result_line.py p1024_random_k_crt_acc
$B | python tools/result_line.py p1024_random_lean
B2="$B --config params2048 --batch 1024 --steps 2"
SGFHE_CRT_LEAN=0 $B2 | python tools/result_line.py p2048_random_k_crt_acc
$B2 | python tools/result_line.py p2048_random_lean
B3="$B --config params512 --batch 4096"
SGFHE_CRT_LEAN=0 $B3 | python tools/result_line.py p512_random_k_crt_acc
$B3 | python tools/result_line.py p512_random_lean
