B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("RESULT", sys.argv[1], round(d["value"],1), "ext_us", round(r["launch_ms"]*1e3,1), "crt_us", round(r["pair_launch_ms"]*1e3,1))'
for i in 1 2; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_sfull.so $B | python -c "$P" sfull_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_off3p.so $B | python -c "$P" off3p_$i
$B | python -c "$P" digitloads_$i
done
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "params1024_vs_oracle or params64_bootstrap or small_synthetic" 2>&1 | tail -2
