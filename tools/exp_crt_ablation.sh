B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("RESULT", sys.argv[1], round(d["value"],1), "ext_us", round(r["launch_ms"]*1e3,1), "crt_us", round(r["pair_launch_ms"]*1e3,1))'
for i in 1 2; do
$B | python -c "$P" base_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_crt_nosd.so $B | python -c "$P" crt_nosd_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_crt_memonly.so $B | python -c "$P" crt_memonly_$i
done
