B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("RESULT", sys.argv[1], round(d["value"],1), "ext_us", round(r["launch_ms"]*1e3,1), "crt_us", round(r["pair_launch_ms"]*1e3,1))'
for i in 1 2; do
for v in base maxilp maxmem bias100; do
SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so $B | python -c "$P" ${v}_$i
done
done
