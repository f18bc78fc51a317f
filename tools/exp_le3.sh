P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("RESULT", sys.argv[1], round(d["value"],1), "ext_us", round(r["launch_ms"]*1e3,1), "crt_us", round(r["pair_launch_ms"]*1e3,1))'
for i in 1 2; do
python bench.py --config params512 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-host-io | python -c "$P" p512_le4_$i
SGFHE_HIP_LIB=$PWD/tools/abl/lib_le3_12.so python bench.py --config params512 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-host-io | python -c "$P" p512_le3_$i
done
