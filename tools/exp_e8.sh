set -x
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("RESULT", sys.argv[1], round(d["value"],1), "ext_us", round(r["launch_ms"]*1e3,1), "crt_us", round(r["pair_launch_ms"]*1e3,1), "chunk", d["config"]["chunk"])'
$B | python -c "$P" e16_c512
SGFHE_HIP_LIB=$PWD/tools/abl/lib_e8.so $B | python -c "$P" e8_c512
SGFHE_HIP_LIB=$PWD/tools/abl/lib_e8nk.so $B | python -c "$P" e8nk_c512
SGFHE_HIP_LIB=$PWD/tools/abl/lib_e8.so $B --chunk 1024 | python -c "$P" e8_c1024
SGFHE_HIP_LIB=$PWD/tools/abl/lib_e8.so python -m pytest tests/test_gpu_parity.py -q -x -k "params1024_vs_oracle or small_synthetic or ntt_matches" 2>&1 | tail -3
