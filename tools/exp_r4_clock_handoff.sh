# Is the residue hand-off's cost a matter of power?  Clock and socket power sampled (rocm-smi) during the default
# workload: the build, and timing-only builds (wrong results) without k_extprod's residue stores
# (-DSGFHE_ABL_NO_YRES), without k_crt_lean's residue loads (-DSGFHE_ABL_NO_YLOAD), without both.
#   (cd sgfhe.jl_amd/csrc && for f in NO_YRES NO_YLOAD; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSGFHE_ABL_$f -shared -o ../../tools/abl/lib_...so engine.hip; done)
probe() {
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-io --no-live-counters > gpurun_out/clock_probe_$1.json 2>/dev/null &
  BP=$!
  sleep 9
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/GPU\[0\]\t\t: //; s/clock level: 1: //; s/Current Socket Graphics Package //' | tr '\n' ' '; echo
    sleep 1.5
  done
  wait $BP
  python tools/result_line.py $1 < gpurun_out/clock_probe_$1.json
}
for i in 1 2; do
echo "== the build"; probe base_$i
for v in no_yres no_yload no_both; do echo "== $v"; SGFHE_HIP_LIB=$PWD/tools/abl/lib_$v.so probe ${v}_$i; done
done
