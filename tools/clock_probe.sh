# Sample GPU clock / power / temperature while the default workload runs (explains box-to-box spread).
python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated > gpurun_out/clock_probe_bench.json 2>/dev/null &
BP=$!
sleep 8
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)|fclk" | head -8
  echo "--"
  sleep 2
done
wait $BP
cut -c1-90 gpurun_out/clock_probe_bench.json
