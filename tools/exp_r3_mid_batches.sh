# Batches between the small-batch threshold and two default chunks: automatic split into an even
# number of equal chunks (two lanes) against one lane (SGFHE lanes 1 = the round-2 schedule).
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-io --no-isolated"
for b in 32 48 64 100 128 256 384 640; do
$B --batch $b --lanes 1 | python tools/result_line.py b${b}_one_lane
$B --batch $b | python tools/result_line.py b${b}_auto
done
