# Which part of the pipelined host-pointer path costs the k-loop time (round 4): SGFHE_IO_EXP bits
# 1 = no result collection in the middle of the call, 2 = all results after the last kernel,
# 4 = all inputs before the first kernel.  Same call.
for e in 0 1 2 4 6 7 0; do
echo "== SGFHE_IO_EXP=$e"
SGFHE_IO_EXP=$e SGFHE_DEBUG_IO=1 python tools/io_phases.py params1024 4096 2>&1 | grep -v "pin_\|amdgpu.ids"
done
