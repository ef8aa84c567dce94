#!/bin/bash
# Samples socket power and clocks (rocm-smi) while the single-launch forward runs in a loop.
python3 tools_prof/ab_mega.py base > gpurun_out/power_ab.log 2>&1 &
PID=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
tail -2 gpurun_out/power_ab.log
