#!/bin/bash
# Clock and package power (rocm-smi samples) while a workload loops on the GPU box:  tools/power_probe.sh <command ...>
# e.g. tools/power_probe.sh ./gpurun_in/mfma_power_probe 8      tools/power_probe.sh python3 tools/bench_conv_gn.py 32 512 256 128 0 60000
"$@" > /tmp/power_probe_cmd.log 2>&1 &
PID=$!
sleep 5
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Package Power" | sed 's/GPU\[0\]\s*: //' | tr '\n' ' '; echo
  sleep 2
done
cat /tmp/power_probe_cmd.log | grep -v amdgpu.ids | tail -4
