#!/bin/bash
# Samples the GPU's shader clock / power while bench.py runs a long timed region.
python3 bench.py --steps 600 --warmup 5 --no-cpu > /tmp/clk_bench.json 2>/dev/null &
pid=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -3
  sleep 0.4
done
wait $pid
python3 -c "import json; d=json.load(open('/tmp/clk_bench.json')); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['perm_kernel']['avg_launch_ms'])"
echo idle; sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -3
