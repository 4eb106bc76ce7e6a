#!/bin/bash
# developer: engine clock / power while the bench workload runs (is the chip at its nominal 2.4 GHz under this kernel?)
python bench.py --steps 12000 --warmup 5 --cpu-budget 0 --no-extras --no-latency --no-check --precision f16_split > /dev/null 2>&1 &
BP=$!
sleep 35
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
echo idle:; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo
