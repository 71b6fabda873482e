#!/bin/bash
# one line per process: its steady ms per cycle and the clocks rocm-smi shows while it runs (sclk / mclk / fclk / socclk, power)
for i in $(seq 1 ${1:-8}); do
  python tools/process_probe.py > /tmp/pp_$i.log 2>&1 &
  pid=$!
  sleep 2.3
  clk=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | sed 's/.*: //' | tr '\n' ' ')
  wait $pid
  echo "$(grep steady /tmp/pp_$i.log | sed 's/steady ms per cycle //') | $clk"
done
