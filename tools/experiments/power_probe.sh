#!/bin/bash
# sample rocm-smi while a command runs
"$@" > /tmp/cmd.out 2>&1 &
pid=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Performance" | tr '\n' ';'; echo
  sleep 0.5
done
wait $pid
tail -2 /tmp/cmd.out | cut -c1-300
