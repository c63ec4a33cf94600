#!/bin/bash
# power / clock samples while one kernel family loops (scripts/kbench.py --kernels $1 --blocks $2 --iters 30000)
python scripts/kbench.py --kernels $1 --blocks $2 --iters ${3:-30000} > /tmp/kb.log 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo
  sleep 1.5
done
kill $PID 2>/dev/null
wait $PID 2>/dev/null
grep "'k'" /tmp/kb.log | head -3
