#!/bin/bash
# Is the block-sparse mean-shift kernel power-limited? Samples rocm-smi (power, clocks) while tools/sparse_ab.py loops the kernel.
# tools/power_probe.sh [form]
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^=\|^$" | head -20
python tools/sparse_ab.py warm ${1:-0} > /dev/null 2>&1
(for i in 1 2 3 4 5 6 7 8; do python tools/sparse_ab.py loop ${1:-0} > /dev/null 2>&1; done) &
PID=$!
sleep 6
for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ' ' | cut -c1-400; echo; sleep 0.7; done
wait $PID
