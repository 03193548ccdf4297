#!/bin/bash
# clocks and power while the conv stack runs (rocm-smi sampled beside tools/run_chunks.py): is the matrix-pipe peak priced at a clock the part sustains?
prec=${1:-f16x2}; reps=${2:-1200}
python tools/run_chunks.py $prec 1005 $reps > /dev/null 2>&1 &
pid=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | head -8
  echo "--"
  sleep 2
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
true
