#!/bin/bash
# does the measured kernel time depend on the length of the run?  (clock ramp of the GPU)
for K in 5 10 20 50 200; do for W in 3 30; do
  python bench.py --workload quad1024_k2 --mode L --steps $K --warmup $W --settle-ms ${SETTLE:-0} --no-cpu-baseline 2>/dev/null | tail -1 | K=$K W=$W python -c "
import sys, json, os
d = json.loads(sys.stdin.read())
print('steps %4s warmup %3s: step %.3f ms kernel %.3f ms' % (os.environ['K'], os.environ['W'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
