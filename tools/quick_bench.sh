#!/bin/bash
# kernel time of the main workloads, one line each:  tools/quick_bench.sh [steps]   (PA_LIB selects a variant build)
K=${1:-20}
for M in L C; do
  for W in quad1024_k2 quad1024_k2_general quad1024_k1 quad1024_k3 obstacle512_k1 quad512_k2_fan; do
    timeout -k 10 200 python bench.py --workload $W --mode $M --steps $K --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-22s %s step %.3f ms kernel %.3f ms frac %.3f' % (d['config']['workload'], d['config'].get('mode', '?'), d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
