#!/bin/bash
# A/B timing of kernel builds in ONE session (interleaved rounds): tools/ab.sh <workload> <rounds> libA.so libB.so ...
W=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for L in "$@"; do
    PA_LIB=$L timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$L', 'round', $r, 'kern_ms %.3f'%r['roofline']['kernel_ms'])"
  done
done
