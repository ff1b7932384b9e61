#!/bin/bash
# profiling aid: time the dominant kernel with single stages skipped (PA_ABLATE bit i = stage i)
W=${1:-quad1024_k2}
for A in 0 1 2 3 4 8 16 32 64 128 129 255; do PA_ABLATE=$A timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ablate',$A,'kern_ms %.3f'%r['roofline']['kernel_ms'])"; done
