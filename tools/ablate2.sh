#!/bin/bash
# tools/ablate2.sh <workload> <lib.so> mask...  -- kernel time with the stages of each mask skipped
W=$1; L=$2; shift 2
for A in "$@"; do PA_LIB=$L PA_ABLATE=$A timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ablate',$A,'kern_ms %.3f'%r['roofline']['kernel_ms'])"; done
