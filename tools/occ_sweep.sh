#!/bin/bash
# kernel time against resident blocks per CU (PA_BLOCKS_PER_CU): is the kernel latency-bound (time ~ 1/blocks) or
# bound by a shared pipe (saturates)?   tools/occ_sweep.sh <workload> <lib.so> blocks...
W=$1; L=$2; shift 2
for B in "$@"; do PA_LIB=$L PA_BLOCKS_PER_CU=$B timeout -k 10 100 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('blocks/CU',$B,'kern_ms %.3f'%r['roofline']['kernel_ms'])"; done
