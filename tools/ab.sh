#!/bin/bash
# A/B of library builds:  tools/ab.sh "<tag> ..." "<workload> ..." "<modes>" [steps]     ("main" = the shipped library)
TAGS=$1; WL=$2; MODES=${3:-L}; K=${4:-20}
for R in 1 2; do
for T in $TAGS; do
  if [ "$T" = main ]; then unset PA_LIB; else export PA_LIB=$PWD/proton_amd/lib/variants/$T/libproton_amd.so; fi
  for M in $MODES; do for W in $WL; do
    timeout -k 10 200 python bench.py --workload $W --mode $M --steps $K --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | T=$T python -c "
import sys, json, os
d = json.loads(sys.stdin.read())
print('%-10s %-22s %s step %.3f ms kernel %.3f ms' % (os.environ['T'], d['config']['workload'], d['config'].get('mode', '?'), d['ms_per_step'], d['roofline']['kernel_ms']))"
  done; done
done
done
