#!/bin/bash
# A/B of library builds:  tools/ab.sh "<tag> ..." "<workload> ..." "<modes>" [steps] [rounds]   ("main" = the shipped library)
# prints every run and, at the end, the minimum kernel time per (tag, workload, mode)
TAGS=$1; WL=$2; MODES=${3:-L}; K=${4:-20}; R=${5:-3}
LOG=$(mktemp)
for r in $(seq $R); do
for T in $TAGS; do
  if [ "$T" = main ]; then unset PA_LIB; else export PA_LIB=$PWD/proton_amd/lib/variants/$T/libproton_amd.so; fi
  for M in $MODES; do for W in $WL; do
    timeout -k 10 200 python bench.py --workload $W --mode $M --steps $K --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | T=$T python -c "
import sys, json, os
d = json.loads(sys.stdin.read())
print('%-10s %-22s %s step %.3f ms kernel %.3f ms' % (os.environ['T'], d['config']['workload'], d['config'].get('mode', '?'), d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $LOG
  done; done
done
done
echo "== minimum over $R rounds"
python3 - $LOG <<'PY'
import sys, collections
best = collections.OrderedDict()
for ln in open(sys.argv[1]):
    f = ln.split()
    if len(f) < 9: continue
    key = (f[0], f[1], f[2]); k = float(f[7])
    best[key] = min(best.get(key, 1e9), k)
for (t, w, m), k in best.items(): print('%-10s %-22s %s kernel %.3f ms' % (t, w, m, k))
PY
