#!/bin/bash
# condensed-mode kernel time for several library builds (and PA_ABLATE masks):  tools/modec_ab.sh "cases" lib[:ablate] ...
CASES=$1; shift
for spec in "$@"; do
  L=${spec%%:*}; A=""
  [[ "$spec" == *:* ]] && A=${spec##*:}
  echo "== $L ablate=${A:-0}"
  PA_LIB=$L PA_ABLATE=${A:-0} timeout -k 10 200 python tools/modec_timing.py $CASES 2>&1 | grep "^N "
done
