#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of one bench workload for several library builds:
#   tools/kprof.sh <workload> lib.so ...      -> lines "<lib> <kernel> calls avg_us"
W=$1; shift
export TMPDIR=/tmp
for L in "$@"; do
  D=gpurun_out/kprof/$(basename $L .so)_$W
  rm -rf $D; mkdir -p $D
  PA_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W > $D/bench.json 2> $D/err.log
  python3 - "$D" "$L" <<'PY'
import csv, glob, sys
d, lib = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print(lib, "no kernel_stats.csv"); sys.exit(0)
for r in list(csv.DictReader(open(f[0])))[:3]:
    print("%-28s %-70s n=%s avg %.1f us" % (lib.split("/")[-1], r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
