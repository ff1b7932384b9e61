#!/bin/bash
# LDS bank-conflict cycles of the local-operator kernel attributed to its stages: one rocprofv3 --pmc pass per PA_ABLATE
# mask (a tuning build, PA_LIB), difference against the full kernel.   tools/lds_conflicts.sh <workload> <lib> [mode]
W=$1; L=$2; M=${3:-L}
export TMPDIR=/tmp
for A in 0 1 4 16 32 64 128; do
  D=gpurun_out/ldsc/${W}_$A; rm -rf $D; mkdir -p $D
  PA_LIB=$L PA_ABLATE=$A rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $D -o p -- python3 bench.py --settle-ms 0 --steps 3 --warmup 1 --no-cpu-baseline --workload $W --mode $M > $D/bench.json 2> $D/err.log
  python3 - "$D" "$A" <<'PY'
import csv, glob, sys, collections
d, a = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hho_local_ops_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
disp = 4.0
print("ablate %4s: conflict %.3e  lds_active %.3e  lds_insts %.3e  wave_cycles %.3e  (per dispatch)" % (a, acc["SQ_LDS_BANK_CONFLICT"] / disp, acc["SQ_LDS_IDX_ACTIVE"] / disp, acc["SQ_INSTS_LDS"] / disp, acc["SQ_WAVE_CYCLES"] / disp))
PY
done
