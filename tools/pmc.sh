#!/bin/bash
# HBM traffic + SQ counters of one bench workload, in SEPARATE --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share
# one on gfx950; MI355X_MICROARCH.md), reduced to profiles-style JSON by tools/pmc_reduce.py:
#   tools/pmc.sh <workload> <out.json>
W=$1; OUT=$2
export TMPDIR=/tmp
D=gpurun_out/pmc_$W
rm -rf $D; mkdir -p $D
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/pass$i -o p -- python3 bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --workload $W > $D/pass$i.json 2> $D/pass$i.err || echo "pass $i failed"
  echo "pass $i ($C) done"
done
python3 tools/pmc_reduce.py $D $W > $OUT && cat $OUT
