#!/bin/bash
# Evidence for profiles/: per bench workload and mode the rocprofv3 kernel stats (one run) and the HBM counters (FETCH_SIZE
# and WRITE_SIZE in separate --pmc passes, gfx950: they cannot share one); SQ counters for the headline; the auxiliary
# kernels once.  Run on the GPU box from the repo root:   tools/profile_round.sh r02 [spec ...]   spec = workload:mode
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
SPECS="$@"
[ -z "$SPECS" ] && SPECS="quad1024_k2:L quad1024_k2:C quad1024_k2:A quad1024_k1:L quad1024_k1:C quad1024_k3:L quad1024_k3:C obstacle512_k1:L obstacle512_k1:C quad2048_k3:L quad2048_k3:C quad256_k1_fan:L quad512_k2_fan:L cuthho512_k2:L cuthho512_k2:C cuthho512_k2_interface:L quad1024_k2_general:L quad1024_k2_general:C"
for spec in $SPECS; do
  W=${spec%%:*}; M=${spec##*:}
  D=$OUT/${W}_$M
  rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W --mode $M > $D/bench.json 2> $D/stats.err || echo "stats run failed: $spec"
  cp $(find $D/stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kstats_${W}_$M.csv 2>/dev/null
  i=0
  PASSES=("FETCH_SIZE" "WRITE_SIZE")
  if [ "$W" = "quad1024_k2" ] || [ "$W" = "quad1024_k3" ] && [ -z "$PA_PROFILE_NO_SQ" ]; then
    PASSES+=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS")
  fi
  for C in "${PASSES[@]}"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/pass$i -o p -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --workload $W --mode $M > $D/pass$i.json 2> $D/pass$i.err || echo "pmc pass $i failed: $spec"
  done
  python3 tools/profile_reduce.py $D $W $M $D/bench.json $OUT/${TAG}_kstats_${W}_$M.csv > $OUT/${TAG}_pmc_${W}_$M.json
  cp $D/bench.json $OUT/${TAG}_bench_under_rocprof_${W}_$M.json
  echo "$spec done: $(python3 -c "import json;r=json.load(open('$OUT/${TAG}_pmc_${W}_$M.json'));print('hbm bytes %.3g, alg %.3g, kernel_ms events %s rocprof %s'%(r['hbm_bytes_per_launch_dominant_kernel'], r.get('algorithmic_bytes_per_launch',0), r.get('bench_kernel_ms_under_profiler'), (r.get('rocprof_kernel_stats') or {}).get('dominant_kernel_ms_per_step')))")"
done
if [ -z "$PA_PROFILE_NO_AUX" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/aux -o p -- python3 tools/profile_aux.py > $OUT/aux.log 2>&1 || echo "aux run failed"
cp $(find $OUT/aux -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kstats_aux.csv 2>/dev/null
fi
echo "profile_round done"
