#!/bin/bash
# every bench workload once (mode L; the tensor workloads also in mode C), JSON lines to gpurun_out/bench_all.jsonl
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out/bench_all.jsonl
: > $OUT
for W in quad1024_k2 quad1024_k2_general quad1024_k1 quad1024_k3 quad256_k1_fan quad512_k2_fan obstacle512_k1 cuthho512_k2 cuthho512_k2_interface quad2048_k3; do
  timeout -k 10 300 python bench.py --workload $W --mode L --steps 10 --warmup 3 2>/dev/null | tail -1 >> $OUT
  echo "$W L done"
done
for W in quad1024_k2 quad1024_k2_general quad1024_k1 quad1024_k3 obstacle512_k1 cuthho512_k2 quad2048_k3; do
  timeout -k 10 300 python bench.py --workload $W --mode C --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT
  echo "$W C done"
done
