#!/bin/bash
L=$PWD/proton_amd/lib/variants/rl/libproton_amd.so
PA_LIB=$L python -m pytest tests/test_gpu_condensed.py -x -q 2>&1 | tail -2
bash tools/ab.sh "main rl" "quad1024_k2 quad1024_k1 quad1024_k3" "C" 20 2 > gpurun_out/ab_rl.log 2>&1; tail -7 gpurun_out/ab_rl.log
bash tools/lds_conflicts.sh quad1024_k2 $L L 2>&1 | tee gpurun_out/ldsc_k2.log
bash tools/lds_conflicts.sh quad1024_k3 $L L 2>&1 | tee gpurun_out/ldsc_k3.log
