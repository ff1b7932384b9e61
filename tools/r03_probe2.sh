cd $GRAFT_REPO_ROOT
timeout -k 5 200 python tools/r03_asm_probe.py 2>&1 | tail -30
timeout -k 10 400 python -m pytest tests/test_gpu_obstacle.py tests/test_gpu_parity.py -x -q -k "not full_size" > gpurun_out/r03_small_tests.log 2>&1; tail -3 gpurun_out/r03_small_tests.log
bash tools/ab.sh "base main small2" "obstacle512_k1" "L" 20 3 > gpurun_out/r03_ab_small.log 2>&1; tail -5 gpurun_out/r03_ab_small.log
