cd $GRAFT_REPO_ROOT
timeout -k 5 120 python bench.py --workload quad256_k1_fan --mode A --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03_A_small.json 2> gpurun_out/r03_A_small.err; echo rc=$?; tail -c 600 gpurun_out/r03_A_small.err; head -c 600 gpurun_out/r03_A_small.json; echo
timeout -k 5 200 python bench.py --workload quad1024_k2 --mode A --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03_A_1024.json 2> gpurun_out/r03_A_1024.err; echo rc=$?; tail -c 600 gpurun_out/r03_A_1024.err; head -c 300 gpurun_out/r03_A_1024.json; echo
