cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_cuthho.py -q -x -s > gpurun_out/r03_cut_dd_tests.log 2>&1; tail -2 gpurun_out/r03_cut_dd_tests.log; grep "config 3" gpurun_out/r03_cut_dd_tests.log | cut -c1-260
timeout -k 5 200 python bench.py --workload cuthho512_k2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 3', 'step %.3f' % d['ms_per_step'], d['stage_ms'], 'frac %.3f' % d['roofline']['frac'])"
export PA_LIB=$PWD/proton_amd/lib/variants/cutab/libproton_amd.so
PA_CUT_CLOCK=1 timeout -k 5 200 python bench.py --workload cuthho512_k2 --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline 2>&1 | grep PA_CUT_CLOCK | tail -2
