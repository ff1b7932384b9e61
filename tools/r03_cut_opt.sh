cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_cuthho.py -m gpu -x -q > gpurun_out/r03_cut_opt_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03_cut_opt_tests.log
grep "cut cells" gpurun_out/r03_cut_opt_tests.log | tail -3
timeout -k 10 200 python tools/r03_cut_clock.py 512 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I proton_amd/csrc -o /tmp/dd_check tools/probe/dd_check.hip && /tmp/dd_check && python tools/probe/dd_check.py
PA_LIB=$PWD/proton_amd/lib/variants/cut/libproton_amd.so PA_CUT_CLOCK=1 timeout -k 10 200 python tools/r03_cut_clock.py 512 2 2>&1 | grep -m2 "PA_CUT_CLOCK\|cut kernel"
PA_LIB=$PWD/proton_amd/lib/variants/cut/libproton_amd.so PA_CUT_CLOCK=1 timeout -k 10 200 python tools/r03_cut_clock.py 512 2 2>&1 | grep "cut kernel"
