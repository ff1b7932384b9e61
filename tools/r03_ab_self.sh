set -e
cd $GRAFT_REPO_ROOT
PA_LIB=$PWD/proton_amd/lib/variants/self/libproton_amd.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "not full_size" > gpurun_out/r03_self_parity.log 2>&1 || true
tail -3 gpurun_out/r03_self_parity.log
bash tools/ab.sh "base main self" "quad1024_k2 quad1024_k1" "L C" 20 3 > gpurun_out/r03_ab_self.log 2>&1
tail -14 gpurun_out/r03_ab_self.log
