cd $GRAFT_REPO_ROOT
export PA_LIB=$PWD/proton_amd/lib/variants/ovl/libproton_amd.so
for r in 1 2 3; do
for O in 1 0; do
  for W in quad1024_k2 quad1024_k1 quad1024_k3; do
  PA_PRE_OVERLAP=$O timeout -k 5 200 python bench.py --workload $W --mode L --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$O $W', 'step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])"
  done
done
done
unset PA_LIB
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
