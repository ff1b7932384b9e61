cd $GRAFT_REPO_ROOT
for T in main asm1 asm4; do
  if [ "$T" = main ]; then unset PA_LIB; else export PA_LIB=$PWD/proton_amd/lib/variants/$T/libproton_amd.so; fi
  for W in quad1024_k2 quad1024_k3; do
  timeout -k 5 200 python bench.py --workload $W --mode A --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T $W', 'step %.3f' % d['ms_per_step'], d['stage_ms'])"
  done
done
unset PA_LIB
timeout -k 10 300 python -m pytest tests/test_gpu_assembler.py -x -q 2>&1 | tail -2
