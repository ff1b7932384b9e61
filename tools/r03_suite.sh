# full GPU suite + the bench lines of a round, one gpurun call:  bash tools/r03_suite.sh
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03_gpu_tests.log
for spec in quad1024_k2:A quad1024_k2:L quad1024_k1:A quad1024_k3:A; do
  W=${spec%%:*}; M=${spec##*:}
  timeout -k 10 300 python bench.py --workload $W --mode $M --steps 20 --warmup 3 > gpurun_out/r03_bench_${W}_$M.json 2> gpurun_out/r03_bench_${W}_$M.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_bench_${W}_$M.json").read().strip().splitlines()[-1])
    print("$spec", "step %.3f ms" % d["ms_per_step"], "kernel %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"], d["stage_ms"], (d.get("matrix_assembly") or {}).get("gpu_over_cpu_all_cores"))
except Exception as e:
    print("$spec FAILED", e, open("gpurun_out/r03_bench_${W}_$M.err").read()[-600:])
PY
done
