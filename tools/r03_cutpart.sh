# cut / general-quadrilateral workloads under the row partition and in condensed mode:  bash tools/r03_cutpart.sh
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_cuthho.py tests/test_gpu_condensed.py -m gpu -x -q -k "row_partition or condensed_mode or slabs_equal_whole_mesh and not config5" > gpurun_out/r03_cutpart_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03_cutpart_tests.log
for spec in cuthho512_k2:L cuthho512_k2:C quad1024_k2_general:C quad1024_k2:C; do
  W=${spec%%:*}; M=${spec##*:}
  timeout -k 10 300 python bench.py --workload $W --mode $M --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03_cp_${W}_$M.json 2> gpurun_out/r03_cp_${W}_$M.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_cp_${W}_$M.json").read().strip().splitlines()[-1])
    print("$spec", "step %.3f ms" % d["ms_per_step"], "kernel %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"], d["stage_ms"])
except Exception as e:
    print("$spec FAILED", e, open("gpurun_out/r03_cp_${W}_$M.err").read()[-800:])
PY
done
# two ranks sharing the GPU (gloo rehearsal of the N = 2 path)
for spec in cuthho512_k2:L cuthho512_k2:C quad1024_k2_general:C; do
  W=${spec%%:*}; M=${spec##*:}
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --workload $W --mode $M --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03_cp_n2_${W}_$M.json 2> gpurun_out/r03_cp_n2_${W}_$M.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_cp_n2_${W}_$M.json").read().strip().splitlines()[-1])
    print("N=2 $spec", "step %.3f ms" % d["ms_per_step"], "checked", d.get("exchange_checked"), d.get("same_step_one_gpu"), d["stage_ms"])
except Exception as e:
    print("N=2 $spec FAILED", e, open("gpurun_out/r03_cp_n2_${W}_$M.err").read()[-800:])
PY
done
