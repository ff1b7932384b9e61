# end-of-round evidence in one call: the driver's bench command, every workload and mode, the GPU suite
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_default_line.json 2> gpurun_out/r03_bench_default_line.err; echo "default rc=$?"
bash tools/bench_all.sh > gpurun_out/r03_bench_all.log 2>&1; cp gpurun_out/bench_all.jsonl gpurun_out/r03_bench_all_workloads.jsonl
for spec in quad1024_k2:A quad1024_k1:A quad1024_k3:A; do W=${spec%%:*}; timeout -k 10 300 python bench.py --workload $W --mode A --steps 10 --warmup 3 2>/dev/null | tail -1 >> gpurun_out/r03_bench_all_workloads.jsonl; echo "$spec done"; done
python - <<'PY'
import json
for ln in open("gpurun_out/r03_bench_all_workloads.jsonl"):
    try: d = json.loads(ln)
    except Exception: continue
    print("%-22s %s step %.3f ms  kernel %.3f  frac %.3f  %.0f Mcells/s  traffic %s" % (d["config"]["workload"], d["config"]["mode"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"] / 1e6, d["roofline"]["traffic"]))
PY
