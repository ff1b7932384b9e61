#!/bin/bash
# the N > 1 code path of bench.py with two ranks sharing the visible GPU (host-staged exchange: numbers not comparable)
export MASTER_ADDR=127.0.0.1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --workload ${1:-quad1024_k2} 2> gpurun_out/rehearse_n2.err | tail -1 > gpurun_out/rehearse_n2.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/rehearse_n2.json").read())
print("n_gpus", d["n_gpus"], d["config"]["workload"], d["config"]["mode"], "step %.3f ms" % d["ms_per_step"], "settle", d["settle"]["passes"], "exchange:", d["config"]["exchange"][:90])
print("same_step_one_gpu", d["same_step_one_gpu"], "exchange_checked", d.get("exchange_checked"), "stage_ms", d["stage_ms"])
PY
