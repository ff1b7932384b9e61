#!/bin/bash
# GPU suite + every bench workload at the current tree; summary to stdout
python -m pytest tests -m gpu -x -q > gpurun_out/gputest_head.log 2>&1
tail -3 gpurun_out/gputest_head.log
bash tools/bench_all.sh && python - <<'PY'
import json
for l in open("gpurun_out/bench_all.jsonl"):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:80]); continue
    print(d["config"]["workload"], d["config"].get("mode"), d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
