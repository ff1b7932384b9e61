#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 6 from a bench_all JSON-lines file and the reduced counter files of a profile round:
    tools/design_table.py profiles/r02f_bench_all_workloads.jsonl r02f"""
import json, os, sys
lines, tag = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
print("| Workload | mode | step | kernel (HIP events; rocprofv3 cooperative + pre-pass) | cells/s (step) | `roofline.frac` (HBM, algorithmic bytes) | measured HBM bytes ÷ algorithmic | `roofline_fp64.frac` (nominal / sustained peak) |")
print("|---|---|---|---|---|---|---|---|")
for ln in open(lines):
    try:
        d = json.loads(ln)
    except Exception:
        continue
    w, m = d["config"]["workload"], d["config"]["mode"]
    r = d["roofline"]
    pm = None
    f = os.path.join(root, "%s_pmc_%s_%s.json" % (tag, w, m))
    if os.path.exists(f):
        pm = json.load(open(f))
    rk = ""
    ratio = ""
    if pm:
        ks = (pm.get("rocprof_kernel_stats") or {}).get("kernels") or {}
        co = sum(v["ms_per_step"] for k, v in ks.items() if "hho_local_ops_kernel" in k)
        pr = sum(v["ms_per_step"] for k, v in ks.items() if "hho_cell_pre_kernel" in k)
        if co:
            rk = " (%.3f + %.3f)" % (co, pr)
        if pm.get("algorithmic_bytes_per_launch"):
            ratio = "%.2f / %.2f GB = %.2f" % (pm["hbm_bytes_per_launch_dominant_kernel"] / 1e9, pm["algorithmic_bytes_per_launch"] / 1e9,
                                               pm["hbm_bytes_per_launch_dominant_kernel"] / pm["algorithmic_bytes_per_launch"])
    st = d.get("stage_ms") or {}
    step = "%.3f ms" % d["ms_per_step"]
    if m == "C":
        step += " (rhs %.2f + ops %.2f + fill %.2f)" % (st.get("rhs", 0), st.get("ops", 0), st.get("fill", 0))
    f64 = d.get("roofline_fp64") or {}
    print("| %s | %s | %s | %.3f ms%s | %.0f M | %.3f | %s | %.2f / %.2f |" % (w, m, step, r["kernel_ms"], rk, d["value"] / 1e6, r["frac"], ratio,
                                                                          f64.get("frac", 0), f64.get("frac_of_sustained", 0)))
