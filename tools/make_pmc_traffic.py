#!/usr/bin/env python3
"""profiles/pmc_traffic.json (what bench.py's roofline.traffic reads) from the reduced counter files of a profile round:
    tools/make_pmc_traffic.py r02f
one entry per workload|mode: HBM bytes of the dominant kernels of one step (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes)
the kernel time they were taken at and the STAMP of the build they were measured on (proton_amd/_build.py:build_stamp -- bench.py
emits an entry as roofline.traffic only when the stamp is the running build's)."""
import glob, json, os, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
out = {}
for f in sorted(glob.glob(os.path.join(root, tag + "_pmc_*.json"))):
    d = json.load(open(f))
    if not d.get("hbm_bytes_per_launch_dominant_kernel") or "bench_kernel_ms_under_profiler" not in d:
        continue
    out["%s|%s" % (d["workload"], d["mode"])] = {
        "n_gpus": 1, "hbm_bytes_per_launch": d["hbm_bytes_per_launch_dominant_kernel"],
        "build_stamp": d.get("build_stamp"),
        "kernel_ms": d["bench_kernel_ms_under_profiler"],
        "rocprof_kernel_ms": (d.get("rocprof_kernel_stats") or {}).get("dominant_kernel_ms_per_step"),
        "algorithmic_bytes_per_launch": d.get("algorithmic_bytes_per_launch"),
        "source": os.path.basename(f) + " (the kernels of roofline.kernel_ms of one step; separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled "
                  "per the guide's gfx950 correction)"}
json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("%d entries" % len(out))
