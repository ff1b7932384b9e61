#!/usr/bin/env python3
"""Reduce the counter_collection CSVs of tools/pmc.sh to per-kernel, per-dispatch means (local-operator kernels only)."""
import collections, csv, glob, json, sys
d, workload = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values per dispatch
for f in glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)                                # (kernel, dispatch, counter) -> summed over instances
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hho_local_ops_kernel" not in k and "hho_cell_pre_kernel" not in k:
            continue
        per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per.items():
        acc[k][c].append(v)
out = {"workload": workload, "kernels": {}}
total = 0.0
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rec = {"per_dispatch_means": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        # KiB; gfx950 tallies a 128-B read request at 64 B: FETCH doubled (upper bound for 16-B loads)
        rec["FETCH_SIZE_bytes"] = m["FETCH_SIZE"] * 1024
        rec["WRITE_SIZE_bytes"] = m["WRITE_SIZE"] * 1024
        rec["hbm_bytes_per_launch_gfx950_corrected"] = 2 * rec["FETCH_SIZE_bytes"] + rec["WRITE_SIZE_bytes"]
        total += rec["hbm_bytes_per_launch_gfx950_corrected"]
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_INST_ANY" in m:
        rec["derived"] = {"wave_wait_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], "waves": m.get("SQ_WAVES")}
    out["kernels"][k[:100]] = rec
out["hbm_bytes_per_step_local_operator_kernels"] = total
out["notes"] = ("separate --pmc passes; FETCH/WRITE in KiB; FETCH doubled per MI355X_MICROARCH.md; SQ_* cycle counters in quad-cycles "
                "except SQ_LDS_IDX_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES")
print(json.dumps(out, indent=1))
