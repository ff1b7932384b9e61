#!/usr/bin/env python3
"""Reduce rocprofv3 counter_collection CSVs (separate --pmc passes) to per-kernel, per-dispatch means for every pa::
kernel; FETCH/WRITE in bytes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of a
wide streaming read: doubled; both counters are in KiB).  A step may launch a kernel several times (2048^2 k=3: two
pieces under the 4 GiB record cap): per-step figures = the sum over a pass's dispatches / the number of steps of that
pass (PA_PROFILE_STEPS, default 4 = --steps 3 --warmup 1).   profile_reduce.py <dir> <workload> <mode> [bench.json]"""
import collections, csv, glob, json, os, sys
d, workload, mode = sys.argv[1], sys.argv[2], sys.argv[3]
STEPS = int(os.environ.get("PA_PROFILE_STEPS", "4"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pa::" not in k:
            continue
        per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per.items():
        acc[k][c].append(v)
out = {"workload": workload, "mode": mode, "kernels": {}}
dominant = 0.0
# the kernels bench.py's roofline.kernel_ms spans: L / A -- the local operators (pre-pass + cooperative kernel, or the thread-per-cell
# kernel of the small pairs; with cut cells also their kernel and the merge); C -- every kernel of the step (cell rhs, operators with
# the condensation fused, CSR fill)
DOM = ["hho_local_ops_kernel", "hho_cell_pre_kernel", "hho_small_ops_kernel", "cut_local_ops_kernel", "cut_merge_cells_kernel", "cut_zero_rhs_kernel",
       "cut_merge_condensed_kernel", "static_condensation_kernel", "cut_interface_kernel", "cut_interface_lc_kernel"]
if mode == "C":
    DOM += ["cell_rhs_kernel", "cond_fill_kernel", "cond_rhs_rows_kernel"]
STEP_KERNELS = DOM + ["cell_rhs_kernel", "cond_fill_kernel", "cond_rhs_rows_kernel", "asm_fill_cells_kernel", "asm_fill_faces_kernel", "dirichlet_data_kernel"]
for k, cs in sorted(acc.items()):
    ndisp = max(len(v) for v in cs.values())
    per_step = ndisp / STEPS if any(n in k for n in STEP_KERNELS) and ndisp >= STEPS else 1.0
    m = {c: sum(v) / len(v) * per_step for c, v in cs.items()}
    rec = {"dispatches_per_step": per_step, "per_step_sums" if per_step != 1.0 else "per_dispatch_means": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rec["FETCH_SIZE_bytes"] = m["FETCH_SIZE"] * 1024
        rec["WRITE_SIZE_bytes"] = m["WRITE_SIZE"] * 1024
        rec["hbm_bytes_per_launch_gfx950_corrected"] = 2 * rec["FETCH_SIZE_bytes"] + rec["WRITE_SIZE_bytes"]
        if any(n in k for n in DOM):
            dominant += rec["hbm_bytes_per_launch_gfx950_corrected"]
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_INST_ANY" in m:
        rec["wave_wait_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    out["kernels"][k[:110]] = rec
out["hbm_bytes_per_launch_dominant_kernel"] = dominant     # pre-pass + cooperative kernel of one step
if len(sys.argv) > 4:
    try:
        b = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
        out["bench_kernel_ms_under_profiler"] = b["roofline"]["kernel_ms"]
        out["build_stamp"] = b.get("build_stamp")
        out["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes_per_cell"] * b["roofline"]["cells_per_launch"]
    except Exception as e:      # noqa: BLE001
        out["bench_line_error"] = str(e)
# rocprofv3 --kernel-trace --stats of the same command: the dominant kernels' time PER STEP = calls x average / steps (a step
# launches them once per piece of <= 192 Ki cells); steps under the profiler = settle passes + warmup + timed steps
if len(sys.argv) > 5 and os.path.exists(sys.argv[5]) and "bench_line_error" not in out and len(sys.argv) > 4:
    try:
        b = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
        nsteps = b["steps"] + b["warmup"] + (b.get("settle") or {}).get("passes", 0)
        per = {}
        for r in csv.DictReader(open(sys.argv[5])):
            if any(n in r["Name"] for n in DOM):
                per[r["Name"][:110]] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                                        "ms_per_step": int(r["Calls"]) * float(r["AverageNs"]) / nsteps * 1e-6}
        out["rocprof_kernel_stats"] = {"steps_under_profiler": nsteps, "kernels": per,
                                       "dominant_kernel_ms_per_step": sum(v["ms_per_step"] for v in per.values())}
    except Exception as e:      # noqa: BLE001
        out["rocprof_kernel_stats_error"] = str(e)
out["notes"] = ("separate rocprofv3 --pmc passes with --kernel-trace only; FETCH_SIZE / WRITE_SIZE in KiB; FETCH doubled per "
                "MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); SQ_* cycle counters in quad-cycles except "
                "SQ_LDS_IDX_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES")
print(json.dumps(out, indent=1))
