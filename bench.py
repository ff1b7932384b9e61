#!/usr/bin/env python3
"""bench.py -- cells assembled per second on an N x N quad mesh (BASELINE.json's metric).

A "step" is one pass of the hot path over the whole mesh: for every cell the local operator
lc = data + stab (make_hho_laplacian + make_hho_fancy_stabilization, hho.hpp:32-237) and the
cell right-hand side (make_rhs, utils.hpp:153-174) are computed and written to HBM
(mode L of BASELINE.md section 4: 8 msize^2 + 8 cbs + 80 algorithmic bytes per cell).
Default workload: the north-star target, 1024 x 1024, k = 2 (hho_degree_info(3, 2)), tensor
Gauss, fancy stabilization.  Inputs (the structured mesh) are generated on the device and are
resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

With N > 1 the cell rows are block-partitioned over the ranks (strong scaling, same mesh) and
every step ends with the exchange the north star names: static condensation of the cell dofs and
an RCCL all_gather of the condensed face-dof blocks (values only; indices are closed-form).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # name: (N, cd, fd, quad, stab, domain lo, hi, rhs fn id, rhs dinc, source config)
    "quad1024_k2": dict(N=1024, cd=3, fd=2, quad="tensor", stab="fancy", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0,
                        note="north-star target: 1024x1024 quad_mesh, hho_degree_info(3,2), fancy stabilization"),
    "quad256_k1_fan": dict(N=256, cd=2, fd=1, quad="fan", stab="naive", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0,
                           note="configs[1]: cuthho_square -M 256 -N 256 -k 1 -f, uncut cells (fan quadrature, naive stabilization)"),
    "quad512_k2_fan": dict(N=512, cd=3, fd=2, quad="fan", stab="naive", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0,
                           note="configs[2] without the cut cells: 512x512 k=2 fan quadrature, naive stabilization"),
    "obstacle512_k1": dict(N=512, cd=0, fd=1, quad="tensor", stab="fancy", lo=(-1.0, -1.0), hi=(1.0, 1.0), fn=3, dinc=1,
                           note="configs[3]: apps/obstacle 512x512 k=1, hho_degree_info(0,1)"),
    "quad2048_k3": dict(N=2048, cd=4, fd=3, quad="tensor", stab="fancy", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0,
                        note="configs[4]: 2048x2048 k=3 Laplacian, hho_degree_info(4,3)"),
    "cuthho512_k2": dict(N=512, cd=3, fd=2, quad="fan", stab="naive", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0, cut=True,
                         note="configs[2]: cuthho_square -M 512 -N 512 -k 2 -f, circle r=0.35, -r 4, node displacement: "
                              "uncut cells (fan quadrature, naive stabilization) + cut cells (Nitsche operators), merged"),
    "quad1024_k2_general": dict(N=1024, cd=3, fd=2, quad="tensor", stab="fancy", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0, perturb=0.1,
                                note="the headline on GENERAL quadrilaterals: interior nodes displaced by U(-0.1 h, 0.1 h) (the commented-out "
                                     "perturbation of convergence_test.cpp:176-187, numpy default_rng(12345)): no two cells are congruent, "
                                     "so nothing can be reused between cells"),
    "quad1024_k1": dict(N=1024, cd=2, fd=1, quad="tensor", stab="fancy", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0,
                        note="1024x1024 k=1"),
    "quad1024_k3": dict(N=1024, cd=4, fd=3, quad="tensor", stab="fancy", lo=(0.0, 0.0), hi=(1.0, 1.0), fn=1, dinc=0,
                        note="1024x1024 k=3"),
}

# BASELINE.json's metric string, verbatim; the achieved HBM GB/s it mentions is roofline.achieved
BASELINE_METRIC = 'cells assembled/sec (+ achieved HBM GB/s) on N×N quad mesh, k=1..3'
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # 256 CU x 4 SIMD x 16 FMA/clk x 2 x 2.4 GHz; v_mfma_f64_16x16x4_f64 measured at 64 clk = the same rate
# FP64 work of the REFERENCE's algorithm per cell (hand operation-count model, +-30 %)
REFERENCE_FLOPS_PER_CELL = {(2, 1, "tensor", "fancy"): 16.7e3, (2, 1, "fan", "naive"): 14.7e3, (3, 2, "fan", "naive"): 60.5e3,
                            (3, 2, "tensor", "fancy"): 67.1e3, (0, 1, "tensor", "fancy"): 9.1e3, (4, 3, "tensor", "fancy"): 197.5e3}


def bytes_per_cell(msize, cbs):
    """Algorithmic bytes per cell, mode L (BASELINE.md section 4 / SURVEY.md section 8d):
    lc written (8 msize^2) + cell rhs written (8 cbs) + 4 node coordinates (64) + 4 u32 ids (16)."""
    return 8 * msize * msize + 8 * cbs + 80


def cpu_baseline_cut(w, target_seconds=15.0):
    """Oracle on a bounded sample of the cut workload: a smaller mesh of the same problem
    (the per-cell cost does not depend on N; the cut-cell fraction scales like 1/N)."""
    import oracle_lib
    import cuthho_driver
    N = min(w["N"], 384)            # ~5-10 s of single-thread work
    msh = oracle_lib.CutMesh(N, refsteps=4)
    di = oracle_lib.degrees(w["cd"], w["fd"])
    t0 = time.perf_counter()
    cuthho_driver.oracle_cut_provider(msh, di)
    dt = time.perf_counter() - t0
    ncut = int((msh.cell_loc == oracle_lib.CUT_ON_INTERFACE).sum())
    return {"value": msh.nc / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "%dx%d mesh of the same problem (%d cells, %d cut), %.1f s, oracle cut + uncut operators and rhs, single thread"
                      % (N, N, msh.nc, ncut, dt)}


def perturbation(N, lo, hi, amount):
    """displacements of the interior nodes, U(-amount h, amount h), numpy default_rng(12345) -- the same
    array for the GPU mesh and for the CPU baseline's mesh"""
    import numpy as np
    rng = np.random.default_rng(12345)
    h = (hi[0] - lo[0]) / N
    d = rng.uniform(-amount * h, amount * h, size=((N + 1) * (N + 1), 2))
    ij = np.arange((N + 1) * (N + 1))
    i, j = ij % (N + 1), ij // (N + 1)
    d[~((i > 0) & (i < N) & (j > 0) & (j < N))] = 0.0
    return d


def general_quad_mesh(torch, N, lo, hi, amount, device):
    """The generator mesh (basic_mesh.hpp:230-298: points min + i h, cells {p, p+1, p+Nx+2, p+Nx+1})
    with displaced interior nodes, built directly as device arrays."""
    i = torch.arange(N + 1, dtype=torch.float64, device=device)
    hx, hy = (hi[0] - lo[0]) / N, (hi[1] - lo[1]) / N
    pts = torch.stack([(lo[0] + i * hx).repeat(N + 1), (lo[1] + i * hy).repeat_interleave(N + 1)], dim=1).contiguous()
    pts += torch.from_numpy(perturbation(N, lo, hi, amount)).to(device)
    ci = torch.arange(N, device=device, dtype=torch.int64)
    p0 = (ci.repeat_interleave(N) * (N + 1) + ci.repeat(N))                    # cell (i, j) -> j (Nx+1) + i, cell id = j Nx + i
    ptids = torch.stack([p0, p0 + 1, p0 + N + 2, p0 + N + 1], dim=1).to(torch.int32).contiguous()    # ids < 2^31: same bits as u32
    return pts, ptids


def workload_mesh(w):
    """host arrays of the workload's mesh for the CPU baseline (oracle side)"""
    import oracle_lib
    N = w["N"]
    mp, points, ptids = oracle_lib.make_mesh(N, N, w["lo"], w["hi"])
    if w.get("perturb"):
        points += perturbation(N, w["lo"], w["hi"], w["perturb"])
    return points, ptids


def cpu_baseline(w, sample_rows, target_seconds=15.0):
    """The oracle (CPU restatement, single thread like the reference) on a bounded sample of the
    same workload: the first `sample_rows` cell rows of the same mesh (0 = as many rows as take
    about `target_seconds` at the rate measured on a small probe)."""
    import oracle_lib
    N = w["N"]
    points, ptids = workload_mesh(w)
    di = oracle_lib.degrees(w["cd"], w["fd"])
    quad = oracle_lib.QUAD_TENSOR if w["quad"] == "tensor" else oracle_lib.QUAD_FAN
    stab = oracle_lib.STAB_FANCY if w["stab"] == "fancy" else oracle_lib.STAB_NAIVE
    probe = min(N * N, 4096)
    t0 = time.perf_counter()
    oracle_lib.local_ops_batch(points, ptids, di, quad, stab, first=0, n=probe, fn=w["fn"], rhs_di=w["dinc"],
                               want=("lc",))                       # warm-up + rate probe
    rate = probe / (time.perf_counter() - t0)
    if sample_rows <= 0:
        sample_rows = max(1, min(N, int(target_seconds * rate / N)))
    n = sample_rows * N
    t0 = time.perf_counter()
    st, _ = oracle_lib.local_ops_batch(points, ptids, di, quad, stab, first=0, n=n, fn=w["fn"], rhs_di=w["dinc"],
                                       want=("lc",))
    dt = time.perf_counter() - t0
    assert st == 0
    return {"value": n / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "first %d of %d cell rows (%d cells) of the same mesh, %.1f s, oracle/hho_oracle.c "
                      "(-O3 -mavx, single thread like the reference)" % (sample_rows, N, n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="quad1024_k2", choices=sorted(WORKLOADS))
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "none"],
                    help="N>1 only: all_gather of the condensed face-dof blocks at the end of every step")
    ap.add_argument("--chunks", type=int, default=4,
                    help="N>1 only: the local rows are processed in this many pieces; the all_gather of a piece overlaps the kernels of the next")
    ap.add_argument("--backend", default=os.environ.get("PA_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl = RCCL, one rank per GPU (what the driver runs); gloo = rehearsal of the N>1 code path with "
                         "several ranks sharing the visible GPU(s) and a host-staged exchange (numbers not comparable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="cell rows timed by the CPU baseline (0 = auto, ~15 s)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    import torch
    import torch.distributed as dist
    import proton_amd as pa
    from proton_amd.batch import BatchAssembler
    from proton_amd.partition import ChunkedExchange, condensed_per_cell, row_partition

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    rehearsal = world > 1 and args.backend == "gloo"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()                                     # communicators exist before anything is timed

    w = WORKLOADS[args.workload]
    N = w["N"]
    quad = pa.QUAD_TENSOR if w["quad"] == "tensor" else pa.QUAD_FAN
    stab = pa.STAB_FANCY if w["stab"] == "fancy" else pa.STAB_NAIVE
    di, _ = pa.degree_info(w["cd"], w["fd"])
    sz = pa.sizes_for(di, quad)
    asm = BatchAssembler(local_rank)
    r0, r1 = row_partition(N, world, rank)
    cut = bool(w.get("cut"))
    if cut:
        if world > 1:
            raise SystemExit("the cut workload is single-GPU in this round")
        asm.cut_preprocess(N, refsteps=4)                    # host preprocessing: outside the timed region
        # (side-stream overlap of the cut cells' kernel, pa_context_set_cut_overlap: measured SLOWER here, 0.75 vs 0.69 ms
        # per step -- the persistent grid of the uncut cells' kernel holds the whole chip, the two only contend)
        asm.ctx.set_cut_overlap(bool(os.environ.get("PA_CUT_OVERLAP")))
    elif w.get("perturb"):
        if world > 1:
            raise SystemExit("the general-quadrilateral workload is single-GPU in this round")
        mesh_keep = general_quad_mesh(torch, N, w["lo"], w["hi"], w["perturb"], asm.device)     # caller-owned device arrays
        asm.ctx.mesh_attach_device(mesh_keep[0].data_ptr(), (N + 1) * (N + 1), mesh_keep[1].data_ptr(), N * N)
    else:
        asm.generate_mesh(N, N, w["lo"], w["hi"], rows=(r0, r1))
    n_local = asm.ncells
    dev = asm.device

    lc = torch.empty((n_local, sz.msize, sz.msize), dtype=torch.float64, device=dev)
    rhs = torch.empty((n_local, sz.cbs), dtype=torch.float64, device=dev)
    out = {"lc": lc}
    exchange = world > 1 and args.exchange == "allgather"
    if exchange:
        nf = 4 * sz.fbs
        ex = ChunkedExchange(N, world, rank, condensed_per_cell(sz.fbs, packed=True), dev, args.chunks, host_staged=rehearsal)

    if cut:
        cut_lc = torch.empty((max(asm.ncut, 1), sz.msize, sz.msize), dtype=torch.float64, device=dev)
        cut_rhs = torch.empty((max(asm.ncut, 1), sz.cbs), dtype=torch.float64, device=dev)

    nchunks = ex.chunks if exchange else 1
    k_start = [[torch.cuda.Event(enable_timing=True) for _ in range(nchunks)] for _ in range(args.steps)]
    k_stop = [[torch.cuda.Event(enable_timing=True) for _ in range(nchunks)] for _ in range(args.steps)]

    def step(i=None):
        if not exchange:
            if cut and asm.ncut:
                # the cut cells first (on the context's side stream if PA_CUT_OVERLAP is set)
                asm.ctx.cut_local_ops(w["fd"], asm.level_set, pa.capi.LOC_NEGATIVE, w["fn"], 2, None, None, None,
                                      cut_lc.data_ptr(), cut_rhs.data_ptr(), None)
            if i is not None:
                k_start[i][0].record()
            asm.local_ops(w["cd"], w["fd"], quad, stab, want=(), out=out)
            if i is not None:
                k_stop[i][0].record()
            asm.cell_rhs(w["cd"], w["fn"], quad, dinc=w["dinc"], out=rhs)
            if cut and asm.ncut:
                asm.ctx.cut_merge(w["fd"], pa.capi.LOC_NEGATIVE, cut_lc.data_ptr(), cut_rhs.data_ptr(), lc.data_ptr(), rhs.data_ptr())
            return
        # N > 1: piece by piece; the collective of piece k overlaps the kernels of piece k + 1
        for k in range(ex.chunks):
            first, n = ex.piece_cells(k)
            if i is not None:
                k_start[i][k].record()
            asm.ctx.local_ops(di, quad, stab, first, n, None, None, None, lc[first:first + n].data_ptr(), None)
            if i is not None:
                k_stop[i][k].record()
            asm.ctx.cell_rhs(w["cd"], w["dinc"], quad, w["fn"], first, n, rhs[first:first + n].data_ptr(), None)
            S_view, g_view = ex.local_S_g(k, nf)
            asm.ctx.static_condensation_packed(di, n, lc[first:first + n].data_ptr(), rhs[first:first + n].data_ptr(),
                                               S_view.data_ptr(), g_view.data_ptr(), None)
            ex.exchange_async(k)
        ex.wait()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kern_ms = sum(a.elapsed_time(b) for sa, sb in zip(k_start, k_stop) for a, b in zip(sa, sb)) / args.steps

    # N > 1, informational (not `value`): the same K steps without the exchange -- local operators, right-hand
    # sides and condensation of every rank's cells, no collective -- so that the cost of the all_gather is visible
    elapsed_noex = None
    if exchange:
        def step_noex():
            asm.ctx.local_ops(di, quad, stab, 0, n_local, None, None, None, lc.data_ptr(), None)
            asm.ctx.cell_rhs(w["cd"], w["dinc"], quad, w["fn"], 0, n_local, rhs.data_ptr(), None)
            for k in range(ex.chunks):
                first, n = ex.piece_cells(k)
                S_view, g_view = ex.local_S_g(k, nf)
                asm.ctx.static_condensation_packed(di, n, lc[first:first + n].data_ptr(), rhs[first:first + n].data_ptr(),
                                                   S_view.data_ptr(), g_view.data_ptr(), None)
        step_noex()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_noex()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        elapsed_noex = time.perf_counter() - t1
    t = torch.tensor([elapsed, kern_ms, elapsed_noex or 0.0], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms, elapsed_noex = float(t[0]), float(t[1]), float(t[2])

    # sanity: the timed work produced finite local matrices (not a cached / skipped result)
    probe = lc[:: max(1, n_local // 64)]
    if not os.environ.get("PA_ABLATE"):      # (profiling-only stage ablation produces garbage on purpose)
        assert bool(torch.isfinite(probe).all()) and float(probe.abs().max()) > 0.0

    if exchange:                             # every rank's condensed blocks arrived (finite, non-trivial)
        for r in range(world):
            Sg, gg = ex.gathered_S_g(r, 4 * sz.fbs)
            pr = Sg[:: max(1, Sg.shape[0] // 16)]
            assert bool(torch.isfinite(pr).all()) and float(pr.abs().max()) > 0.0, "gathered block of rank %d" % r

    if rank == 0:
        total_cells = N * N
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells / (elapsed / args.steps)
        bpc = bytes_per_cell(sz.msize, sz.cbs)
        li = asm.ctx.launch_info(di, quad, stab, n_local)
        achieved = n_local * bpc / (kern_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf)).get(args.workload)
                if rec and rec.get("n_gpus", 1) == world:
                    traffic = rec["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        res = {
            "metric": BASELINE_METRIC,
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "mesh": "%dx%d quad_mesh on [%g,%g]^2 (reference generator)" % (N, N, w["lo"][0], w["hi"][0]),
                       "hho_degree_info": [w["cd"], w["fd"]], "k": w["fd"], "quadrature": w["quad"], "stabilization": w["stab"],
                       "cells": total_cells, "msize": sz.msize, "outputs": "lc (msize^2 f64) + cell rhs per cell, to HBM",
                       "parallelism": "cell rows block-partitioned over %d GPU(s)" % world,
                       "exchange": (("static condensation + %s all_gather of the condensed face blocks (upper triangles, values only)"
                                     % ("host-staged gloo (REHEARSAL, not RCCL)" if rehearsal else "RCCL")) if exchange else "none"),
                       "note": w["note"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": li.kernel_name.decode() + " + hho_cell_pre (its one-thread-per-cell pre-pass)",
                         "kernel_ms": kern_ms,
                         "kernel_ms_is": "HIP events around the local-operator launches of one step (pre-pass + cooperative kernel where the path is split: the sum of their rocprofv3 averages)",
                         "algorithmic_bytes_per_cell": bpc, "cells_per_launch": n_local,
                         "lanes_per_cell": li.lanes_per_cell, "grid_blocks": li.grid_blocks,
                         "lds_bytes_per_block": li.lds_bytes_per_block},
            # the survey's second roof: the path is FP64-bound, not HBM-bound, for k >= 2 (arithmetic intensity of the
            # reference's algorithm 17-25 flop/B against a machine balance of 9.8).  Informational: the kernel executes
            # fewer flops than the reference's algorithm (cell integrals through moments), so this is work done per
            # reference flop, not an instruction-level utilisation.
            "roofline_fp64": (lambda fl: None if fl is None else {
                "bound": "fp64", "achieved": n_local * fl / (kern_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": n_local * fl / (kern_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "reference_flops_per_cell": fl, "source": "SURVEY.md 8(d) operation-count model of the reference's algorithm, +-30 %"})(
                    REFERENCE_FLOPS_PER_CELL.get((w["cd"], w["fd"], w["quad"], w["stab"]))),
            "without_exchange": None if not exchange else {
                "value": total_cells / (elapsed_noex / args.steps), "unit": "cells/s", "ms_per_step": elapsed_noex / args.steps * 1e3,
                "what": "the same steps (local operators + rhs + condensation on every rank) without the all_gather, max over ranks"},
            "kernel_only_cells_per_s": total_cells / (kern_ms * 1e-3) if world == 1 else n_local * world / (kern_ms * 1e-3),
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline_cut(w) if cut else cpu_baseline(w, args.cpu_sample_rows)
            res["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
