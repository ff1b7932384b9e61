"""bench.py's output contract (the JSON line the driver parses) and its refusal to run without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_refuses_to_run_without_a_gpu():
    """No CPU fallback: on a machine without a GPU the bench exits with an error instead of timing anything else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    r = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=300)
    assert r.returncode != 0
    assert "GPU" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]       # no result line


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields():
    """One JSON line with BASELINE.json's metric, the roofline and cpu_baseline objects, the settle phase reported, and
    numbers that are consistent with each other (value = cells / time of a step; achieved = algorithmic bytes / kernel time)."""
    r = _run(["--gpus", "1", "--steps", "4", "--warmup", "1", "--settle-ms", "30", "--workload", "quad256_k1_fan", "--cpu-sample-rows", "8"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "cells/s"
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["config"]["workload"] == "quad256_k1_fan" and "model" not in d["config"]
    assert d["settle"]["passes"] >= 1
    cells = d["config"]["cells"]
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12 and 0.0 < ro["frac"] < 1.0
    assert abs(ro["achieved"] - ro["cells_per_launch"] * ro["algorithmic_bytes_per_cell"] / (ro["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * ro["achieved"]
    assert ro["kernel_ms"] <= d["ms_per_step"] * 1.05
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "cells/s" and cb["sample"]
    assert d["gpu_over_cpu"] > 10.0                                     # the north star's floor
    # counters are never replayed across builds: the line names its build, and `traffic` is that build's or null with the reason
    from proton_amd import _build
    assert d["build_stamp"] == _build.build_stamp() and ro["traffic_source"]
    assert ro["traffic"] is None or "counters of this build" in ro["traffic_source"]


@pytest.mark.gpu
def test_bench_mode_A_reports_the_matrix_assembly_span():
    """--mode A: operators + rhs + assembler<Mesh>'s own system directly in CSR, next to the CPU restatement's "Matrix assembly" span"""
    r = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--settle-ms", "20", "--workload", "quad256_k1_fan", "--mode", "A",
              "--cpu-sample-rows", "8"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["mode"] == "A" and d["stage_ms"]["fill"] > 0 and d["stage_ms"]["ops"] > 0
    ma = d["matrix_assembly"]
    assert ma["cpu_one_core"] > 0 and ma["gpu_over_cpu_one_core"] > 10.0 and abs(ma["value"] - d["value"]) < 1e-9 * d["value"]
    pk = d["roofline"]["per_kernel"]
    assert pk["fill"]["bytes"] > 0 and 0 < pk["fill"]["frac"] < 1
