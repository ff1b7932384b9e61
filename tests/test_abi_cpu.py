"""CPU-side checks of the boundary (no GPU, no compute calls): libproton_amd.so loads, exports every
symbol include/proton_amd.h declares, the host-only entry points behave like the reference's
hho_degree_info (src/core/core_bits/utils.hpp:62-111) and integrate() size rules, a missing GPU is
an ERROR (there is no CPU fallback), and the C++ drop-in headers compile against the C ABI alone."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "proton_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", h)))


@pytest.fixture(scope="module")
def lib():
    from proton_amd import capi
    return capi.lib()


def test_library_exports_every_declared_symbol(lib):
    from proton_amd import capi
    names = declared_symbols()
    assert len(names) >= 40
    assert names == sorted(capi.EXPORTS)                 # the ctypes binding lists exactly what the header declares
    for n in names:
        assert getattr(lib, n) is not None, n            # ctypes raises AttributeError for a missing export
    # and nothing that looks like an entry point is exported without being declared
    so = os.path.join(ROOT, "proton_amd", "lib", "libproton_amd.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (pa_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == names


def test_every_entry_cites_the_reference():
    """each declaration of the header carries a reference file:line in the comment above it"""
    h = open(os.path.join(ROOT, "include", "proton_amd.h")).read()
    assert len(re.findall(r"[a-z_]+\.(?:hpp|cpp):\d+", h)) >= 40


def test_degree_info_semantics(lib):
    from proton_amd import capi
    assert lib.pa_abi_version() >= 1
    # hho_degree_info(cd, fd): cd in {fd-1, fd, fd+1}, else equal order with a flag (utils.hpp:77-92)
    for cd, fd, ok in [(2, 1, True), (1, 1, True), (0, 1, True), (3, 1, False), (0, 2, False), (4, 3, True)]:
        di, fell_back = capi.degree_info(cd, fd)
        assert bool(fell_back) == (not ok)
        assert (di.cell_deg, di.face_deg, di.rec_deg) == ((cd, fd, fd + 1) if ok else (fd, fd, fd + 1))
    # sizes: (3,2): rbs 10, cbs 10, fbs 3, msize 22, 16 cell points, 3 face points
    di, _ = capi.degree_info(3, 2)
    sz = capi.sizes_for(di, capi.QUAD_TENSOR)
    assert (sz.rbs, sz.cbs, sz.fbs, sz.msize, sz.oper_rows, sz.cell_qps, sz.face_qps) == (10, 10, 3, 22, 9, 16, 3)
    sz = capi.sizes_for(di, capi.QUAD_FAN)
    assert sz.cell_qps == 4 * 13                         # rules[6] = rule_7 (13 points) per fan triangle: the off-by-one
    di, _ = capi.degree_info(0, 1)
    assert capi.sizes_for(di, capi.QUAD_TENSOR).msize == 9


def test_error_codes_without_a_gpu(lib):
    from proton_amd import capi
    sz = capi.Sizes()
    # the Dunavant degree-8 hole: a status code instead of the reference's empty rule (quadratures.hpp:242-268)
    assert lib.pa_sizes_for(capi.DegreeInfo(4, 3, 4), capi.QUAD_FAN, C.byref(sz)) == 3          # PA_ERR_QUADRATURE
    assert lib.pa_sizes_for(capi.DegreeInfo(9, 9, 10), capi.QUAD_TENSOR, C.byref(sz)) in (2, 3)  # beyond the closed-form rules
    assert lib.pa_sizes_for(capi.DegreeInfo(2, 1, 2), 7, C.byref(sz)) == 1                       # PA_ERR_INVALID_ARG
    # NULL context: every entry point refuses instead of crashing
    assert lib.pa_context_synchronize(None) != 0
    assert lib.pa_mesh_counts(None, None, None) != 0
    assert lib.pa_local_ops_batch(None, capi.DegreeInfo(2, 1, 2), 0, 2, 0, 0, None, None, None, None, None) != 0


def test_no_cpu_fallback():
    """without a GPU the context cannot be created: PA_ERR_HIP, not a silent host path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from proton_amd import capi
    h = C.c_void_p()
    st = capi.lib().pa_context_create(0, None, 1, C.byref(h))
    assert st == 4 and not h.value                        # PA_ERR_HIP
    with pytest.raises(RuntimeError):
        capi.Context(0)


@pytest.mark.parametrize("name", ["convergence_driver", "obstacle_driver", "cuthho_driver"])
def test_host_headers_compile_against_the_c_abi(name):
    """proton_amd/host/hho.hpp and host/cuthho.hpp need nothing but a C++17 compiler and the C ABI"""
    out_dir = os.path.join(ROOT, "tests", "cpp", "build")
    os.makedirs(out_dir, exist_ok=True)
    lib_dir = os.path.join(ROOT, "proton_amd", "lib")
    cmd = ["g++", "-O0", "-std=c++17", "-Wall", "-Werror", "-o", os.path.join(out_dir, name + "_cpu"),
           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-L" + lib_dir, "-lproton_amd", "-Wl,-rpath," + lib_dir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
