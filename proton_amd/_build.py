"""Build libproton_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

One translation unit per (cell degree, face degree, quadrature kind) listed in
csrc/pa_configs.def, compiled in parallel, plus csrc/capi.hip, csrc/csr.hip, csrc/solver.hip,
csrc/condensed.hip, csrc/assembler_csr.hip and csrc/comm.hip; linked into
proton_amd/lib/libproton_amd.so.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# PA_BUILD_TAG=<name>: an A/B tuning build with its own objects and library (lib/variants/<name>/), selected at run time
# with PA_LIB; the shipped library is never built from a tagged tree
_TAG = os.environ.get("PA_BUILD_TAG")
_OUT = os.path.join(HERE, "lib", "variants", _TAG) if _TAG else os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(_OUT, "obj")
LIB_PATH = os.path.join(_OUT, "libproton_amd.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if _TAG:                                    # tuning builds read the profiling knobs (PA_ABLATE, PA_BLOCKS_PER_CU, PA_LANES_PER_CELL)
    FLAGS.append("-DPA_TUNING=1")
if os.environ.get("PA_EXTRA_FLAGS"):        # experiments: extra compiler flags, e.g. "-mllvm -amdgpu-use-amdgpu-trackers"
    FLAGS.extend(os.environ["PA_EXTRA_FLAGS"].split())
if os.environ.get("PA_WAVES_PER_EU"):      # tuning knob: register budget of the local-operator kernel
    FLAGS.append("-DPA_WAVES_PER_EU=" + os.environ["PA_WAVES_PER_EU"])


# Per-instance compiler settings, each measured on the MI355X against the default.
# No instance may spill: a scratch reload in the store phase waits for every outstanding store (vmcnt(0)).
PER_CONFIG_FLAGS = {
    # k = 2 tensor: lc through the LDS image (-3 %).  Three waves per SIMD: with the table reads of S3b, the rows of L in the
    # forward substitution and the corner tile fetched in batches (one LDS round trip each instead of one per use) the
    # kernel wants more than the 128 registers of a fourth wave -- 1.42 ms at 3 waves against 1.54 (9 spills) at 4
    (3, 2, 0): ["-mllvm", "-amdgpu-use-amdgpu-trackers", "-DPA_DIRECT_MIN=99"],
    (3, 2, 1): ["-mllvm", "-amdgpu-use-amdgpu-trackers", "-DPA_FWD_CAP=16"],       # (24 doubles of L per batch: one register spilled)
    (2, 2, 0): ["-DPA_FWD_CAP=16"],
    (2, 2, 1): ["-DPA_FWD_CAP=16"],
    (2, 1, 0): ["-mllvm", "-amdgpu-use-amdgpu-trackers", "-DPA_WAVES_PER_EU=4"],
    (2, 1, 1): ["-mllvm", "-amdgpu-use-amdgpu-trackers", "-DPA_WAVES_PER_EU=4"],
    # k = 3 pre-pass: 290 VGPRs leave one wave per SIMD; bounded to 256 it spills 34 (no stores in flight there) and
    # runs two: 258 -> 233 us per 1 M cells
    # ... and lc through the LDS image: the direct 8-byte stores of the accumulators made the L2 fetch the partially
    # written lines (FETCH 3.6 GB against 1.2 GB of records per 1 M cells); full 16-byte runs: 2.93 -> 2.83 ms
    (4, 3, 0): ["-DPA_PRE_WAVES=2", "-DPA_DIRECT_MIN=99"],
}


if _TAG and os.environ.get("PA_DROP_FLAGS"):          # A/B builds: per-instance settings to leave out
    _drop = set(os.environ["PA_DROP_FLAGS"].split())
    PER_CONFIG_FLAGS = {k: [f for f in v if f not in _drop] for k, v in PER_CONFIG_FLAGS.items()}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the HIP extension")
    return exe


def configs():
    out = []
    with open(os.path.join(CSRC, "pa_configs.def")) as f:
        for line in f:
            m = re.match(r"\s*PA_CONFIG\(\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*\)", line)
            if m:
                out.append(tuple(int(x) for x in m.groups()))
    return out


def _deps_mtime():
    deps = [os.path.join(CSRC, n) for n in os.listdir(CSRC)]
    deps += [os.path.join(HERE, "..", "include", "proton_amd.h"), os.path.join(HERE, "_build.py")]
    return max(os.path.getmtime(d) for d in deps)


def _stamp(defs):
    """the flag set an object was compiled with, kept next to it: an object built under PA_EXTRA_FLAGS / PA_WAVES_PER_EU
    (tuning) is stale for a default build even when its mtime is recent"""
    return " ".join(FLAGS + defs)


def build_stamp():
    """sha256 (16 hex digits) of everything the shipped library is compiled from: every file of csrc/, the C ABI header, the
    flags and the per-instance settings of this file.  bench.py prints it, counter files under profiles/ carry it, and a
    counter entry is used only for the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for n in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(n.encode()); h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "proton_amd.h"), "rb") as f:
        h.update(f.read())
    h.update(repr((FLAGS, sorted(PER_CONFIG_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def _stale(obj, defs, newest):
    if not os.path.exists(obj) or os.path.getmtime(obj) < newest:
        return True
    try:
        with open(obj + ".flags") as f:
            return f.read() != _stamp(defs)
    except OSError:
        return True


def _compile(job):
    src, obj, defs = job
    cmd = [hipcc()] + FLAGS + defs + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
    with open(obj + ".flags", "w") as f:
        f.write(_stamp(defs))
    return obj


def build(force=False, verbose=False, jobs=None):
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest = _deps_mtime()
    todo, objs = [], []
    for (cd, fd, q, gmin) in configs():
        obj = os.path.join(OBJ_DIR, "inst_%d_%d_%d.o" % (cd, fd, q))
        objs.append(obj)
        defs = ["-DPA_CD=%d" % cd, "-DPA_FD=%d" % fd, "-DPA_QUAD=%d" % q, "-DPA_GMIN=%d" % gmin] + \
            ([] if os.environ.get("PA_WAVES_PER_EU") or os.environ.get("PA_NO_PER_CONFIG_FLAGS")
             else PER_CONFIG_FLAGS.get((cd, fd, q), []))
        if force or _stale(obj, defs, newest):
            todo.append((os.path.join(CSRC, "hho_inst.hip"), obj, defs))
    for unit in ("capi", "csr", "solver", "condensed", "assembler_csr", "comm"):
        unit_obj = os.path.join(OBJ_DIR, unit + ".o")
        objs.append(unit_obj)
        if force or _stale(unit_obj, [], newest):
            todo.append((os.path.join(CSRC, unit + ".hip"), unit_obj, []))
    if todo:
        if verbose:
            print("proton_amd: compiling %d translation units for %s" % (len(todo), ARCH), flush=True)
        jobs = jobs or min(8, os.cpu_count() or 1)
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            for obj in ex.map(_compile, todo):
                if verbose:
                    print("  built", os.path.basename(obj), flush=True)
    if todo or not os.path.exists(LIB_PATH):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose:
            print("proton_amd: linked", LIB_PATH, flush=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
