#!/usr/bin/env python3
"""Instruction census per stage of one kernel instance from a -S listing built with -DPA_MARKERS.
usage: isa_stages.py file.s <substring of the mangled kernel name>   (e.g. CfgILi3ELi2ELi0ELi2ELi32ELi0EEELi0E)
       isa_stages.py --build cd fd quad gmin <substring>              (compiles the instance with _build.py's flags first)
The loop of the kernel is what follows the first marker (S0); `pre` is the prologue in front of the cell loop."""
import collections, os, re, subprocess, sys, tempfile

CATS = ["valu64", "valu", "mfma", "lds", "vmem", "scratch", "salu", "wait", "barrier", "other"]


def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("v_") and ("f64" in op): return "valu64"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def census(path, key):
    """{stage: Counter(category -> static instruction count)} of the first hho_local_ops_kernel whose mangled name contains key,
    plus the resource lines of the kernel ({'vgpr': n, 'scratch': n})"""
    lines = open(path).read().split("\n")
    start = next(i for i, s in enumerate(lines) if s.startswith("_ZN2pa20hho_local_ops_kernel") and key in s and ":" in s)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    stage = "pre"
    cnt = collections.OrderedDict()
    for s in lines[start + 1:end]:
        t = s.strip()
        m = re.match(r"; PAMARK (\S+)", t)
        if m:
            stage = m.group(1); continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        cnt.setdefault(stage, collections.Counter())[cat(t.split()[0])] += 1
    res = {}
    for s in lines[end:end + 80]:
        m = re.match(r"\s*; NumVgprs: (\d+)", s)
        if m: res["vgpr"] = int(m.group(1))
        m = re.match(r"\s*; ScratchSize: (\d+)", s)
        if m: res["scratch"] = int(m.group(1)); break
    return cnt, res


def build_listing(cd, fd, quad, gmin, out):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from proton_amd import _build as B
    cmd = [B.hipcc()] + B.FLAGS + ["-DPA_CD=%d" % cd, "-DPA_FD=%d" % fd, "-DPA_QUAD=%d" % quad, "-DPA_GMIN=%d" % gmin, "-DPA_MARKERS"] + \
        B.PER_CONFIG_FLAGS.get((cd, fd, quad), []) + ["--cuda-device-only", "-S", "-o", out, os.path.join(B.CSRC, "hho_inst.hip")]
    subprocess.run(cmd, check=True, capture_output=True)


def loop_totals(cnt):
    tot = collections.Counter()
    for st, c in cnt.items():
        if st != "pre":
            tot.update(c)
    return tot


if __name__ == "__main__":
    if sys.argv[1] == "--build":
        cd, fd, quad, gmin = (int(x) for x in sys.argv[2:6])
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "k.s")
            build_listing(cd, fd, quad, gmin, path)
            cnt, res = census(path, sys.argv[6])
    else:
        cnt, res = census(sys.argv[1], sys.argv[2])
    print("%-6s" % "stage" + "".join("%9s" % c for c in CATS))
    tot = collections.Counter()
    for st, c in cnt.items():
        print("%-6s" % st + "".join("%9d" % c[k] for k in CATS)); tot.update(c)
    print("%-6s" % "total" + "".join("%9d" % tot[k] for k in CATS))
    lt = loop_totals(cnt)
    print("%-6s" % "loop" + "".join("%9d" % lt[k] for k in CATS))
    print(res)
