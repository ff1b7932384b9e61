#!/usr/bin/env python3
"""Instruction census per stage of one kernel instance from a -S listing built with -DPA_MARKERS.
usage: isa_stages.py file.s <substring of the mangled kernel name>   (e.g. CfgILi3ELi2ELi0ELi2ELi32EEELb0)"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, s in enumerate(lines) if s.startswith("_ZN2pa20hho_local_ops_kernel") and key in s and s.rstrip().endswith(("E:", )) or (s.startswith("_ZN2pa20hho_local_ops_kernel") and key in s and ": " in s))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
stage = "pre"
cnt = collections.OrderedDict()
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
for s in lines[start + 1:end]:
    t = s.strip()
    m = re.match(r"; PAMARK (\S+)", t)
    if m:
        stage = m.group(1); continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    op = t.split()[0]
    cnt.setdefault(stage, collections.Counter())[cat(op)] += 1
cats = ["valu64", "valu", "mfma", "lds", "vmem", "scratch", "salu", "wait", "barrier", "other"]
print("%-6s" % "stage" + "".join("%9s" % c for c in cats))
tot = collections.Counter()
for st, c in cnt.items():
    print("%-6s" % st + "".join("%9d" % c[k] for k in cats)); tot.update(c)
print("%-6s" % "total" + "".join("%9d" % tot[k] for k in cats))
