#!/usr/bin/env python3
"""Register footprint of every local-operator kernel instance (compiled to assembly with the flags _build.py uses):
name, VGPRs, spilled VGPRs, scratch bytes.  A spill in the store phase costs a vmcnt(0) wait per reload."""
import concurrent.futures, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from proton_amd import _build as B

def one(cfg):
    cd, fd, q, gmin = cfg
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [B.hipcc()] + B.FLAGS + ["-DPA_CD=%d" % cd, "-DPA_FD=%d" % fd, "-DPA_QUAD=%d" % q, "-DPA_GMIN=%d" % gmin] + \
            B.PER_CONFIG_FLAGS.get((cd, fd, q), []) + ["--cuda-device-only", "-S", "-o", out, os.path.join(B.CSRC, "hho_inst.hip")]
        subprocess.run(cmd, check=True, capture_output=True)
        txt = open(out).read()
    rows = []
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
        name = m.group(1)
        t = re.search(r"CfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+EEE(Li\d)?", name)
        kind = "pre" if "cell_pre" in name else "small" if "hho_small" in name else {"Li0": "lc", "Li1": "split", "Li2": "cond"}[t.group(6)]
        rows.append((tuple(int(x) for x in t.groups()[:5]), kind, int(m.group(3)), int(m.group(4)), int(m.group(2))))
    return rows

if __name__ == "__main__":
    bad = 0
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        for rows in ex.map(one, B.configs()):
            for cfg, kind, vg, sp, scr in sorted(rows):
                flag = "  <-- spills" if sp or scr else ""
                bad += bool(sp or scr)
                print("cd=%d fd=%d quad=%d stab=%d G=%-2d %-5s vgpr %3d spilled %3d scratch %4d%s" % (*cfg, kind, vg, sp, scr, flag))
    print("instances with spills:", bad)
