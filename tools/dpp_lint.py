#!/usr/bin/env python3
"""ISA check for the DPP operands of the substitutions (hho_device.hpp, Cfg::DPPFWD): a VALU write of a register needs 2 wait
states before a DPP instruction reads it as its DPP source.  The inline assembly is invisible to the compiler's hazard
recognizer, so every instance is compiled to assembly and each  row_newbcast  instruction is checked against the two
instructions before it.  Exit status 1 and a listing if any is found."""
import concurrent.futures, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from proton_amd import _build as B


def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def one(cfg):
    cd, fd, q, gmin = cfg
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [B.hipcc()] + B.FLAGS + ["-DPA_CD=%d" % cd, "-DPA_FD=%d" % fd, "-DPA_QUAD=%d" % q, "-DPA_GMIN=%d" % gmin] + \
            B.PER_CONFIG_FLAGS.get((cd, fd, q), []) + ["--cuda-device-only", "-S", "-o", out, os.path.join(B.CSRC, "hho_inst.hip")]
        subprocess.run(cmd, check=True, capture_output=True)
        lines = [ln.strip() for ln in open(out)]
    ins = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) and not ln.endswith(":")]
    bad, ndpp = [], 0
    for i, ln in enumerate(ins):
        if "row_newbcast" not in ln:
            continue
        ndpp += 1
        ops = [t.strip() for t in ln.split(None, 1)[1].split(",")]
        src = regs(ops[1].split()[0])                          # the DPP source: first source operand
        for back in (1, 2):
            if i - back < 0:
                continue
            p = ins[i - back]
            if p.startswith("v_") and "row_newbcast" not in p:
                dst = regs(p.split(None, 1)[1].split(",")[0].strip())
                if dst & src:
                    bad.append((cfg, p, ln))
            elif p.startswith("s_nop"):
                break
    return ndpp, bad


if __name__ == "__main__":
    total, bad = 0, []
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        for n, b in ex.map(one, B.configs()):
            total += n
            bad += b
    print("DPP instructions checked:", total, " hazards:", len(bad))
    for cfg, p, ln in bad[:20]:
        print(cfg, "|", p, "|", ln)
    sys.exit(1 if bad else 0)
