#!/usr/bin/env python3
"""ISA check for the DPP operands of the substitutions (hho_device.hpp, Cfg::DPPFWD): a VALU write of a register needs 2 wait
states before a DPP instruction reads it as its DPP source.  The inline assembly is invisible to the compiler's hazard
recognizer, so every instance is compiled to assembly and each  row_newbcast  instruction is checked against the two
instructions before it -- counting wait states (s_nop N = N + 1), following the branches that target a label in front of the DPP
instruction (a write at the end of a loop body, a read at its head) and looking for VALU writes of EXEC within 5 wait states.
Exit status 1 and a listing if any is found."""
import concurrent.futures, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from proton_amd import _build as B


def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def wait_states(ins):
    """wait states an instruction puts between the one before it and the one after it: s_nop N counts N + 1, anything else 1"""
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def predecessors(ins, labels, i, need):
    """Yield (instruction, wait states between it and position i) for every instruction that can execute within `need` wait states
    before position i -- along the fall-through path AND, at a label, along every branch that targets it (a VALU write at the
    end of a loop body feeds a DPP read at the loop head).  ins: list of (kind, text), kind 'i' instruction / 'l' label."""
    seen = set()
    stack = [(i - 1, 0)]
    while stack:
        j, ws = stack.pop()
        while j >= 0 and ws < need:
            kind, text = ins[j]
            if kind == "l":
                for src in labels.get(text, ()):              # the branches that jump here: continue in front of each (the branch itself is a wait state)
                    if (src, ws) not in seen:
                        seen.add((src, ws))
                        stack.append((src - 1, ws + 1))
                j -= 1
                continue
            yield text, ws
            ws += wait_states(text)
            if text.startswith(("s_branch", "s_endpgm", "s_setpc")):      # unconditional: nothing falls through from above
                break
            j -= 1


def one(cfg):
    """-> (DPP instructions checked, hazards).  Hazards looked for (gfx90a+ data hazards the compiler would pad for its own instructions):
       * VALU write of a VGPR, then a DPP read of it as the DPP source: 2 wait states;
       * VALU write of EXEC (v_cmpx*, v_readlane/readfirstlane do not write it), then any DPP instruction: 5 wait states."""
    cd, fd, q, gmin = cfg
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [B.hipcc()] + B.FLAGS + ["-DPA_CD=%d" % cd, "-DPA_FD=%d" % fd, "-DPA_QUAD=%d" % q, "-DPA_GMIN=%d" % gmin] + \
            B.PER_CONFIG_FLAGS.get((cd, fd, q), []) + ["--cuda-device-only", "-S", "-o", out, os.path.join(B.CSRC, "hho_inst.hip")]
        subprocess.run(cmd, check=True, capture_output=True)
        lines = [ln.strip() for ln in open(out)]
    return lint_listing(lines, cfg)


def lint_listing(lines, cfg=None):
    """the checks of one() on the lines of an assembly listing"""
    ins = []
    for ln in lines:
        if not ln or ln.startswith((";", "//")):
            continue
        if ln.endswith(":") and not ln.startswith("."):
            ins.append(("l", ln[:-1]))
        elif re.match(r"\.LBB\S+:$", ln):
            ins.append(("l", ln[:-1]))
        elif not ln.startswith("."):
            ins.append(("i", ln.split(";")[0].strip()))
    labels = {}
    for i, (kind, text) in enumerate(ins):
        if kind == "i" and text.startswith(("s_cbranch", "s_branch")):
            labels.setdefault(text.split()[-1], []).append(i)
    bad, ndpp = [], 0
    for i, (kind, ln) in enumerate(ins):
        if kind != "i" or "row_newbcast" not in ln:
            continue
        ndpp += 1
        ops = [t.strip() for t in ln.split(None, 1)[1].split(",")]
        src = regs(ops[1].split()[0])                          # the DPP source: first source operand
        for p, ws in predecessors(ins, labels, i, 5):
            if p.startswith("v_cmpx"):
                bad.append((cfg, p, ln, "EXEC written %d wait states before a DPP instruction (5 needed)" % ws))
            elif ws < 2 and p.startswith("v_") and "row_newbcast" not in p and not p.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                dst = regs(p.split(None, 1)[1].split(",")[0].strip())
                if dst & src:
                    bad.append((cfg, p, ln, "DPP source written %d wait states before the read (2 needed)" % ws))
    return ndpp, bad


if __name__ == "__main__":
    total, bad = 0, []
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        for n, b in ex.map(one, B.configs()):
            total += n
            bad += b
    print("DPP instructions checked:", total, " hazards:", len(bad))
    for cfg, p, ln, why in bad[:20]:
        print(cfg, "|", p, "|", ln, "|", why)
    sys.exit(1 if bad else 0)
