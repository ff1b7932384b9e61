"""Build-time guards that need no GPU (hipcc cross-compiles gfx950 here)."""
import concurrent.futures
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spills():
    spec = importlib.util.spec_from_file_location("pa_spills", os.path.join(ROOT, "tools", "spills.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# (cell degree, face degree, quadrature, min lanes) of the BASELINE.json configurations, and the stabilization
# / lanes-per-cell of the instance each one runs
BASELINE_INSTANCES = [
    ((2, 1, 0, 16), (2, 16)),      # convergence_test / 1024^2 k=1: fancy
    ((3, 2, 0, 32), (2, 32)),      # north-star 1024^2 k=2: fancy
    ((4, 3, 0, 32), (2, 32)),      # 2048^2 k=3: fancy
    ((0, 1, 0, 16), (2, 16)),      # obstacle pair: dense fancy
    ((2, 1, 1, 16), (1, 16)),      # cuthho k=1: fan quadrature, naive
    ((3, 2, 1, 32), (1, 32)),      # cuthho k=2
]


def test_local_operator_kernels_of_the_baseline_configurations_do_not_spill():
    """A spilled register in the cooperative kernel is reloaded with a scratch load, which waits for every outstanding
    global store of the wavefront (vmcnt(0)): measured +3 % ... +45 %.  The lc-only kernel of every
    BASELINE.json configuration must compile without spills or scratch at the occupancy _build.py asks for; the pre-pass
    may spill only where _build.py says so (k = 3)."""
    sp = _spills()
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(sp.one, [c for c, _ in BASELINE_INSTANCES]))
    for (cfg, (stab, lanes)), rows in zip(BASELINE_INSTANCES, results):
        hit = [r for r in rows if r[0] == (cfg[0], cfg[1], cfg[2], stab, lanes) and r[1] == "lc"]
        assert len(hit) == 1, (cfg, rows)
        _, _, vgpr, spilled, scratch = hit[0]
        assert spilled == 0 and scratch == 0, (cfg, hit[0])
        pre = [r for r in rows if r[0] == (cfg[0], cfg[1], cfg[2], stab, lanes) and r[1] == "pre"]
        assert len(pre) == 1
        if cfg[:3] != (4, 3, 0):
            assert pre[0][3] == 0 and pre[0][4] == 0, (cfg, pre[0])
        # the condensed-mode instance of the same configuration: no spills either
        cond = [r for r in rows if r[0] == (cfg[0], cfg[1], cfg[2], stab, lanes) and r[1] == "cond"]
        assert len(cond) == 1, (cfg, rows)
        assert cond[0][3] == 0 and cond[0][4] == 0, (cfg, cond[0])


def test_dpp_operands_have_no_valu_write_hazard():
    """The substitutions take the packed factor through DPP operands written in inline assembly (hho_device.hpp, Cfg::DPPFWD),
    which the compiler's hazard recognizer does not see: a VALU write of the DPP source register within the two instructions
    before the DPP read would need wait states.  tools/dpp_lint.py checks the generated ISA of every instance: wait states counted
    (s_nop N = N + 1), branches into a label followed backwards, VALU writes of EXEC within 5 wait states of a DPP instruction."""
    spec = importlib.util.spec_from_file_location("pa_dpp_lint", os.path.join(ROOT, "tools", "dpp_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(lint.one, lint.B.configs()))      # EVERY instance of pa_configs.def, not the headline ones only
    assert sum(n for n, _ in results) > 0                     # the instances do use the DPP form
    assert not [b for _, bad in results for b in bad], [b for _, bad in results for b in bad][:5]


def test_dpp_lint_sees_the_hazards_it_is_there_for():
    """synthetic listings: a write one instruction before the read, `s_nop 0` (ONE wait state) between them, a write at the end of a
    loop body feeding the read at the loop head, a v_cmpx three instructions before a DPP instruction -- and the clean forms of each"""
    spec = importlib.util.spec_from_file_location("pa_dpp_lint", os.path.join(ROOT, "tools", "dpp_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    dpp = "v_fmac_f64_dpp v[0:1], v[4:5], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf"
    n, bad = lint.lint_listing(["v_mov_b32_e32 v4, v9", dpp])
    assert n == 1 and len(bad) == 1
    assert len(lint.lint_listing(["v_mov_b32_e32 v4, v9", "s_nop 0", dpp])[1]) == 1            # one wait state is not two
    assert len(lint.lint_listing(["v_mov_b32_e32 v4, v9", "s_nop 1", dpp])[1]) == 0
    assert len(lint.lint_listing(["v_mov_b32_e32 v4, v9", "v_add_u32 v7, v7, v8", "s_mov_b32 s0, 0", dpp])[1]) == 0
    loop = [".LBB0_1:", dpp, "v_add_f64 v[6:7], v[6:7], v[0:1]", "v_mov_b32_e32 v5, v9", "s_cbranch_scc1 .LBB0_1"]
    assert len(lint.lint_listing(loop)[1]) == 1                                                # back edge: write, branch, read
    loop_ok = [".LBB0_1:", dpp, "v_mov_b32_e32 v5, v9", "v_add_f64 v[6:7], v[6:7], v[0:1]", "s_nop 0", "s_cbranch_scc1 .LBB0_1"]
    assert len(lint.lint_listing(loop_ok)[1]) == 0
    assert len(lint.lint_listing(["v_cmpx_gt_u32_e32 16, v0", "v_mov_b32_e32 v20, v9", "v_mov_b32_e32 v21, v9", dpp])[1]) == 1
    assert len(lint.lint_listing(["v_cmpx_gt_u32_e32 16, v0", "s_nop 4", dpp])[1]) == 0


# (instance, mangled-name key of its lc-only kernel, budget of the cell loop: vector instructions, matrix instructions, LDS instructions)
LOOP_BUDGETS = [
    ((2, 1, 0, 16), "CfgILi2ELi1ELi0ELi2ELi16ELi0EEELi0E", 220, 16, 58),      # measured at the end of round 2: 200 / 16 / 50
    ((3, 2, 0, 32), "CfgILi3ELi2ELi0ELi2ELi32ELi0EEELi0E", 405, 20, 116),     # 372 / 20 / 105
    ((4, 3, 0, 32), "CfgILi4ELi3ELi0ELi2ELi32ELi0EEELi0E", 660, 40, 165),     # 604 / 40 / 150
]


def test_cell_loop_of_the_headline_kernels_stays_within_its_instruction_budget():
    """The cooperative kernel answers to its instruction count (half of the vector instructions of a pass were
    index arithmetic, moves and selects before they were removed), and part of that count is in the compiler's hands: the
    per-pass opaque lane index, the profiling branches that keep per-lane values from being hoisted, the packed per-lane index
    words.  A census of the generated ISA of the cell loop (tools/isa_stages.py on a -DPA_MARKERS listing, static counts over
    both arms of the profiling branches) must stay within ~10 % of what was measured; the matrix instructions exactly."""
    spec = importlib.util.spec_from_file_location("pa_isa_stages", os.path.join(ROOT, "tools", "isa_stages.py"))
    isa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(isa)
    import tempfile

    def one(job):
        (cd, fd, q, gmin), key, _, _, _ = job
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "k.s")
            isa.build_listing(cd, fd, q, gmin, path)
            cnt, res = isa.census(path, key)
        return isa.loop_totals(cnt), res

    with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
        results = list(ex.map(one, LOOP_BUDGETS))
    for (cfg, key, vmax, mfma, ldsmax), (tot, res) in zip(LOOP_BUDGETS, results):
        assert res.get("scratch") == 0, (cfg, res)
        assert tot["scratch"] == 0, (cfg, dict(tot))
        assert tot["mfma"] == mfma, (cfg, dict(tot))
        assert tot["valu64"] + tot["valu"] <= vmax, (cfg, dict(tot))
        assert tot["lds"] <= ldsmax, (cfg, dict(tot))
