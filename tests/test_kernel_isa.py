"""CPU suite: the code generation of the hot kernels is pinned ("nothing pins the hot loops ... a toolchain bump
can silently move the headline").  The step time of these kernels is (instructions per wave) x ~4.4 cycles with one wave per
SIMD, and their register allocation sits close to where hipcc starts spilling; profiles/r06_isa_baseline.json records, for the
headline kernels, registers, scratch and the static instruction mix of the build the round's numbers were measured with.
This test disassembles the library that was just built (llvm-objdump on its gfx950 code object) and compares:
  * no scratch (spills) in the headline 16-lane kernels, registers within +8 of the recorded allocation;
  * VALU / DPP-modified / s_nop / transcendental instruction counts within 3 % (s_nop 10 %) of the recorded ones.
A deliberate change of the kernels regenerates the file: python tools/kernel_isa_stats.py --write-baseline."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = os.path.join(ROOT, "profiles", "r06_isa_baseline.json")


def test_hot_kernel_code_generation_matches_the_recorded_baseline():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_isa_stats as K
    from paddlerobotics_amd import build
    if not os.path.exists(K.LLVM + "/llvm-objdump"):
        pytest.skip("no llvm-objdump in this image")
    so = build.build()
    base = json.load(open(BASE))
    got = K.stats(so, list(base["kernels"].keys()))
    assert set(got) == set(base["kernels"]), sorted(set(base["kernels"]) - set(got))
    report = []
    for sym, want in base["kernels"].items():
        g = got[sym]
        if want["scratch"] == 0:
            assert g["scratch"] == 0, "%s spills (%d B of scratch)" % (sym, g["scratch"])
        else:   # (the closed-loop kernel parks ~115 dwords per lane around its policy tile -- none inside the tick: DESIGN.md section 4)
            assert g["scratch"] <= want["scratch"] + 64, "%s: scratch grew to %d B" % (sym, g["scratch"])
        assert g["vgpr"] <= want["vgpr"] + 8 and g["agpr"] <= want["agpr"] + 16, (sym, g["vgpr"], g["agpr"])
        for key, tol in (("valu", 0.03), ("dpp", 0.03), ("trans", 0.03), ("s_nop", 0.10), ("all", 0.03)):
            if abs(g.get(key, 0) - want[key]) > tol * want[key] + 2:
                report.append("%s: %s %d, recorded %d" % (sym, key, g.get(key, 0), want[key]))
    assert not report, "code generation moved (regenerate profiles/r06_isa_baseline.json if intended):\n" + "\n".join(report)


def test_no_accumulator_is_its_own_broadcast_source():
    """The multi-instruction asm blocks of the sweeps accumulate into read-write operands and read their broadcast sources LATER
    in the block.  An input that holds the same value as an accumulator's start (the second rows' impulses and three of
    row2_velocity's partial sums all start at 0) may be given the accumulator's register unless the accumulator is early-clobber
    ("+&v"): the block then reads its own partial sum as the broadcast source -- `v_fmac_f32_dpp vN, vN, ...`.  Round 5 shipped
    that for one build (the non-plain kernels were 1e-2 rad off the oracle on the GPU while the emulation was right); no code of
    ours wants such an instruction, so none may exist in the library."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_isa_stats as K
    from paddlerobotics_amd import build
    if not os.path.exists(K.LLVM + "/llvm-objdump"):
        pytest.skip("no llvm-objdump in this image")
    tmp, cos = K.code_objects(build.build())
    bad = []
    for co in cos:
        sym = None
        for line in K.disassemble(co).splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                sym = m.group(1)
            elif re.search(r"v_fmac_f32_dpp v(\d+), v\1,", line):
                bad.append((sym, line.strip()))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    assert not bad, bad[:8]


def test_no_dpp_operand_is_read_inside_its_writers_wait_states():
    """A VALU write needs 2 wait states before a DPP instruction reads the register as its DPP operand (and a transcendental's
    result 1 before any VALU instruction reads it); the asm blocks of the
    sweeps pad by hand (an independent instruction or s_nop in each slot) and the compiler cannot check them, nor the values it
    hands them.  tools/kernel_isa_stats.dpp_hazards walks the disassembly of every kernel of the library; the checker itself is
    checked on a synthetic listing first."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_isa_stats as K
    from paddlerobotics_amd import build
    c = "  // 000000001000: 00000000"
    dpp = " row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    listing = "\n".join(["0000 <k>:",
                         "\tv_mul_f32_e32 v1, v2, v3" + c, "\tv_add_f32_e32 v9, v2, v3" + c, "\tv_fmac_f32_dpp v4, v1, v5" + dpp + c,   # 1 slot: hazard
                         "\tv_mul_f32_e32 v1, v2, v3" + c, "\ts_nop 1" + c, "\tv_fmac_f32_dpp v4, v1, v5" + dpp + c,                   # 2 slots: fine
                         "\tv_pk_mul_f32 v[6:7], v[2:3], v[4:5]" + c, "\tv_mov_b32_dpp v8, v7 quad_perm:[0,0,0,0] row_mask:0xf" + c,   # pair write: hazard
                         "\tv_mul_f32_e32 v1, v2, v3" + c, "\tv_add_f32_e32 v9, v2, v3" + c, "\tv_add_f32_e32 v10, v2, v3" + c,
                         "\tv_add_f32_dpp v1, v1, v1 quad_perm:[0,2,1,3] row_mask:0xf" + c,                                          # 2 instructions: fine
                         "\tv_rsq_f32_e32 v11, v2" + c, "\tv_mul_f32_e32 v3, v4, v11" + c,                                               # transcendental read at once: hazard
                         "\tv_rsq_f32_e32 v11, v2" + c, "\ts_nop 0" + c, "\tv_mul_f32_e32 v3, v4, v11" + c])                             # 1 wait state: fine
    n, bad = K.dpp_hazards(listing)
    assert n == 4 and len(bad) == 3, (n, bad)
    if not os.path.exists(K.LLVM + "/llvm-objdump"):
        pytest.skip("no llvm-objdump in this image")
    import shutil
    tmp, cos = K.code_objects(build.build())
    total, bad = 0, []
    for co in cos:
        n, b = K.dpp_hazards(K.disassemble(co))
        total += n
        bad += b
    shutil.rmtree(tmp, ignore_errors=True)
    assert total > 20000, total          # (the 16-lane kernels alone hold ~2000 DPP instructions each)
    assert not bad, bad[:8]
