#!/usr/bin/env python
"""Why the reference's recorded gait yaws on the per-direction friction pyramid -- CPU oracle only.

The gait of gait_action_list_ETG_exp.npy (fitted as exp_w / exp_b, tests/golden/etg.npz) is replayed OPEN LOOP for its 600
control steps (env_test.py:51-54 records 600 rows unconditionally; nothing in the reference says the robot walks them).  For
each foot-friction coefficient and both friction models: distance, final lateral offset and yaw, the yaw every 100 steps, and
the fraction of loaded feet whose friction impulse sits ON the cone / pyramid boundary at the end of a control step = sliding.
Then the left / right asymmetries of the model are removed one at a time at the reference's default friction.
Writes the table to stdout (committed as profiles/r04_yaw_rootcause.txt)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd import a1_model as A  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "etg.npz"))


def yaw_of(s):
    x, y, z, w = s[3:7]
    return float(np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)))


def walk(mu=0.2, fm=0, model=None):
    orc = OracleSim(A.default_config(1, friction_model=fm), model=model)
    dyn = A.default_dynamic_row()[None].copy()
    dyn[0, 1] = mu
    orc.set_params(dyn=dyn, etg_w=g["exp_w"], etg_b=g["exp_b"])
    orc.reset()
    x0 = orc.get_state()[0, 0]
    yaws, on_edge, loaded = [], 0, 0
    for k in range(600):
        orc.step(np.zeros((1, 12)))
        lam = orc.get_lambda()[0].reshape(4, 3)
        for ln, l1, l2 in lam:
            if ln > 1e-9:
                loaded += 1
                r = np.hypot(l1, l2) if fm == 0 else max(abs(l1), abs(l2))
                on_edge += r >= 0.98 * mu * ln
        if k % 100 == 99:
            yaws.append(yaw_of(orc.get_state()[0]))
    s = orc.get_state()[0]
    return s[0] - x0, s[1], yaws, on_edge / max(loaded, 1)


def variant(sym_foot=False, sym_com=False, diag_trunk=False):
    m = A.default_model()
    if sym_foot:
        for j, v in enumerate([0.18, -0.14, -0.23, 0.18, 0.14, -0.23, -0.18, -0.14, -0.23, -0.18, 0.14, -0.23]):
            m.base_foot[j] = v
    if sym_com:
        for leg in range(4):
            m.hip_origin[leg][1] = -0.047 if leg % 2 == 0 else 0.047
    if diag_trunk:
        for k in (3, 4, 5):
            m.trunk.inertia[k] = 0.0
    return m


print("friction coefficient x friction model (0 = implicit cone / disc: pybullet's default; 1 = per-direction pyramid)")
print("%-6s %-5s %9s %9s %9s   %-52s %s" % ("mu", "model", "dist m", "y m", "yaw rad", "yaw after 100 .. 600 steps", "loaded feet on the friction boundary"))
for mu in (0.2, 0.4, 0.7, 1.0):
    for fm in (0, 1):
        d, y, yaws, edge = walk(mu, fm)
        print("%-6.1f %-5d %9.2f %9.2f %9.3f   %-52s %.0f %%" % (mu, fm, d, y, yaws[-1], " ".join("%+.3f" % v for v in yaws), 100 * edge))
print()
print("model asymmetries removed one at a time, mu = 0.2 (param2dynamic_dict(zeros), train.py:116)")
for name, kw in (("as shipped (BASE_FOOT y = -0.15 / 0.148 / -0.14 / 0.135, COM offset, trunk products of inertia)", {}),
                 ("symmetric BASE_FOOT (y = -+0.14)", dict(sym_foot=True)),
                 ("hips at y = -+0.047 (no lateral COM offset)", dict(sym_com=True)),
                 ("all three symmetric", dict(sym_foot=True, sym_com=True, diag_trunk=True))):
    for fm in (0, 1):
        d, y, yaws, edge = walk(0.2, fm, variant(**kw))
        print("%-100s model %d: dist %.2f m, y %+.2f m, yaw %+.3f rad" % (name, fm, d, y, yaws[-1]))
