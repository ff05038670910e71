#!/usr/bin/env python
"""Golden vectors for the motor model's strength ratios and torque limit (run in the build container only; /root/reference
is read).  Imports the reference's own motor model, deployment/robots/laikago_motor.py (LaikagoMotorModel.set_strength_ratios
:67-76, convert_to_torque :103-175), and records torques for random states with per-motor strength ratios, with and without
torque limits, in POSITION and TORQUE mode.  Output: tests/golden/pd_strength.npz."""
import os
import sys

import numpy as np

REF = "/root/reference/QuadrupedalRobots/ETGRL/"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF + "deployment")
import collections
import collections.abc
collections.Sequence = collections.abc.Sequence   # laikago_motor.py:62 uses the pre-3.10 alias
from robots import laikago_motor, robot_config  # noqa: E402

rng = np.random.default_rng(17)
n = 16
kp = np.full(12, 100.0)
kd = np.array([1.0, 2.0, 2.0] * 4)
qdes, q, qd = rng.normal(size=(n, 12)), rng.normal(size=(n, 12)), rng.normal(size=(n, 12)) * 3
strength = rng.uniform(0.3, 1.0, size=(n, 12))
out = dict(qdes=qdes, q=q, qd=qd, kp=kp, kd=kd, strength=strength, limit=np.array(33.5))
for name, limit in (("tau", None), ("tau_limited", 33.5)):
    mm = laikago_motor.LaikagoMotorModel(kp=kp, kd=kd, torque_limits=limit, motor_control_mode=robot_config.MotorControlMode.POSITION)
    rows = []
    for i in range(n):
        mm.set_strength_ratios(strength[i])
        rows.append(mm.convert_to_torque(qdes[i], q[i], qd[i], qd[i], robot_config.MotorControlMode.POSITION)[0])
    out[name] = np.stack(rows)
mm = laikago_motor.LaikagoMotorModel(kp=kp, kd=kd, torque_limits=5.0, motor_control_mode=robot_config.MotorControlMode.TORQUE)
rows = []
for i in range(n):
    mm.set_strength_ratios(strength[i])
    rows.append(mm.convert_to_torque(qdes[i] * 20, q[i], qd[i], qd[i], robot_config.MotorControlMode.TORQUE)[0])   # passes the limit: no clip in TORQUE mode
out["tau_torque_mode"] = np.stack(rows)
np.savez(os.path.join(OUT, "pd_strength.npz"), **out)
print("wrote pd_strength.npz", out["tau"].shape, float(np.abs(out["tau_limited"]).max()), float(np.abs(out["tau_torque_mode"]).max()))
