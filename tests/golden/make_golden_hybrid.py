#!/usr/bin/env python
"""Golden vectors for the HYBRID motor command (run in the build container only; /root/reference is read).

Imports the reference's own motor model, deployment/robots/laikago_motor.py (LaikagoMotorModel.convert_to_torque,
HYBRID branch :152-167), and records torques for random 60-vectors 12 x (q_des, kp, qd_des, kd, tau_ff).
Output: tests/golden/pd_hybrid.npz (cmd, q, qd, tau)."""
import os
import sys

import numpy as np

REF = "/root/reference/QuadrupedalRobots/ETGRL/"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF + "deployment")
from robots import laikago_motor, robot_config  # noqa: E402

rng = np.random.default_rng(11)
mm = laikago_motor.LaikagoMotorModel(kp=100.0, kd=1.0, motor_control_mode=robot_config.MotorControlMode.HYBRID)
n = 24
cmd = np.zeros((n, 60))
cmd[:, 0::5] = rng.normal(size=(n, 12))            # q_des
cmd[:, 1::5] = rng.uniform(20, 200, size=(n, 12))  # kp
cmd[:, 2::5] = rng.normal(size=(n, 12)) * 2        # qd_des
cmd[:, 3::5] = rng.uniform(0, 5, size=(n, 12))     # kd
cmd[:, 4::5] = rng.normal(size=(n, 12)) * 3        # tau_ff
q = rng.normal(size=(n, 12))
qd = rng.normal(size=(n, 12)) * 3
tau = np.stack([mm.convert_to_torque(cmd[i], q[i], qd[i], qd[i], robot_config.MotorControlMode.HYBRID)[0] for i in range(n)])
np.savez(os.path.join(OUT, "pd_hybrid.npz"), cmd=cmd, q=q, qd=qd, tau=tau)
print("wrote pd_hybrid.npz", tau.shape, float(np.abs(tau).max()))
