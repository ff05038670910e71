#!/usr/bin/env python
"""Opportunistic CPU baseline on the REAL engine the reference uses ("CPU baseline timing" (ii)).

The reference's simulator is pybullet driving pybullet_data/a1/a1.urdf through rlschool; neither ships with
/root/reference and neither is installed in this image, so this script normally prints that pybullet is
unavailable and bench.py's `cpu_baseline` stays the repo's own oracle (kind "port").  Where pybullet IS
installed it times the loop the reference runs per control step -- restated from minitaur.py:242-260 (sub-step
loop), :904-947 (PD -> TORQUE_CONTROL), :1151-1170 (state read-back) -- with this repo's own harness (no
reference code is imported), and can dump joint/base trajectories for a parity look at the oracle.

The pybullet leg is NOT exercised in this repository's CI (there is no pybullet here); the dump FORMAT and the comparison
against the oracle are (tests/test_oracle_physics.py): `--dump f.npy` here on any box with pybullet, then
`python tests/pybullet_compare.py f.npy` anywhere prints the measured gap (the comparison uses the oracle, so it lives under
tests/).  usage: pybullet_baseline.py [--steps 400] [--dump F]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(steps, action_repeat=13, dt=0.002, dump=None):
    try:
        import pybullet as p
        import pybullet_data
    except ImportError:
        return None
    from paddlerobotics_amd import a1_model as A
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points, etg_joint_action
    cid = p.connect(p.DIRECT)
    p.setAdditionalSearchPath(pybullet_data.getDataPath())
    p.setGravity(0, 0, -10)
    p.setTimeStep(dt)
    p.loadURDF("plane.urdf")
    robot = p.loadURDF("a1/a1.urdf", [0, 0, 0.32])
    names = ["%s_%s_joint" % (leg, j) for leg in ("FR", "FL", "RR", "RL") for j in ("hip", "upper", "lower")]
    by_name = {p.getJointInfo(robot, i)[1].decode(): i for i in range(p.getNumJoints(robot))}
    motors = [by_name[n] for n in names]
    for i in range(p.getNumJoints(robot)):
        p.changeDynamics(robot, i, linearDamping=0, angularDamping=0)
    p.setJointMotorControlArray(robot, motors, p.VELOCITY_CONTROL, forces=[0.0] * 12)   # default motors off
    pose = A.INIT_MOTOR_ANGLES.copy()
    for m, q in zip(motors, pose):
        p.resetJointState(robot, m, q, 0.0)
    kp, kd = np.full(12, 100.0), np.array([1.0, 2.0, 2.0] * 4)

    def substeps(qdes, n):
        for _ in range(n):
            js = p.getJointStates(robot, motors)
            q = np.array([s[0] for s in js]); qd = np.array([s[1] for s in js])
            tau = -kp * (q - qdes) - kd * qd
            p.setJointMotorControlArray(robot, motors, p.TORQUE_CONTROL, forces=tau.tolist())
            p.stepSimulation()

    substeps(pose, 500)                                             # settle, a1.py:289-304
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w, b, _ = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    traj = []
    t0 = time.perf_counter()
    for k in range(steps):
        qdes = pose + etg_joint_action(layer, w, b, (k + 1) * 0.026)    # ETG -> IK residual, a1.py:97-110
        substeps(qdes, action_repeat)
        if dump is not None:
            pos, orn = p.getBasePositionAndOrientation(robot)
            traj.append(list(pos) + list(orn) + [s[0] for s in p.getJointStates(robot, motors)])
    elapsed = time.perf_counter() - t0
    p.disconnect(cid)
    if dump is not None:
        np.save(dump, np.array(traj))
    return steps / elapsed


DUMP_FORMAT = """trajectory dump, numpy .npy, float64 [steps, 19]: one row per CONTROL step (13 ticks of 2 ms), taken after the step:
  [0:3] base position (world, m) | [3:7] base orientation quaternion xyzw | [7:19] motor angles, order FR hip/upper/lower,
  FL, RR, RL (rad).  Scenario = BASELINE configs[0]: a1.urdf at (0, 0, 0.32), gravity (0, 0, -10), joints at
  INIT_MOTOR_ANGLES, 500 settle ticks holding that pose, then per step q_des = pose + ETG(t = (k+1) 0.026 s) with the
  Opt_with_points prior (ETG_T 0.5, Footheight 0.1, Steplength 0.05), PD kp = 100, kd = (1, 2, 2) (a1.py:75-80) applied as
  torques every tick, foot friction as in the URDF."""


if __name__ == "__main__":
    ap = argparse.ArgumentParser(epilog=DUMP_FORMAT, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--dump", type=str, default=None, help="(needs pybullet) save the trajectory, format below")
    a = ap.parse_args()
    rate = run(a.steps, dump=a.dump)
    if rate is None:
        print(json.dumps({"pybullet": "unavailable", "note": "baseline is the repo's CPU restatement (oracle/)"}))
    else:
        print(json.dumps({"pybullet": "available", "env_steps_per_s_single_process": rate, "steps": a.steps}))
