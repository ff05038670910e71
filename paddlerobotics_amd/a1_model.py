"""A1 robot model constants and the ctypes mirrors of include/etgsim.h.

Kinematic constants are the reference's (QuadrupedalRobots/ETGRL, file:line):
  INIT_POSITION            deployment/robots/a1.py:52
  COM_OFFSET, HIP_OFFSETS  deployment/robots/a1.py:70-73
  INIT_MOTOR_ANGLES        deployment/robots/a1.py:83
  l_up, l_low, l_hip       deployment/robots/a1.py:98-100
  ETG_mean / ETG_std       deployment/envs/EnvWrapper.py:50-55
  base_foot                pinned by ETGRL/gait_action_list_ETG_exp.npy
Inertial constants are RECALLED from pybullet_data/a1/a1.urdf, which is absent
from the reference tree: they are this repo's definition of the
robot, not a checked copy.
"""
import ctypes as C
import math

import numpy as np

NUM_LEGS = 4
NUM_MOTORS = 12
OBS_DIM = 49
STATE_DIM = 37
DYN_DIM = 48
RBF_H = 20
INFO_DIM = 64
EXTRA_DIM = 84
EXTRA_SLICES = {"ETG_obs": (0, 20), "footpose": (20, 32), "dynamic_vec": (32, 80), "force_vec": (80, 83)}

INFO_SLICES = {
    "torso": (0, 1), "feet": (1, 2), "up": (2, 3), "tau": (3, 4), "stand": (4, 5),
    "badfoot": (5, 6), "footcontact": (6, 7), "done": (7, 8), "velx": (8, 9),
    "ETG_act": (9, 21), "joint_angle": (21, 33), "obs-IMU": (33, 39),
    "FootContactSensor": (39, 43), "real_action": (43, 55), "base": (55, 58),
    "rpy": (58, 61), "energy": (61, 62), "steps": (62, 63), "solver_sweeps": (63, 64),
}
INFO_SWEEPS = 63   # ETG_INFO_SWEEPS

# reward-term order of EtgConfig.reward_w
REWARD_KEYS = ("torso", "feet", "up", "tau", "stand", "badfoot", "footcontact", "done")
# train.py:481-487 defaults (+ done weight 1)
DEFAULT_REWARD_PARAM = {"torso": 1.5, "feet": 0.3, "up": 0.6, "tau": 0.07, "stand": 0.0,
                        "badfoot": 0.1, "footcontact": 0.1, "done": 1.0}


class EtgLink(C.Structure):
    _fields_ = [("mass", C.c_double), ("com", C.c_double * 3), ("inertia", C.c_double * 6)]


class EtgRobotModel(C.Structure):
    _fields_ = [
        ("trunk", EtgLink),
        ("hip", EtgLink * 4), ("thigh", EtgLink * 4), ("calf", EtgLink * 4), ("foot", EtgLink * 4),
        ("hip_origin", (C.c_double * 3) * 4),
        ("thigh_y", C.c_double * 4),
        ("upper_len", C.c_double), ("lower_len", C.c_double), ("foot_radius", C.c_double),
        ("init_pos", C.c_double * 3),
        ("pose_ori", C.c_double * 12),
        ("base_foot", C.c_double * 12),
        ("etg_mean", C.c_double * 12),
        ("etg_std", C.c_double * 12),
    ]


class EtgConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("action_repeat", C.c_int32), ("settle_ticks", C.c_int32),
        ("solver_iters", C.c_int32), ("enable_action_interp", C.c_int32),
        ("enable_action_filter", C.c_int32), ("obs_normal", C.c_int32), ("terrain", C.c_int32),
        ("sim_dt", C.c_double), ("erp", C.c_double), ("contact_margin", C.c_double),
        ("warmstart", C.c_double), ("torque_limit", C.c_double),
        ("etg_T", C.c_double), ("etg_T2", C.c_double), ("etg_amp", C.c_double),
        ("etg_sigma_sq", C.c_double), ("etg_phase", C.c_double * 2), ("etg_dt", C.c_double),
        ("reward_w", C.c_double * 8), ("reward_p", C.c_double), ("vel_d", C.c_double),
        ("filter_b", C.c_double * 3), ("filter_a", C.c_double * 3),
        ("hf_nx", C.c_int32), ("hf_ny", C.c_int32),
        ("hf_cell", C.c_double), ("hf_x0", C.c_double), ("hf_y0", C.c_double),
        ("lanes_per_robot", C.c_int32),
        ("hf_bands", C.c_int32),
        ("motor_mode", C.c_int32),
        ("clip_motor_commands", C.c_double),
        ("body_contacts", C.c_int32),
        ("knee_radius", C.c_double),
        ("enable_etg", C.c_int32),
        ("joint_limits", C.c_int32),
        ("joint_lower", C.c_double * 3), ("joint_upper", C.c_double * 3),
        ("trunk_half", C.c_double * 3),
        ("solver_residual", C.c_double),
        ("friction_model", C.c_int32),
        ("pd_latency", C.c_double),
        ("warmstart_friction", C.c_double),
        ("contact_slop", C.c_double),
        ("foot_restitution", C.c_double),
        ("body_friction", C.c_double),
        ("body_blend", C.c_double),
    ]


# ---- reference kinematic constants -------------------------------------------------
COM_OFFSET = -np.array([0.012731, 0.002186, 0.000515])                    # a1.py:70
HIP_OFFSETS = np.array([[0.183, -0.047, 0.], [0.183, 0.047, 0.],
                        [-0.183, -0.047, 0.], [-0.183, 0.047, 0.]]) + COM_OFFSET  # a1.py:71-73
INIT_POSITION = (0.0, 0.0, 0.32)                                          # a1.py:52
INIT_MOTOR_ANGLES = np.array([0, 0.9, -1.8] * 4, dtype=np.float64)        # a1.py:83
L_HIP = 0.08505                                                           # a1.py:100
L_UP = 0.2                                                                # a1.py:98
L_LOW = 0.2                                                               # a1.py:99
BASE_FOOT = np.array([0.18, -0.15, -0.23, 0.18, 0.148, -0.23,
                      -0.18, -0.14, -0.23, -0.18, 0.135, -0.23])
ETG_MEAN = np.array([2.1505982e-02, 3.6674485e-02, -6.0444288e-02,
                     2.4625482e-02, 1.5869144e-02, -3.2513142e-02, 2.1506395e-02,
                     3.1869926e-02, -6.0140789e-02, 2.4625063e-02, 1.1628972e-02,
                     -3.2163858e-02])                                     # EnvWrapper.py:50-53
ETG_STD = np.array([4.5967497e-02, 2.0340437e-01, 3.7410179e-01, 4.6187632e-02, 1.9441207e-01,
                    3.9488649e-01, 4.5966785e-02, 2.0323379e-01, 3.7382501e-01, 4.6188373e-02,
                    1.9457331e-01, 3.9302582e-01])                        # EnvWrapper.py:54-55
FOOT_RADIUS = 0.02
# joint limits of (hip, thigh, calf): the bounds of ACTION_CONFIG, a1.py:186-195 (= the URDF limits Bullet enforces)
JOINT_LOWER = (-0.802851455917, -1.0471975512, -2.69653369433)
JOINT_UPPER = (0.802851455917, 4.18879020479, -0.916297857297)
# half extents of the trunk's collision box (recalled from pybullet_data/a1/a1.urdf: box 0.267 x 0.194 x 0.114)
TRUNK_HALF = (0.1335, 0.097, 0.057)

# ---- recalled URDF inertials (FR leg; mirrored below) -------------------------------
_TRUNK = dict(mass=4.713, com=(0.0, 0.0, 0.0),
              inertia=(0.01683993, 0.056579028, 0.064713601, 8.3902e-05, 0.000597679, 2.5134e-05))
_HIP_FR = dict(mass=0.696, com=(-0.003311, -0.000635, 3.1e-05),
               inertia=(0.000469246, 0.00080749, 0.000552929, 9.409e-06, -3.42e-07, 4.66e-07))
_THIGH_FR = dict(mass=1.013, com=(-0.003237, 0.022327, -0.027326),
                 inertia=(0.005529065, 0.005139339, 0.001367788, -4.825e-06, 0.000343869, -2.2448e-05))
_CALF = dict(mass=0.166, com=(0.006435, 0.0, -0.107388),
             inertia=(0.002997972, 0.003014022, 3.2426e-05, 0.0, -0.000141163, 0.0))
_FOOT = dict(mass=0.06, com=(0.0, 0.0, 0.0), inertia=(9.6e-06, 9.6e-06, 9.6e-06, 0.0, 0.0, 0.0))


def _mirror(link, mx, my):
    """Mirror a link's inertial frame through the yz-plane (mx) and/or xz-plane (my)."""
    sx = -1.0 if mx else 1.0
    sy = -1.0 if my else 1.0
    cx, cy, cz = link["com"]
    ixx, iyy, izz, ixy, ixz, iyz = link["inertia"]
    return dict(mass=link["mass"], com=(sx * cx, sy * cy, cz),
                inertia=(ixx, iyy, izz, sx * sy * ixy, sx * ixz, sy * iyz))


def _fill_link(dst, src):
    dst.mass = src["mass"]
    for k in range(3):
        dst.com[k] = src["com"][k]
    for k in range(6):
        dst.inertia[k] = src["inertia"][k]


def hip_sign(leg):
    """(-1)**(leg+1): -1 for right legs (FR, RR), +1 for left (a1.py:485)."""
    return (-1.0) ** (leg + 1)


def default_model():
    m = EtgRobotModel()
    _fill_link(m.trunk, _TRUNK)
    for leg in range(4):
        rear = leg >= 2
        left = leg % 2 == 1
        _fill_link(m.hip[leg], _mirror(_HIP_FR, rear, left))
        _fill_link(m.thigh[leg], _mirror(_THIGH_FR, False, left))
        _fill_link(m.calf[leg], _CALF)
        _fill_link(m.foot[leg], _FOOT)
        for k in range(3):
            m.hip_origin[leg][k] = HIP_OFFSETS[leg, k]
        m.thigh_y[leg] = L_HIP * hip_sign(leg)
    m.upper_len, m.lower_len, m.foot_radius = L_UP, L_LOW, FOOT_RADIUS
    for k in range(3):
        m.init_pos[k] = INIT_POSITION[k]
    for j in range(12):
        m.pose_ori[j] = INIT_MOTOR_ANGLES[j]
        m.base_foot[j] = BASE_FOOT[j]
        m.etg_mean[j] = ETG_MEAN[j]
        m.etg_std[j] = ETG_STD[j]
    return m


def butter2_lowpass(fc, fs):
    """Order-2 Butterworth low-pass (b, a), bilinear transform -- what
    scipy.signal.butter(2, fc/(fs/2)) returns (action_filter.py:163-170)."""
    k = math.tan(math.pi * fc / fs)
    q = math.sqrt(0.5)
    norm = 1.0 / (1.0 + k / q + k * k)
    b0 = k * k * norm
    return (b0, 2 * b0, b0), (1.0, 2 * (k * k - 1) * norm, (1 - k / q + k * k) * norm)


SOLVER_ITERS = 50          # pybullet numSolverIterations (setPhysicsEngineParameter default)
SOLVER_RESIDUAL = 1e-7     # pybullet solverResidualThreshold (default)


def solver_preset(name, action_repeat=13):
    """Named contact-solver settings -> dict(solver_iters, solver_residual, friction_model).

    The env layer of the reference (rlschool.quadrupedal) is NOT in /root/reference, so which engine parameters it sets in its
    reset() cannot be read off a file; both candidates are stated here and selectable:
      * "pybullet" (the library default): pybullet's own defaults left untouched -- numSolverIterations 50,
        solverResidualThreshold 1e-7, enableConeFriction 1 (the implicit cone);
      * "locomotion_gym": what the motion_imitation / minitaur locomotion_gym_env lineage (which rlschool's env derives from) is
        recalled to set in reset(): numSolverIterations = int(300 / action_repeat) (23 at 13 sub-steps) and enableConeFriction = 0
        (the per-direction friction pyramid); the residual exit stays pybullet's 1e-7.
    ASSUMPTION, stated as such in DESIGN.md section 2: the default is "pybullet"; re-check against rlschool's source when it is at hand."""
    if name in (None, "pybullet"):
        return dict(solver_iters=SOLVER_ITERS, solver_residual=SOLVER_RESIDUAL, friction_model=0)
    if name == "locomotion_gym":
        return dict(solver_iters=max(1, 300 // int(action_repeat)), solver_residual=SOLVER_RESIDUAL, friction_model=1)
    raise ValueError("solver_preset %r: 'pybullet' or 'locomotion_gym'" % (name,))


def solver_rule(solver_iters=None, solver_residual=None):
    """(sweep cap, residual threshold) of the contact solve.  Nothing given: pybullet's documented defaults -- up to 50
    sweeps per tick with the 1e-7 residual exit.  Only a sweep count: exactly that many sweeps (residual test off)."""
    if solver_iters is None:
        return SOLVER_ITERS, (SOLVER_RESIDUAL if solver_residual is None else float(solver_residual))
    return int(solver_iters), (0.0 if solver_residual is None else float(solver_residual))


def default_config(num_envs, *, action_repeat=13, sim_dt=0.002, settle_ticks=500, solver_iters=None, solver_residual=None,
                   enable_action_interp=False, enable_action_filter=False, normal=1, terrain=0,
                   erp=0.2, contact_margin=0.02, warmstart=0.1, warmstart_friction=0.0, contact_slop=1e-5,
                   foot_restitution=0.0, torque_limit=0.0,
                   ETG_T=0.5, ETG_T2=0.5, etg_amp=0.2, etg_sigma_sq=0.04,
                   etg_phase=(-math.pi / 2, 0.0), reward_param=None, reward_p=5.0, vel_d=0.5,
                   heightfield=None, lanes_per_robot=0, motor_mode=0, clip_motor_commands=0.0,
                   body_contacts=2, body_friction=0.5, knee_radius=0.02, enable_etg=1, joint_limits=1, friction_model=0,
                   pd_latency=0.0, body_blend=1e-3):
    """EtgConfig with the defaults of train.py:296-297,470-487 and SURVEY App. A."""
    c = EtgConfig()
    c.num_envs = int(num_envs)
    c.action_repeat = int(action_repeat)
    c.settle_ticks = int(settle_ticks)
    c.solver_iters, c.solver_residual = solver_rule(solver_iters, solver_residual)
    c.friction_model = int(friction_model)
    c.pd_latency = float(pd_latency)
    c.enable_action_interp = int(bool(enable_action_interp))
    c.enable_action_filter = int(bool(enable_action_filter))
    c.obs_normal = int(bool(normal))
    c.terrain = int(terrain)
    c.sim_dt, c.erp, c.contact_margin = sim_dt, erp, contact_margin
    c.warmstart, c.torque_limit = warmstart, torque_limit
    c.warmstart_friction, c.contact_slop, c.foot_restitution = warmstart_friction, contact_slop, foot_restitution
    c.etg_T, c.etg_T2, c.etg_amp, c.etg_sigma_sq = ETG_T, ETG_T2, etg_amp, etg_sigma_sq
    c.etg_phase[0], c.etg_phase[1] = etg_phase
    c.etg_dt = sim_dt * action_repeat
    rp = dict(DEFAULT_REWARD_PARAM)
    if reward_param:
        rp.update({k: v for k, v in reward_param.items() if k in rp})
    for i, k in enumerate(REWARD_KEYS):
        c.reward_w[i] = float(rp[k])
    c.reward_p, c.vel_d = reward_p, vel_d
    fb, fa = butter2_lowpass(4.0, 1.0 / c.etg_dt)   # action_filter.py:141 (4 Hz)
    for k in range(3):
        c.filter_b[k] = fb[k]
        c.filter_a[k] = fa[k]
    c.lanes_per_robot = int(lanes_per_robot)
    c.motor_mode = int(motor_mode)
    c.clip_motor_commands = float(clip_motor_commands)
    c.body_contacts = int(body_contacts)
    c.body_friction = float(body_friction)
    c.body_blend = float(body_blend)
    c.knee_radius = float(knee_radius)
    c.enable_etg = int(bool(enable_etg))
    c.joint_limits = int(bool(joint_limits))
    for k in range(3):
        c.joint_lower[k], c.joint_upper[k] = JOINT_LOWER[k], JOINT_UPPER[k]
        c.trunk_half[k] = TRUNK_HALF[k]
    if heightfield is not None:
        c.hf_ny, c.hf_nx = heightfield["heights"].shape
        c.hf_cell = heightfield["cell"]
        c.hf_x0, c.hf_y0 = heightfield["origin"]
        c.hf_bands = int(heightfield.get("bands", 1))
        if c.hf_ny % max(c.hf_bands, 1):
            raise ValueError("heightfield rows (%d) must be a multiple of bands (%d)" % (c.hf_ny, c.hf_bands))
    return c


def param2dynamic_dict(params):
    """48-vector in [-1,1] -> physical dynamic_param dict (train.py:112-126)."""
    param = np.clip(np.asarray(params, dtype=np.float64).copy(), -1, 1)
    d = {}
    d['control_latency'] = np.clip(40 + 10 * param[0], 0, 80)
    d['footfriction'] = np.clip(0.2 + 10 * param[1], 0, 20)
    d['basemass'] = np.clip(1.5 + 1 * param[2], 0.5, 3)
    d['baseinertia'] = np.clip(np.ones(3) + 1 * param[3:6], 0.1, 3)
    d['legmass'] = np.clip(np.ones(3) + 1 * param[6:9], 0.1, 3)
    d['leginertia'] = np.clip(np.ones(12) + 1 * param[9:21], 0.1, 3)
    d['motor_kp'] = np.clip(80 * np.ones(12) + 40 * param[21:33], 20, 200)
    d['motor_kd'] = np.clip(np.array([1., 2., 2.] * 4) + param[33:45] * np.array([1, 2, 2] * 4), 0, 5)
    if param.shape[0] > 45:
        d['gravity'] = np.clip(np.array([0, 0, -10]) + param[45:48] * np.array([2, 2, 10]),
                               np.array([-5, -5, -20]), np.array([5, 5, -4]))
    return d


def dynamic_dict_to_row(d):
    """Flatten a dynamic_param dict into the 48-float row etg_set_params takes."""
    g = d.get('gravity', np.array([0.0, 0.0, -10.0]))
    return np.concatenate([
        np.atleast_1d(d['control_latency']), np.atleast_1d(d['footfriction']),
        np.atleast_1d(d['basemass']), np.asarray(d['baseinertia']).reshape(3),
        np.asarray(d['legmass']).reshape(3), np.asarray(d['leginertia']).reshape(12),
        np.asarray(d['motor_kp']).reshape(12), np.asarray(d['motor_kd']).reshape(12),
        np.asarray(g).reshape(3)]).astype(np.float64)


def param2dynamic_rows(params):
    """[n,48] parameter vectors in [-1,1] -> [n,48] physical rows: param2dynamic_dict + dynamic_dict_to_row,
    vectorised over n (tests check it against the per-row functions)."""
    P = np.clip(np.atleast_2d(np.asarray(params, dtype=np.float64)), -1, 1)
    n = P.shape[0]
    if P.shape[1] < 48:
        P = np.concatenate([P, np.zeros((n, 48 - P.shape[1]))], axis=1)
    kd0 = np.array([1., 2., 2.] * 4)
    out = np.empty((n, 48))
    out[:, 0] = np.clip(40 + 10 * P[:, 0], 0, 80)
    out[:, 1] = np.clip(0.2 + 10 * P[:, 1], 0, 20)
    out[:, 2] = np.clip(1.5 + 1 * P[:, 2], 0.5, 3)
    out[:, 3:6] = np.clip(1 + P[:, 3:6], 0.1, 3)
    out[:, 6:9] = np.clip(1 + P[:, 6:9], 0.1, 3)
    out[:, 9:21] = np.clip(1 + P[:, 9:21], 0.1, 3)
    out[:, 21:33] = np.clip(80 + 40 * P[:, 21:33], 20, 200)
    out[:, 33:45] = np.clip(kd0 + P[:, 33:45] * kd0, 0, 5)
    out[:, 45:48] = np.clip(np.array([0, 0, -10]) + P[:, 45:48] * np.array([2, 2, 10]),
                            np.array([-5, -5, -20]), np.array([5, 5, -4]))
    return out


_P2D_CACHE = {}


def _p2d_tables():
    """offset / scale / lower / upper of the 48 affine-then-clip maps of param2dynamic_dict (train.py:112-126), as numpy rows"""
    kd0 = np.array([1., 2., 2.] * 4)
    off = np.concatenate([[40.0, 0.2, 1.5], np.ones(18), 80 * np.ones(12), kd0, [0.0, 0.0, -10.0]])
    scl = np.concatenate([[10.0, 10.0, 1.0], np.ones(18), 40 * np.ones(12), kd0, [2.0, 2.0, 10.0]])
    lo = np.concatenate([[0.0, 0.0, 0.5], 0.1 * np.ones(18), 20 * np.ones(12), np.zeros(12), [-5.0, -5.0, -20.0]])
    hi = np.concatenate([[80.0, 20.0, 3.0], 3 * np.ones(18), 200 * np.ones(12), 5 * np.ones(12), [5.0, 5.0, -4.0]])
    return off, scl, lo, hi


def param2dynamic_rows_torch(params):
    """the same mapping for a [n,48] tensor on any device (random dynamics drawn on the GPU: no host RNG, no upload): every
    entry is clip(offset + scale * clip(p, -1, 1), lower, upper) -- three tensor ops with the four rows cached per device"""
    import torch
    P = params.clamp(-1, 1).to(torch.float32)
    key = str(P.device)
    if key not in _P2D_CACHE:
        _P2D_CACHE[key] = tuple(torch.tensor(t, dtype=torch.float32, device=P.device) for t in _p2d_tables())
    off, scl, lo, hi = _P2D_CACHE[key]
    if P.shape[1] < 48:
        P = torch.cat([P, torch.zeros(P.shape[0], 48 - P.shape[1], device=P.device)], dim=1)
    return torch.minimum(torch.maximum(off + scl * P, lo), hi)


def default_dynamic_row():
    return dynamic_dict_to_row(param2dynamic_dict(np.zeros(48)))
