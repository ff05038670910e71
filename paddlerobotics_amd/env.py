"""Batched drop-in for the Gym surface ETGRL drives.

Reference surface (QuadrupedalRobots/ETGRL):
    env = rlschool.make_env('Quadrupedal', task=, motor_control_mode=, render=, sensor_mode=,
                            normal=, dynamic_param=, reward_param=, ETG=, ETG_T=, reward_p=,
                            ETG_path=, random_param=, ETG_H=, vel_d=, step_y=,
                            enable_action_filter=)                     train.py:305-309
    obs, info = env.reset(ETG_w=w, ETG_b=b, x_noise=0)                 train.py:131,186,215
    obs, info = env.reset(hardset=False, dynamic_param=dict)           Dynamic_parallel_model.py:55
    obs, reward, done, info = env.step(action * act_bound, donef=bool) train.py:147,195,228
    env.observation_space.shape[0], env.action_space.shape[0]          train.py:311-312
Here every array gains a leading [num_envs] dimension and lives on the GPU as a torch
tensor; the work is done by hand-written gfx950 kernels behind the C-ABI of
include/etgsim.h.  There is no CPU path.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import a1_model as A
from . import _lib
from .etg import ETG_layer, Opt_with_points
from .terrain import TERRAIN_TASKS, make_task_heightfield

TASKS = ("ground", "heightfield") + TERRAIN_TASKS

# observation columns of the 49-float row by sensor (sorted-by-key order of EnvWrapper.py:60-109)
_OBS_COLS = {"dis": (0, 3), "contact": (3, 7), "imu_rpy": (7, 10), "imu_drpy": (10, 13), "motor_q": (13, 25),
             "motor_qd": (25, 37), "ETG": (37, 49)}
DEFAULT_SENSOR_MODE = {"dis": 1, "motor": 1, "imu": 1, "contact": 1, "ETG": 1}
# optional sensors of train.py:268-271: columns of the etg_extra_sensors() row (include/etgsim.h ETG_EXTRA_*), appended
# after the 49-float row in this order.  rlschool's definitions are absent from the reference tree: DESIGN.md section 6.
EXTRA_SENSORS = ("ETG_obs", "footpose", "dynamic_vec", "force_vec")
# sensor_mode['noise'] (train.py:272 `--sensor_noise`): the noise levels the reference itself applies to these columns
# when it perturbs observations (BCtrain.py:53-59 obs2noise): motor angle 1e-2 rad, motor velocity 0.5 rad/s, rpy 6e-2,
# rpy rate 1e-1; order of observation_noise_stdev (minitaur.py:102): angle, velocity, torque, rpy, rpy rate
SENSOR_NOISE_STDEV = (1e-2, 0.5, 0.0, 6e-2, 1e-1)


def sensor_columns(sensor_mode):
    """Columns of the full 49-float observation selected by a reference sensor_mode dict, following
    SimpleEnv.get_observation (deployment/envs/EnvWrapper.py:60-109) and the obs-dim rule of deployment/test.py:26-46:
    motor 1 -> angles + velocities, 2 -> angles; imu 1 -> rpy + rpy rate, 2 -> the rpy RATE alone
    (EnvWrapper.py:91-92 `sensors_dict["IMU"] = drpy`); dis / contact / ETG on-off."""
    sm = dict(DEFAULT_SENSOR_MODE)
    sm.update(sensor_mode or {})
    rnn = sm.get("RNN")
    if rnn and rnn.get("time_steps", 0) > 0 and rnn.get("mode", "stack") not in ("stack", "GRU"):
        raise NotImplementedError("sensor_mode['RNN']['mode'] must be 'stack' or 'GRU'")
    cols = []
    def add(name):
        a, b = _OBS_COLS[name]
        cols.extend(range(a, b))
    if sm.get("dis"):
        add("dis")
    if sm.get("contact"):
        add("contact")
    if sm.get("imu") == 1:
        add("imu_rpy")
        add("imu_drpy")
    elif sm.get("imu") == 2:
        add("imu_drpy")
    if sm.get("motor") in (1, 2):
        add("motor_q")
        if sm["motor"] == 1:
            add("motor_qd")
    if sm.get("ETG"):
        add("ETG")
    return cols


def extra_sensor_columns(sensor_mode):
    """columns of the etg_extra_sensors() row selected by sensor_mode's optional sensors (train.py:268-271)"""
    cols = []
    for k in EXTRA_SENSORS:
        if (sensor_mode or {}).get(k):
            a, b = A.EXTRA_SLICES[k]
            cols.extend(range(a, b))
    return cols


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _InfoView(dict):
    """info dict of a step: the keys callers read (train.py:150-156, env_test.py:54, Dynamic_parallel_model.py:63-64)
    as [N] / [N,k] views of the one device buffer the kernel filled.  The views are made on first access -- building
    all 14 of them eagerly costs more host time per step than the step kernel takes."""

    def __init__(self, buf):
        super().__init__()
        self._buf = buf

    def __missing__(self, key):
        a, b = A.INFO_SLICES[key]
        v = self._buf[:, a] if b - a == 1 else self._buf[:, a:b]
        self[key] = v
        return v

    def __contains__(self, key):
        return key in A.INFO_SLICES

    def get(self, key, default=None):
        return self[key] if key in A.INFO_SLICES else default

    def keys(self):
        return A.INFO_SLICES.keys()

    def items(self):
        return [(k, self[k]) for k in A.INFO_SLICES]

    def __iter__(self):
        return iter(A.INFO_SLICES)

    def __len__(self):
        return len(A.INFO_SLICES)


class FusedKernelUnavailable(ValueError):
    """the configuration is outside what a fused rollout kernel covers (callers may fall back to the stepping loop)"""


def _body_contacts_mode(v):
    """make_env's body_contacts keyword -> EtgConfig.body_contacts"""
    if v in (3, "simultaneous"):
        return 3
    if v in (2, "all"):
        return 2
    if isinstance(v, str):
        raise ValueError("body_contacts=%r: 0 / False, 1 / True, 2 / 'all' or 3 / 'simultaneous'" % (v,))
    if v is True or v == 1:
        return 1
    if v is False or v is None or v == 0:
        return 0
    raise ValueError("body_contacts=%r: 0 / False, 1 / True, 2 / 'all' or 3 / 'simultaneous'" % (v,))


def _uniform_torque_limit(limits):
    """motor_torque_limits of the robot class (minitaur.py:209-214): a scalar or one value per motor.  The kernels clip every
    motor with ONE limit, so a sequence is accepted when its entries agree and refused (not silently collapsed) otherwise."""
    if limits is None:
        return 0.0
    a = np.asarray(limits, dtype=np.float64)
    if a.ndim == 0:
        return float(a)
    a = a.reshape(-1)
    if a.size not in (1, A.NUM_MOTORS):
        raise ValueError("motor_torque_limits: a scalar or %d values (minitaur.py:209-214), got %d" % (A.NUM_MOTORS, a.size))
    if not np.all(a == a[0]):
        raise ValueError("motor_torque_limits: per-motor limits that differ are not supported (the motor model clips every "
                         "motor with one limit); pass a scalar or equal entries")
    return float(a[0])


class BatchedQuadrupedEnv:
    def __init__(self, num_envs=1, device="cuda:0", task="ground", motor_control_mode=None, render=False,
                 sensor_mode=None, normal=1, dynamic_param=None, reward_param=None, ETG=1, ETG_T=0.5,
                 reward_p=5.0, ETG_path="", random_param=None, ETG_H=20, vel_d=0.5, step_y=0.05,
                 enable_action_filter=False, ETG_T2=0.5, action_repeat=13, sim_time_step=0.002,
                 settle_ticks=500, solver_iters=None, solver_residual=None, friction_model=None, pd_latency=0.0,
                 enable_action_interpolation=False,
                 heightfield=None, lanes_per_robot=0, terrain_variants=16, terrain_seed=0,
                 random_dynamics_scale=0.3, random_force_prob=0.02, random_force_steps=8,
                 random_force_range=(5.0, 25.0), seed=0, enable_clip_motor_commands=False,
                 observation_noise_stdev=None, body_contacts=2, body_friction=0.5, knee_radius=0.02, joint_limits=True,
                 auto_reset=False, random_dynamics_refresh=256, warmstart=0.1, warmstart_friction=0.0, contact_slop=1e-5,
                 foot_restitution=0.0, motor_torque_limits=None, solver_preset=None, body_blend=1e-3, **unused):
        if render:
            raise ValueError("render is not supported by the batched GPU simulator")
        if int(ETG_H) != A.RBF_H:
            raise ValueError("ETG_H must be %d" % A.RBF_H)
        if task not in TASKS:
            raise ValueError("task %r is not available: this simulator has %s (for 'heightfield' pass "
                             "heightfield=dict(heights=[ny,nx], cell=, origin=(x0,y0)))" % (task, ", ".join(TASKS)))
        if task in TERRAIN_TASKS:
            if heightfield is not None:
                raise ValueError("task=%r builds its own heightfield" % task)
            heightfield = make_task_heightfield(task, variants=int(terrain_variants), seed=int(terrain_seed))
        elif (task == "heightfield") != (heightfield is not None):
            raise ValueError("task='heightfield' and the heightfield= argument go together")
        self.task = task
        self.terrain = heightfield
        # train.py:56-58 mode_map: "pose"/"traj" -> POSITION, "torque" -> TORQUE; enum values also accepted
        mname = getattr(motor_control_mode, "name", motor_control_mode)
        if mname in (None, "pose", "traj", "POSITION", 1):
            motor_mode = 0
        elif mname in ("torque", "TORQUE", 2):
            motor_mode = 1
        elif mname in ("hybrid", "HYBRID", 3):
            motor_mode = 2          # the action is the 60-vector of laikago_motor.py:152-161
        else:
            raise NotImplementedError("motor_control_mode %r: POSITION, TORQUE and HYBRID exist" % (motor_control_mode,))
        self.motor_mode = motor_mode
        self._cols = sensor_columns(sensor_mode)
        self._xcols = extra_sensor_columns(sensor_mode)
        if (sensor_mode or {}).get("noise") and observation_noise_stdev is None:
            observation_noise_stdev = SENSOR_NOISE_STDEV
        self.auto_reset = bool(auto_reset)
        # observation history (ObservationWrapper, deployment/envs/EnvWrapper.py:195-238): the current reading
        # preceded by `time_steps` older ones taken `time_interval` control steps apart, flattened ("stack", obs
        # dim x (time_steps + 1), deployment/test.py:44-45) or kept as a sequence ("GRU")
        rnn = (sensor_mode or {}).get("RNN") or {}
        self._hist_T = int(rnn.get("time_steps", 0))
        self._hist_dt = int(rnn.get("time_interval", 1)) if self._hist_T > 0 else 1
        self._hist_mode = rnn.get("mode", "stack")
        self._hist = None
        self._hist_head = 0
        self._last_seq = None
        rp = dict(random_param or {})
        self._rand_dyn = bool(rp.get("random_dynamics", 0))
        self._rand_force = bool(rp.get("random_force", 0))
        self._rand_dyn_scale = float(random_dynamics_scale)
        # random_dynamics under auto_reset: the rows of the robots' NEXT episodes are drawn and settled every this many control
        # steps in one launch (etg_prepare_next_dynamics) instead of one settle launch per control step; <= 1: a fresh draw at
        # every single reset, through the masked-reset path (exact, ~70x slower at 4096 robots)
        self._nx_refresh = int(random_dynamics_refresh)
        self._nx_on, self._nx_count, self._nx_pending = False, 0, None
        self._rf_prob, self._rf_steps, self._rf_range = float(random_force_prob), int(random_force_steps), random_force_range
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("BatchedQuadrupedEnv runs on a HIP device only (device='cuda:N')")
        self.ETG = int(ETG)
        if solver_preset is not None:      # named engine settings (a1_model.solver_preset); every explicit keyword wins on its own
            ps = A.solver_preset(solver_preset, action_repeat)
            if solver_iters is None and solver_residual is None:
                solver_iters, solver_residual = ps["solver_iters"], ps["solver_residual"]
            elif solver_iters is None:     # (a residual threshold alone keeps the preset's sweep cap; a bare solver_iters is a
                solver_iters = ps["solver_iters"]   # fixed count by the rule of a1_model.solver_rule and stays one)
            if friction_model is None:
                friction_model = ps["friction_model"]
        if friction_model is None:
            friction_model = 0
        self.cfg = A.default_config(
            self.num_envs, action_repeat=action_repeat, sim_dt=sim_time_step, settle_ticks=settle_ticks,
            solver_iters=solver_iters, solver_residual=solver_residual, friction_model=friction_model, pd_latency=pd_latency,
            enable_action_interp=enable_action_interpolation,
            enable_action_filter=enable_action_filter, normal=normal,
            terrain=1 if heightfield is not None else 0, ETG_T=ETG_T, ETG_T2=ETG_T2,
            reward_param=reward_param, reward_p=reward_p, vel_d=vel_d, heightfield=heightfield,
            lanes_per_robot=lanes_per_robot, motor_mode=motor_mode,
            clip_motor_commands=0.2 if enable_clip_motor_commands else 0.0,   # MAX_MOTOR_ANGLE_CHANGE_PER_STEP, a1.py
            # which link shapes collide besides the toe spheres (Bullet: every link's, a1.py:276-287).  2 or "all" (the default):
            # one contact per leg, with friction, on the deepest of knee / shin midpoint / trunk corner; 1 / True: the knee
            # sphere only; 0 / False: toe spheres only; 3 or "simultaneous": all three spheres at once, frictionless
            body_contacts=_body_contacts_mode(body_contacts), body_friction=float(body_friction),
            knee_radius=knee_radius, body_blend=float(body_blend),
            enable_etg=1 if self.ETG else 0, joint_limits=1 if joint_limits else 0,
            # contact-solver settings: the defaults are pybullet's (a1_model.default_config)
            warmstart=warmstart, warmstart_friction=warmstart_friction, contact_slop=contact_slop,
            foot_restitution=foot_restitution,
            # motor_torque_limits of the robot class (minitaur.py:99,127; one value for all motors): the clip of laikago_motor.py:168-173
            torque_limit=_uniform_torque_limit(motor_torque_limits))
        self.model = A.default_model()
        if task == "balancebeam":
            # README "step_y: the foot position at y axis for balance beam task" (train.py:463): the ETG's nominal
            # foot positions are pulled in to y = -+step_y so the feet land on the beam (right legs negative y)
            for leg in range(4):
                self.model.base_foot[3 * leg + 1] = (-1.0 if leg % 2 == 0 else 1.0) * float(step_y)
        d = len(self._cols) + len(self._xcols)
        if self._hist_T > 0:
            shape = (d * (self._hist_T + 1),) if self._hist_mode == "stack" else (self._hist_T + 1, d)
        else:
            shape = (d,)
        self.observation_space = Box(-np.inf, np.inf, shape)
        self.action_space = Box(-1.0, 1.0, (60 if motor_mode == 2 else A.NUM_MOTORS,))
        self._lib = _lib.load()
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._lib.etg_create(C.byref(self.cfg), C.byref(self.model), idx, C.byref(self._h)))
        self.lanes_per_robot = int(self._lib.etg_lanes_per_robot(self._h))   # 0 (auto) resolved by the library
        N, dev = self.num_envs, self.device
        self.obs = torch.zeros(N, A.OBS_DIM, device=dev)
        self.reward = torch.zeros(N, device=dev)
        self.done = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.info_buf = torch.zeros(N, A.INFO_DIM, device=dev)
        self._col_idx = None if len(self._cols) == A.OBS_DIM else torch.tensor(self._cols, device=dev)
        self._xcol_idx = torch.tensor(self._xcols, device=dev) if self._xcols else None
        self.extra = torch.zeros(N, A.EXTRA_DIM, device=dev) if self._xcols else None
        self._push_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._dyn_gen = torch.Generator(device=dev)
        self._dyn_gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        if observation_noise_stdev is not None:
            self.set_sensor_noise(observation_noise_stdev, seed=seed)
        self._dyn_stage = self._dyn_dev = self._dyn_evt = None
        self._all_done = None
        self._noise_offsets = False
        self._last_view = None
        self._hf = None
        if heightfield is not None:
            self._hf = torch.as_tensor(np.ascontiguousarray(heightfield["heights"], dtype=np.float32), device=dev)
            _lib.check(self._lib.etg_set_heightfield(self._h, _ptr(self._hf), self._stream()))
        # ETG prior (train.py:296-299) unless ETG_path provides w, b
        self._etg_layer = ETG_layer(ETG_T, self.cfg.etg_dt, A.RBF_H, 0.04, np.array([-np.pi / 2, 0]), 0.2, ETG_T2)
        if ETG_path:
            data = np.load(ETG_path)
            w0, b0 = data["w"], data["b"]
        elif self.ETG:
            w0, b0, _ = Opt_with_points(self._etg_layer, ETG_T=ETG_T, Footheight=0.1, Steplength=0.05)
        else:
            w0, b0 = np.zeros((3, A.RBF_H)), np.zeros(3)
        self.set_etg(w0, b0)
        if dynamic_param is not None:
            self.set_dynamic_param(dynamic_param)

    # ---- plumbing --------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, x, shape, what):
        t = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s" % (what, tuple(shape), tuple(t.shape)))
        return t

    def _mask(self, env_ids):
        """uint8 [N] mask of the robots a call applies to: `env_ids` is None (all robots), a list / array / tensor of
        robot indices, or a boolean (or uint8) [N] mask such as the `done` tensor step() returns."""
        if env_ids is None:
            return None
        t = torch.as_tensor(env_ids, device=self.device)
        if t.dtype in (torch.bool, torch.uint8):
            if tuple(t.shape) != (self.num_envs,):
                raise ValueError("a boolean env_ids mask must have shape [N]")
            return t.to(torch.uint8).contiguous()
        if t.dim() > 1 or t.is_floating_point():
            raise ValueError("env_ids must be robot indices or a boolean [N] mask")
        m = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        m[t.reshape(-1).long()] = 1
        return m

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            torch.cuda.synchronize(self.device)
            self._lib.etg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ------------------------------------------------------------
    def set_etg(self, ETG_w, ETG_b, env_ids=None):
        w = torch.as_tensor(np.asarray(ETG_w) if not torch.is_tensor(ETG_w) else ETG_w, dtype=torch.float32,
                            device=self.device).contiguous()
        b = torch.as_tensor(np.asarray(ETG_b) if not torch.is_tensor(ETG_b) else ETG_b, dtype=torch.float32,
                            device=self.device).contiguous()
        per_env = int(w.dim() == 3)
        if per_env:
            if tuple(w.shape) != (self.num_envs, 3, A.RBF_H) or tuple(b.shape) != (self.num_envs, 3):
                raise ValueError("per-env ETG_w/ETG_b must be [N,3,20] / [N,3]")
        elif tuple(w.shape) != (3, A.RBF_H) or tuple(b.shape) != (3,):
            raise ValueError("ETG_w/ETG_b must be [3,20] / [3]")
        m = self._mask(env_ids)
        _lib.check(self._lib.etg_set_params(self._h, None, _ptr(w), _ptr(b), per_env, _ptr(m), self._stream()))
        self._keep = (w, b, m)

    def set_dynamic_param(self, dynamic_param, env_ids=None):
        """dynamic_param: dict as produced by param2dynamic_dict (train.py:112-126), a [48] row,
        or an [N,48] tensor/array of rows (one per robot)."""
        if isinstance(dynamic_param, dict):
            dynamic_param = A.dynamic_dict_to_row(dynamic_param)
        if torch.is_tensor(dynamic_param) and dynamic_param.device.type == "cuda":
            t = dynamic_param.to(device=self.device, dtype=torch.float32)
            if t.dim() == 1:
                t = t.unsqueeze(0).expand(self.num_envs, -1)
            t = t.contiguous()
        else:
            # host rows go through ONE persistent pinned staging buffer: a fresh pageable->device copy makes the
            # runtime allocate pinned staging memory now and then (50-90 ms a time, measured)
            rows = np.asarray(dynamic_param.cpu() if torch.is_tensor(dynamic_param) else dynamic_param, dtype=np.float32)
            if rows.ndim == 1:
                rows = np.broadcast_to(rows, (self.num_envs, rows.shape[0]))
            if rows.shape != (self.num_envs, A.DYN_DIM):
                raise ValueError("dynamic_param rows must be [N,48]")
            if self._dyn_stage is None:
                self._dyn_stage = torch.empty(self.num_envs, A.DYN_DIM, dtype=torch.float32, pin_memory=True)
                self._dyn_dev = torch.empty(self.num_envs, A.DYN_DIM, dtype=torch.float32, device=self.device)
                self._dyn_evt = torch.cuda.Event()
            else:
                self._dyn_evt.synchronize()                   # the previous upload has left the staging buffer
            self._dyn_stage.numpy()[...] = rows
            self._dyn_dev.copy_(self._dyn_stage, non_blocking=True)
            self._dyn_evt.record(torch.cuda.current_stream(self.device))
            t = self._dyn_dev
        if tuple(t.shape) != (self.num_envs, A.DYN_DIM):
            raise ValueError("dynamic_param rows must be [N,48]")
        m = self._mask(env_ids)
        _lib.check(self._lib.etg_set_params(self._h, _ptr(t), None, None, 0, _ptr(m), self._stream()))
        self._keep_dyn = (t, m)

    # ---- Gym surface -----------------------------------------------------------
    def _info(self):
        return _InfoView(self.info_buf)

    def reset(self, env_ids=None, ETG_w=None, ETG_b=None, dynamic_param=None, x_noise=0, hardset=False, _keep_offsets=False, **kwargs):
        if ETG_w is not None:
            self.set_etg(ETG_w, ETG_b, env_ids)
        if dynamic_param is not None:
            self.set_dynamic_param(dynamic_param, env_ids)
            if env_ids is None:
                self._nx_on = False      # explicit parameters for everybody: no prepared random rows until the next plain full reset
        elif self._rand_dyn:
            # random_param['random_dynamics'] (train.py:253): a fresh draw of the 48 dynamic parameters for
            # every robot being reset, param2dynamic_dict(U(-1,1) * scale) (train.py:112-126)
            # drawn on the device (torch generator seeded from `seed`): no host RNG pass, no upload per reset
            p = (torch.rand(self.num_envs, A.DYN_DIM, device=self.device, generator=self._dyn_gen) * 2 - 1) * self._rand_dyn_scale
            self.set_dynamic_param(A.param2dynamic_rows_torch(p), env_ids)   # masked: only the robots being reset take their row
        m = self._mask(env_ids)
        if x_noise:
            # start-position jitter (train.py:131 `x_noise=args.x_noise`, an int flag there; rlschool's own
            # distribution is absent): x0 ~ U(-0.1, 0.1) * x_noise metres for every robot being reset
            # (drawn on the device from the env's seeded generator: no host RNG pass, no upload per reset)
            xy = torch.zeros(self.num_envs, 2, dtype=torch.float32, device=self.device)
            xy[:, 0] = (torch.rand(self.num_envs, device=self.device, generator=self._dyn_gen) * 0.2 - 0.1) * float(x_noise)
            self.set_reset_offsets(xy, env_ids)
            self._noise_offsets = True
        elif self._noise_offsets and not _keep_offsets:       # an earlier reset jittered these robots: back to the nominal start
            # (_keep_offsets: the restart of step(auto_reset) -- finished robots start over where their last reset put them, on the
            # fused path and on this one alike)
            self.set_reset_offsets(None, env_ids)
            self._noise_offsets = env_ids is not None
        if self._rand_force:
            _lib.check(self._lib.etg_clear_pushes(self._h, _ptr(m), self._stream()))
        _lib.check(self._lib.etg_reset(self._h, _ptr(m), _ptr(self.obs), self._stream()))
        if self._rand_dyn and self.auto_reset and self._nx_refresh > 1 and env_ids is None and dynamic_param is None:
            # the next episodes' rows are prepared ahead of time (etg_prepare_next_dynamics); the library leaves out robots in the
            # first 64 ticks of their episode (they still read their pre-reset history from the settle cache the call replaces),
            # so the FIRST call waits until the fresh episodes are that old; a robot that finishes before it restarts on its
            # current rows (same draw again), through the same fused launch
            self._nx_on = True
            self._nx_first = -(-64 // int(self.cfg.action_repeat))          # control steps until every robot is old enough
            self._nx_count = self._nx_refresh - self._nx_first              # (the first call after exactly _nx_first steps)
        info = {"ETG_act": None}
        self._last_view = self._obs_view(reset_mask=m, first=True)
        return self._last_view, info

    def _draw_dynamics_rows(self):
        """param2dynamic_dict(U(-1,1) * scale) rows for all robots (train.py:112-126, 253), drawn on the device"""
        p = (torch.rand(self.num_envs, A.DYN_DIM, device=self.device, generator=self._dyn_gen) * 2 - 1) * self._rand_dyn_scale
        return A.param2dynamic_rows_torch(p).to(torch.float32).contiguous()

    def _prepare_next_dynamics(self, mask):
        """draw + settle the dynamics of the NEXT episodes of the masked robots (uint8 [N] device tensor, None = all);
        False when the library cannot (not every robot has a cached settle, e.g. a heightfield with start jitter)"""
        rows = self._draw_dynamics_rows()
        rc = self._lib.etg_prepare_next_dynamics(self._h, _ptr(rows), _ptr(mask), self._stream())
        self._nx_rows = rows      # keep the buffer alive until the stream has consumed it
        if rc not in (0, _lib.ETG_ERR_STATE):
            _lib.check(rc)        # allocation / launch errors are errors, not "cannot prepare"
        return rc == 0

    def _refresh_next_dynamics(self):
        """every random_dynamics_refresh control steps: the robots whose prepared rows were consumed by a restart get new ones"""
        self._nx_count += 1
        if self._nx_count < self._nx_refresh:
            return
        self._nx_count = 0
        if self._nx_pending is None:
            self._nx_pending = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.etg_next_dynamics_pending(self._h, _ptr(self._nx_pending), self._stream()))
        consumed = (self._nx_pending == 0).to(torch.uint8)
        self._nx_mask = consumed
        if not self._prepare_next_dynamics(consumed):
            self._nx_on = False
            import warnings
            warnings.warn("random_dynamics under auto_reset: the library cannot prepare the next episodes' dynamics ahead of time "
                          "(%s); falling back to a masked reset per control step, which is much slower"
                          % self._lib.etg_last_error().decode(), RuntimeWarning)

    def _obs_view(self, reset_mask=None, first=False):
        """the observation the caller sees: sensor_mode column selection, then (optionally) the history stack.
        reset_mask: uint8 [N] of the robots that were just reset (None = all, with first=True)."""
        o = self.obs if self._col_idx is None else self.obs.index_select(1, self._col_idx)
        if self._xcol_idx is not None:   # optional sensors (train.py:268-271), appended after the 49-float row's columns
            _lib.check(self._lib.etg_extra_sensors(self._h, _ptr(self.obs), _ptr(self.extra), self._stream()))
            o = torch.cat([o, self.extra.index_select(1, self._xcol_idx)], dim=1)
        if self._hist_T == 0:
            return o
        H = self._hist_T * self._hist_dt
        if self._hist is None:
            self._hist = torch.zeros(H, self.num_envs, o.shape[1], device=self.device)
        if first:   # reset: zero history for the reset robots; the list is read BEFORE the new reading is stored
            if reset_mask is None:
                self._hist.zero_()
            else:
                self._hist.mul_((1 - reset_mask.float()).view(1, -1, 1))
        older = [self._hist[(self._hist_head + t * self._hist_dt) % H] for t in range(self._hist_T)]
        seq = torch.stack(older + [o], dim=1)                      # [N, T+1, D]; copies, taken BEFORE the ring is updated
        if first:
            last = (self._hist_head + H - 1) % H
            if reset_mask is None:
                self._hist[last] = o
            else:   # robots that were not reset keep their own history ...
                rm = reset_mask.bool()
                self._hist[last] = torch.where(rm.unsqueeze(1), o, self._hist[last])
                # ... and their own view: the ring already holds their newest reading, so `older` above is one reading too
                # new for them -- they see what the last step showed them (a masked reset does not advance their time)
                if self._last_seq is not None:
                    seq = torch.where(rm.view(-1, 1, 1), seq, self._last_seq)
        else:
            self._hist_head = (self._hist_head + 1) % H          # logical shift by one ...
            self._hist[(self._hist_head + H - 1) % H] = o          # ... and the newest reading goes last
        self._last_seq = seq
        return seq.reshape(self.num_envs, -1) if self._hist_mode == "stack" else seq

    def set_sensor_noise(self, stdev, seed=0):
        """Gaussian sensor noise (minitaur.py:102,136 `observation_noise_stdev`): 5 standard deviations -- motor
        angle, motor velocity, motor torque, base rpy, base rpy rate -- or None to switch it off.  Applies to the
        observation rows (through step() and the fused rollouts alike); the dynamics never see it."""
        if stdev is None:
            _lib.check(self._lib.etg_set_sensor_noise(self._h, None, C.c_uint64(0)))
            self._noise_on = False
            return
        a = np.ascontiguousarray(stdev, dtype=np.float32)
        if a.shape != (5,):
            raise ValueError("observation_noise_stdev needs 5 values (angle, velocity, torque, rpy, rpy rate)")
        self._noise_on = bool(np.any(a != 0))
        _lib.check(self._lib.etg_set_sensor_noise(self._h, a.ctypes.data_as(C.c_void_p), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)))

    def set_reset_offsets(self, xy, env_ids=None):
        """start offsets xy [N,2] (m, added to the initial x / y) of the following resets of `env_ids`
        (None = all robots); xy None = zero. `reset(x_noise=...)` draws them itself."""
        m = self._mask(env_ids)
        if xy is None:
            _lib.check(self._lib.etg_set_reset_offsets(self._h, None, _ptr(m), self._stream()))
            return
        self._offsets = self._f32(xy, (self.num_envs, 2), "reset offsets").clone()   # kept alive until the copy ran
        _lib.check(self._lib.etg_set_reset_offsets(self._h, _ptr(self._offsets), _ptr(m), self._stream()))

    def set_external_force(self, force):
        """force [N,3] (world frame, N) pushed on every trunk until replaced; None clears it."""
        if force is None:
            _lib.check(self._lib.etg_set_external_force(self._h, None, self._stream()))
            return
        self._force = self._f32(force, (self.num_envs, 3), "force").clone()   # kept alive until the copy ran
        _lib.check(self._lib.etg_set_external_force(self._h, _ptr(self._force), self._stream()))

    def set_motor_strength_ratios(self, ratios, env_ids=None):
        """Minitaur.SetMotorStrengthRatios / SetMotorStrengthRatio (minitaur.py:1280-1294): per-motor factors on the motor
        model's output torque (laikago_motor.py:67-76,138,167).  ratios: a scalar, [12] or [N,12]; None = 1 for everybody.
        The touched robots settle again at their next reset (the settle runs under the motor model)."""
        m = self._mask(env_ids)
        if ratios is None:
            _lib.check(self._lib.etg_set_motor_strength(self._h, None, _ptr(m), self._stream()))
            return
        r = torch.as_tensor(ratios, dtype=torch.float32, device=self.device)
        if r.dim() == 0:
            r = r.expand(A.NUM_MOTORS)
        if r.dim() == 1:
            r = r.unsqueeze(0).expand(self.num_envs, -1)
        if tuple(r.shape) != (self.num_envs, A.NUM_MOTORS):
            raise ValueError("motor strength ratios must be a scalar, [12] or [N,12]")
        self._strength = r.contiguous().clone()             # kept alive until the copy ran
        _lib.check(self._lib.etg_set_motor_strength(self._h, _ptr(self._strength), _ptr(m), self._stream()))

    SetMotorStrengthRatios = set_motor_strength_ratios      # the robot class's own name (minitaur.py:1288)

    def _random_pushes(self):
        """random_param['random_force'] (train.py:254): one tiny kernel per control step samples / expires the
        pushes on the device (include/etgsim.h: etg_random_pushes)."""
        _lib.check(self._lib.etg_random_pushes(self._h, C.c_uint64(self._push_seed), C.c_float(self._rf_prob),
                                               int(self._rf_steps), C.c_float(self._rf_range[0]),
                                               C.c_float(self._rf_range[1]), self._stream()))

    def leg_kinematics(self, q, want_jacobian=True):
        """foot_positions_in_base_frame [n,4,3] and analytical_leg_jacobian [n,4,3,3] (a1.py:113-173) of joint-angle
        rows q [n,12], computed on the device by the routine the physics tick's contact rows use."""
        t = torch.as_tensor(q, dtype=torch.float32, device=self.device).contiguous()
        if t.dim() != 2 or t.shape[1] != A.NUM_MOTORS:
            raise ValueError("q must be [n,12]")
        n = t.shape[0]
        foot = torch.empty(n, 4, 3, device=self.device)
        jac = torch.empty(n, 4, 3, 3, device=self.device) if want_jacobian else None
        _lib.check(self._lib.etg_leg_kinematics(self._h, _ptr(t), n, _ptr(foot), _ptr(jac), self._stream()))
        return (foot, jac) if want_jacobian else foot

    # ---- sub-batches (etg_step_range): the Gym loop with one barrier per group instead of one per batch
    def group_ranges(self, groups):
        """[(first robot, one past the last)] of `groups` sub-batches: equal sizes rounded up to whole wavefronts of the 4-lane
        mapping (16 robots), the last one takes what is left"""
        G = int(groups)
        if G < 1:
            raise ValueError("groups must be >= 1")
        per = -(-self.num_envs // G)
        per = -(-per // 16) * 16
        return [(a, min(a + per, self.num_envs)) for a in range(0, self.num_envs, per)]

    def _groups_ok(self, what):
        if self.auto_reset or self._rand_force or self._hist_T > 0 or len(self._xcols) > 0:
            raise ValueError("%s: sub-batches are for envs without auto_reset, random pushes, observation history and extra sensors" % what)

    def _group_streams(self, n, fresh=False):
        """the n side streams of a split into n sub-batches (kept per n: tune_groups() may have picked them)"""
        sets = self.__dict__.setdefault("_gstreams", {})
        if fresh or n not in sets:
            sets[n] = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        return sets[n]

    def tune_groups(self, candidates=(2, 4), steps=15, tries=3):
        """Pick the number of sub-batches for run_groups() / rollout_policy(fused=False, groups=G) by measurement; returns
        (best G, {G: microseconds per control step}) with G = 1 always among the candidates.  Steps the robots (zero residual
        action, restarts from the settle cache in between): call it before the reset that starts the real work.

        Why measure: the gain comes from the sub-batches' kernels running CONCURRENTLY, which needs their streams on different
        hardware queues.  The HIP runtime multiplexes streams onto 4 hardware queues (GPU_MAX_HW_QUEUES); two sub-batches that
        land on one queue run one after the other, and a step kernel's duration is its slowest wavefront's whatever its size
        -- so a collision costs ~2x instead of gaining ~10 % (profiles/r06_groups.txt).  A candidate that measures slower than
        1.15 x the one-batch loop is retried on fresh streams (`tries` times) before it is given up."""
        self._groups_ok("tune_groups")

        def measure(G):
            self.reset()
            self.run_groups(steps, G)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            self.run_groups(steps, G)
            torch.cuda.synchronize(self.device)
            return (time.perf_counter() - t0) / steps * 1e6
        table = {1: min(measure(1) for _ in range(2))}
        for G in candidates:
            G = len(self.group_ranges(G))
            if G <= 1 or G in table:
                continue
            best_t, best_streams = None, None
            for k in range(max(1, int(tries))):
                streams = self._group_streams(G, fresh=k > 0)
                t = min(measure(G) for _ in range(2))
                if best_t is None or t < best_t:
                    best_t, best_streams = t, streams
                if t <= 1.15 * table[1]:
                    break
            self._gstreams[G] = best_streams
            table[G] = best_t
        best = min(table, key=table.get)
        return best, table

    def step_range(self, first, last, action, donef=None, want_info=False):
        """One control step of robots [first, last) only, enqueued on the CURRENT stream (etg_step_range).  `action` / `donef`
        are the whole batch's [N, ...] arrays (rows outside the range are not read); returns views of the range's rows of
        (obs, reward, done).  Ranges come from group_ranges(); those of one control step are called in ascending order."""
        self._groups_ok("step_range")
        a = None if action is None else self._f32(action, (self.num_envs, self.action_space.shape[0]), "action")
        df = None if donef is None else torch.as_tensor(donef, device=self.device).to(torch.uint8).contiguous()
        _lib.check(self._lib.etg_step_range(self._h, int(first), int(last - first), _ptr(a), _ptr(df), _ptr(self.obs), _ptr(self.reward),
                                            _ptr(self.done), _ptr(self.info_buf) if want_info else None, self._stream()))
        self._keep_step = (a, df)
        return self.obs[first:last], self.reward[first:last], self.done[first:last]

    def _step_grouped(self, a, df, want_info, groups):
        """step() as `groups` launches on as many side streams, joined on the current stream: the same result, robot by robot"""
        self._groups_ok("step(groups=G)")
        cur = torch.cuda.current_stream(self.device)
        streams = self._group_streams(len(self.group_ranges(groups)))
        for (lo, hi), st in zip(self.group_ranges(groups), streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                _lib.check(self._lib.etg_step_range(self._h, lo, hi - lo, _ptr(a), _ptr(df), _ptr(self.obs), _ptr(self.reward), _ptr(self.done),
                                                    _ptr(self.info_buf) if want_info else None, self._stream()))
        for st in streams:
            cur.wait_stream(st)

    def step(self, action, donef=None, want_info=True, groups=1):
        """One control step of every robot (env.step of the reference, batched).  groups=G > 1: the batch is stepped as G
        sub-batches on G streams (identical results); see rollout_policy(fused=False, groups=G) for the loop that gains from it."""
        a = None if action is None else self._f32(action, (self.num_envs, self.action_space.shape[0]), "action")   # NULL = zero residual
        df = None
        if donef is not None:
            if isinstance(donef, (bool, int, np.bool_)):
                if bool(donef):                               # False is the same as no donef at all
                    if self._all_done is None:
                        self._all_done = torch.ones(self.num_envs, dtype=torch.uint8, device=self.device)
                    df = self._all_done
            else:
                df = torch.as_tensor(donef, device=self.device).to(torch.uint8).contiguous()
                if tuple(df.shape) != (self.num_envs,):
                    raise ValueError("donef must be a bool or have shape [N]")
        if self._rand_force:
            self._random_pushes()
        # auto_reset: robots whose episode just ended start the next one inside the same call (settle cache -> state, control
        # state, first observation): their rows of `obs` are the reset observation, reward / done / info are the finished
        # step's, info["reset"] (= done) marks them.  The done bytes are read on the device: no host synchronisation.
        # With random_dynamics the reset robots first draw new parameters, which needs the masked calls of reset().
        fused_reset = self.auto_reset and (not self._rand_dyn or self._nx_on)   # (_nx_on: the next episodes' dynamics are prepared)
        step_fn = self._lib.etg_step_autoreset if fused_reset else self._lib.etg_step
        if int(groups) > 1:
            self._step_grouped(a, df, want_info, groups)
        else:
            _lib.check(step_fn(self._h, _ptr(a), _ptr(df), _ptr(self.obs), _ptr(self.reward), _ptr(self.done),
                               _ptr(self.info_buf) if want_info else None, self._stream()))
        self._keep_step = (a, df)
        info = self._info() if want_info else {}
        if self.auto_reset:
            if not fused_reset:
                if self._hist_T > 0:
                    self._obs_view()                                         # the terminal reading enters the history first
                self._reset_mask = self.done.clone()
                self.reset(env_ids=self._reset_mask, _keep_offsets=True)
            if want_info:
                info["reset"] = self.done.view(torch.bool)
            if self._hist_T > 0 and fused_reset:
                self._obs_view()                                             # the terminal reading enters the history ...
                self._last_view = self._obs_view(reset_mask=self.done, first=True)   # ... and is cleared for the reset robots
            elif fused_reset:
                self._last_view = self._obs_view()
            if self._nx_on:
                self._refresh_next_dynamics()
        else:
            self._last_view = self._obs_view()
        return (self._last_view, self.reward, self.done.view(torch.bool) if want_info else self.done, info)

    def set_rollout_mode(self, simulate_finished=False):
        """The fused rollouts (rollout_openloop / _policy / _policy_record / _actions) stop simulating a robot when its episode
        ends, as the reference's loops do (pretrain.py:137-153, train.py:226-247): its state stays the terminal state, its
        observation row the last one.  simulate_finished=True brings back the behaviour of rounds 1-5 (finished robots are
        simulated on, their accumulators masked) -- for measurements and for looking at a robot after its fall."""
        _lib.check(self._lib.etg_set_rollout_mode(self._h, int(bool(simulate_finished))))
        self.simulate_finished = bool(simulate_finished)

    def rollout_wave_cycles(self):
        """shader-clock cycles of every wavefront in each launch of the last rollout_openloop: int64 [launches, waves]"""
        nl, nw = C.c_int32(0), C.c_int32(0)
        _lib.check(self._lib.etg_rollout_wave_cycles(self._h, None, 0, C.byref(nl), C.byref(nw), self._stream()))
        out = torch.zeros(max(nl.value, 0), max(nw.value, 1), dtype=torch.int64, device=self.device)
        if nl.value > 0:
            _lib.check(self._lib.etg_rollout_wave_cycles(self._h, _ptr(out), out.numel(), C.byref(nl), C.byref(nw), self._stream()))
        return out

    def rollout_openloop(self, n_steps, out=None):
        """n_steps control steps with zero residual action (pretrain.py:129-154) enqueued back to
        back; returns (episode_return[N], episode_len[N]) since the last reset, alive-masked.  out = (ret float32 [N],
        len int32 [N]) reuses the caller's buffers."""
        if self.auto_reset:
            raise ValueError("rollout_openloop does not restart finished robots: step an auto_reset env with step(), or roll out an "
                             "env made without auto_reset")
        if out is not None:
            ret, ln = out
            if ret.dtype != torch.float32 or ln.dtype != torch.int32 or ret.numel() != self.num_envs or ln.numel() != self.num_envs \
                    or ret.device.type != self.device.type or ln.device.type != self.device.type or not (ret.is_contiguous() and ln.is_contiguous()):
                raise ValueError("out must be (float32 [N], int32 [N]) contiguous tensors on the env's device")
        else:
            ret = torch.empty(self.num_envs, device=self.device)           # every entry is written by the kernel: no fill launch
            ln = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.etg_rollout_openloop(self._h, int(n_steps), _ptr(self.obs), _ptr(ret), _ptr(ln),
                                                  self._stream()))
        return ret, ln

    def rollout_policy(self, policy, n_steps, act_scale=0.3, precision=0, fused=None, groups=1):
        """n_steps closed-loop control steps with a fixed actor (policy.predict semantics); returns (episode_return[N],
        episode_len[N]).  The batched run_EStrain_episode / run_evaluate_episodes (train.py:182-249).  fused: True = the
        fused kernel (actor MLP + control step, up to 400 steps per launch; FusedKernelUnavailable when the configuration is outside
        it), False = policy.predict() + step() per control step, None = whichever is faster for this env: the fused kernel
        on the 16-lane mapping (7-13 % ahead), and on the 4-lane mapping only without body rows (there it is level with
        stepping; with body rows its 512-register budget spills and it is 25 % behind -- tools/closed_loop_probe.py)."""
        contiguous = self._cols == list(range(self._cols[0], self._cols[0] + len(self._cols)))   # e.g. the student's 3..48
        # (the fused kernels never restart a finished robot: an auto_reset env takes the stepping loop, which does)
        ok = (self.num_envs % (16 if self.lanes_per_robot == 16 else 64) == 0 and self.motor_mode != 2 and contiguous
              and self._hist_T == 0 and not self._rand_force and policy.obs_dim == len(self._cols)
              and policy.action_dim == A.NUM_MOTORS and not self.auto_reset
              and self.cfg.body_contacts != 3)   # (three body rows per leg: no closed-loop instantiation)
        if fused and not ok:
            raise FusedKernelUnavailable("rollout_policy(fused=True): this configuration is outside the fused closed-loop kernel")
        if fused is None:
            fused = ok and (self.lanes_per_robot == 16 or self.cfg.body_contacts == 0)
        if not fused and int(groups) > 1:
            return self._rollout_policy_grouped(policy, n_steps, act_scale, precision, groups)
        if not fused:
            act = None
            for _ in range(int(n_steps)):
                act = policy.predict(self._last_view.contiguous().view(self.num_envs, -1), act_scale, precision, out=act)
                self.step(act, want_info=False)
            return self.episode_stats()
        ret = torch.empty(self.num_envs, device=self.device)               # every entry is written by the kernel: no fill launch
        ln = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.etg_rollout_policy(self._h, policy._h, int(n_steps), C.c_float(act_scale), int(precision),
                                                int(self._cols[0]), _ptr(self.obs), _ptr(ret), _ptr(ln), self._stream()))
        self._last_view = self._obs_view()
        return ret, ln

    def run_groups(self, n_steps, groups, action_fn=None):
        """The Gym loop (train.py:129-178) per sub-batch: n_steps control steps of every robot, the batch split into `groups`
        sub-batches (group_ranges) whose loops run on their own streams -- sub-batch g's step k + 1 is queued behind ITS step k
        only, so it starts when its own slowest wavefront has finished, not the batch's.  action_fn(g, first, last, act_rows)
        is called on group g's stream before each of its steps and fills act_rows (the view [first:last] of the batch's action
        array) from self.obs[first:last]; None = zero residual action (the open loop of pretrain.py:129-154).  Per-robot results
        are those of groups = 1, bit for bit.  Returns (episode_return[N], episode_len[N])."""
        self._groups_ok("run_groups")
        ranges = self.group_ranges(groups)
        cur = torch.cuda.current_stream(self.device)
        streams = self._group_streams(len(ranges))
        act = None if action_fn is None else torch.empty(self.num_envs, self.action_space.shape[0], device=self.device)
        for st in streams:
            st.wait_stream(cur)
        for _ in range(int(n_steps)):
            for g, ((lo, hi), st) in enumerate(zip(ranges, streams)):
                with torch.cuda.stream(st):
                    if action_fn is not None:
                        action_fn(g, lo, hi, act[lo:hi])
                    _lib.check(self._lib.etg_step_range(self._h, lo, hi - lo, _ptr(act), None, _ptr(self.obs), _ptr(self.reward),
                                                        _ptr(self.done), None, self._stream()))
        for st in streams:
            cur.wait_stream(st)
        self._keep_step = (act, None)
        self._last_view = self._obs_view()
        return self.episode_stats()

    def _rollout_policy_grouped(self, policy, n_steps, act_scale, precision, groups):
        """predict() + step() per control step as in rollout_policy(fused=False), per sub-batch on its own stream (run_groups)"""
        if policy.obs_dim != len(self._cols):
            raise ValueError("the policy reads %d observation columns, the env shows %d" % (policy.obs_dim, len(self._cols)))
        full = len(self._cols) == A.OBS_DIM

        def act_rows(g, lo, hi, out):
            rows = self.obs[lo:hi] if full else self.obs[lo:hi].index_select(1, self._col_idx)
            policy.predict(rows, act_scale, precision, out=out)
        return self.run_groups(n_steps, groups, act_rows)

    def rollout_policy_record(self, policy, n_steps, act_scale=0.3, precision=0, noise=None):
        """rollout_policy that also records the episode (etg_rollout_policy_record): returns (ret [N], len [N], rec) with
        rec = dict(obs [T,N,49] the rows the actor acted on, action [T,N,12] unscaled, reward [T,N], done [T,N] bool,
        final_obs [N,49]).  The data half of run_EStrain_episode with es_rpm (train.py:213-249) at the fused kernel's speed;
        replay.store_recorded() moves the rows of live robots into a DeviceReplayMemory.  Needs the configuration the
        fused kernel needs (16-lane mapping, num_envs % 16 == 0, the full 49-float observation, POSITION / TORQUE mode)."""
        if self.auto_reset:
            raise FusedKernelUnavailable("rollout_policy_record does not restart finished robots: use an env without auto_reset")
        if not (self.lanes_per_robot == 16 and self.num_envs % 16 == 0 and self.motor_mode != 2 and self._hist_T == 0
                and not self._rand_force and self._cols == list(range(A.OBS_DIM)) and self._xcol_idx is None
                and policy.obs_dim == A.OBS_DIM and policy.action_dim == A.NUM_MOTORS):
            raise FusedKernelUnavailable("rollout_policy_record needs the 16-lane mapping, num_envs % 16 == 0, the plain 49-float "
                                         "observation and a 49 -> 12 actor; use replay.collect_transitions() for the other configurations")
        T, N = int(n_steps), self.num_envs
        if noise is not None:   # the stochastic actor (SAC.sample): tanh(mean + exp(clamp(log_std)) * noise), noise [T, N, 12] ~ N(0, 1)
            noise = torch.as_tensor(noise, dtype=torch.float32, device=self.device).contiguous()
            if tuple(noise.shape) != (T, N, A.NUM_MOTORS):
                raise ValueError("noise must be [n_steps, num_envs, 12]")
        # (zeros: the steps after a robot's episode has ended write reward 0 / done 1 only -- include/etgsim.h "fused rollouts")
        rec = {"obs": torch.zeros(T, N, A.OBS_DIM, device=self.device), "action": torch.zeros(T, N, A.NUM_MOTORS, device=self.device),
               "reward": torch.empty(T, N, device=self.device), "done": torch.empty(T, N, dtype=torch.uint8, device=self.device)}
        ret = torch.empty(N, device=self.device)
        ln = torch.empty(N, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.etg_rollout_policy_record(self._h, policy._h, T, C.c_float(act_scale), int(precision), 0, _ptr(self.obs),
                                                       _ptr(noise), _ptr(rec["obs"]), _ptr(rec["action"]), _ptr(rec["reward"]), _ptr(rec["done"]),
                                                       _ptr(ret), _ptr(ln), self._stream()))
        self._last_view = self._obs_view()
        rec["done"] = rec["done"].view(torch.bool)
        rec["final_obs"] = self.obs.clone()
        return ret, ln, rec

    def rollout_actions(self, actions, record=("joint_angle", "obs-IMU")):
        """len(actions) control steps over a KNOWN action tape in one launch per 50 steps (etg_rollout_actions): actions
        [T, N, 12] (already scaled, what step() takes) or [T, 12] (the same command for every robot).  record: the per-step
        outputs to keep -- any of "joint_angle" [T,N,12], "obs-IMU" [T,N,6] (the info columns the dynamics-identification
        replay reads, Dynamic_parallel_model.py:63-64), "obs" [T,N,49], "reward" [T,N], "done" [T,N].
        Returns (episode_return[N], episode_len[N], rec dict).  The same source as T calls of step() in another kernel: equal to
        rounding noise (the compiler contracts multiply-adds differently), not bit for bit."""
        N = self.num_envs
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device)
        if a.dim() == 2:
            a = a.unsqueeze(1).expand(a.shape[0], N, a.shape[1])
        a = a.contiguous()
        if a.dim() != 3 or tuple(a.shape[1:]) != (N, A.NUM_MOTORS):
            raise ValueError("actions must be [T, num_envs, 12] or [T, 12]")
        # configurations the fused kernel does not cover: callers that can step instead catch FusedKernelUnavailable
        if self.auto_reset:
            raise FusedKernelUnavailable("rollout_actions does not restart finished robots: use an env without auto_reset")
        if self.motor_mode == 2:
            raise FusedKernelUnavailable("rollout_actions: POSITION / TORQUE commands only (HYBRID rows have 60 columns)")
        if "obs" in record and getattr(self, "_noise_on", False):
            raise FusedKernelUnavailable("rollout_actions records observations without sensor noise: switch it off or step")
        T = a.shape[0]
        unknown = set(record) - {"joint_angle", "obs-IMU", "obs", "reward", "done"}
        if unknown:
            raise ValueError("rollout_actions cannot record %s" % sorted(unknown))
        rec = {}
        # (zeros: rows of the steps after a robot's episode has ended are not written -- include/etgsim.h "fused rollouts")
        if "joint_angle" in record:
            rec["joint_angle"] = torch.zeros(T, N, A.NUM_MOTORS, device=self.device)
        if "obs-IMU" in record:
            rec["obs-IMU"] = torch.zeros(T, N, 6, device=self.device)
        if "obs" in record:
            rec["obs"] = torch.zeros(T, N, A.OBS_DIM, device=self.device)
        if "reward" in record:
            rec["reward"] = torch.empty(T, N, device=self.device)
        if "done" in record:
            rec["done"] = torch.empty(T, N, dtype=torch.uint8, device=self.device)
        ret = torch.empty(N, device=self.device)
        ln = torch.empty(N, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.etg_rollout_actions(self._h, _ptr(a), int(T), _ptr(self.obs), _ptr(rec.get("joint_angle")), _ptr(rec.get("obs-IMU")),
                                                 _ptr(rec.get("obs")), _ptr(rec.get("reward")), _ptr(rec.get("done")), _ptr(ret), _ptr(ln),
                                                 self._stream()))
        self._keep_tape = a
        self._last_view = self._obs_view()
        if "done" in rec:
            rec["done"] = rec["done"].view(torch.bool)
        return ret, ln, rec

    def episode_stats(self):
        """(return[N], length[N]) accumulated on device since each robot's last reset, frozen at its
        first `done` (what run_episode / run_EStrain_episode return per candidate)."""
        ret = torch.empty(self.num_envs, device=self.device)               # every entry is written by the kernel: no fill launch
        ln = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.etg_episode_stats(self._h, _ptr(ret), _ptr(ln), self._stream()))
        return ret, ln

    # ---- state access (parity tests) -------------------------------------------
    def get_state(self):
        st = torch.zeros(self.num_envs, A.STATE_DIM, device=self.device)
        _lib.check(self._lib.etg_get_state(self._h, _ptr(st), self._stream()))
        return st

    def set_state(self, state):
        st = self._f32(state, (self.num_envs, A.STATE_DIM), "state")
        _lib.check(self._lib.etg_set_state(self._h, _ptr(st), self._stream()))
        self._keep_state = st

    def get_contact_impulses(self):
        """[N,16] the contact impulses of the last tick the solver warm-starts from: per leg (foot n, t1, t2, body contact's normal)"""
        lam = torch.zeros(self.num_envs, 16, device=self.device)
        _lib.check(self._lib.etg_get_contact_impulses(self._h, _ptr(lam), self._stream()))
        return lam

    def set_contact_impulses(self, lam):
        """install them (after set_state, which zeroes them): a re-created step then starts from the robot's own warm start"""
        lam = self._f32(lam, (self.num_envs, 16), "contact impulses")
        _lib.check(self._lib.etg_set_contact_impulses(self._h, _ptr(lam), self._stream()))
        self._keep_lam = lam


class SingleRobotEnv:
    """The reference's own scalar surface over a batch of ONE robot, for the scripts that drive a single env object and
    expect numpy / Python values (env_test.py:43-58, run_evaluate_episodes train.py:182-211, deployment/test.py):

        obs, info = env.reset(ETG_w=w, ETG_b=b)            obs: np.ndarray [obs_dim] float32
        obs, reward, done, info = env.step(action, donef)   reward: float, done: bool, info[key]: float / np.ndarray

    Every call copies its results to the host (one synchronisation per step): this is the compatibility path, the
    throughput path is the batched env.  The batched env is reachable as `.batched`."""

    def __init__(self, **kwargs):
        if kwargs.get("num_envs", 1) != 1:
            raise ValueError("SingleRobotEnv drives exactly one robot (num_envs = 1)")
        kwargs["num_envs"] = 1
        self.batched = BatchedQuadrupedEnv(**kwargs)
        self.observation_space, self.action_space = self.batched.observation_space, self.batched.action_space

    def _info(self, info):
        out = {}
        for k in info.keys():
            v = info[k]
            if v is None:
                out[k] = None
                continue
            v = v[0].detach().cpu().numpy()
            out[k] = float(v) if v.ndim == 0 else v.astype(np.float64)
        return out

    def reset(self, **kwargs):
        for k in ("ETG_w", "ETG_b"):
            if kwargs.get(k) is not None:
                kwargs[k] = np.asarray(kwargs[k], dtype=np.float32)
        obs, info = self.batched.reset(**kwargs)
        return obs[0].detach().cpu().numpy(), dict(info)

    def step(self, action, donef=False):
        b = self.batched
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, -1), device=b.device)
        obs, reward, done, info = b.step(a, donef=bool(donef))
        return obs[0].detach().cpu().numpy(), float(reward[0].item()), bool(done[0].item()), self._info(info)

    def close(self):
        self.batched.close()


def make_env(name="Quadrupedal", single=False, **kwargs):
    """rlschool.make_env('Quadrupedal', ...) stand-in (train.py:305-309) plus num_envs/device.  single=True: one robot
    behind the reference's numpy / scalar surface (SingleRobotEnv) for unmodified single-env scripts."""
    if name != "Quadrupedal":
        raise ValueError("only the 'Quadrupedal' environment exists here")
    return SingleRobotEnv(**kwargs) if single else BatchedQuadrupedEnv(**kwargs)
