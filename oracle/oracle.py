"""ctypes wrapper of oracle/libetgsim_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of etgsim_oracle.cpp).  The product package
paddlerobotics_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from paddlerobotics_amd import a1_model as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False, target="libetgsim_oracle.so"):
    so = os.path.join(_HERE, target)
    srcs = [os.path.join(_HERE, "etgsim_oracle.cpp"), os.path.join(_HERE, "..", "include", "etgsim.h")]
    if target == "libetgsim_cpu.so":
        srcs.append(os.path.join(_HERE, "etgsim_cpu_abi.cpp"))
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", target], stdout=subprocess.DEVNULL)
    return so


def build_cpu_abi(force=False):
    """oracle/libetgsim_cpu.so: the C-ABI of include/etgsim.h on host pointers, device = -1 (etgsim_cpu_abi.cpp)"""
    return build(force, "libetgsim_cpu.so")


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for sfx in ("64", "32"):
            getattr(_LIB, "etgo_create" + sfx).restype = C.c_void_p
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleSim:
    """N-robot CPU oracle with the same reset/step contract as the C-ABI."""

    def __init__(self, cfg, model=None, dtype=np.float64, threads=1):
        self.cfg = cfg
        self.model = model if model is not None else A.default_model()
        self.dtype = np.dtype(dtype)
        self.sfx = "64" if self.dtype == np.float64 else "32"
        self.N = cfg.num_envs
        self.threads = threads
        self._l = lib()
        self._h = C.c_void_p(self._f("create")(C.byref(cfg), C.byref(self.model)))
        dyn = np.tile(A.default_dynamic_row(), (self.N, 1))
        self.set_params(dyn=dyn, etg_w=np.zeros((3, A.RBF_H)), etg_b=np.zeros(3))

    def _f(self, name):
        return getattr(self._l, "etgo_" + name + self.sfx)

    def __del__(self):
        try:
            if self._h:
                self._f("destroy")(self._h)
                self._h = None
        except Exception:
            pass

    def _arr(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None:
            assert a.shape == tuple(shape), (a.shape, shape)
        return a

    def set_params(self, dyn=None, etg_w=None, etg_b=None, mask=None):
        per_env = 0
        if etg_w is not None:
            etg_w = self._arr(etg_w)
            per_env = int(etg_w.ndim == 3)
            etg_b = self._arr(etg_b)
        if dyn is not None:
            dyn = self._arr(dyn, (self.N, A.DYN_DIM))
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._f("set_params")(self._h, _p(dyn), _p(etg_w), _p(etg_b), per_env, _p(mask))

    def set_heightfield(self, heights):
        h = np.ascontiguousarray(heights, dtype=np.float32)
        self._f("set_heightfield")(self._h, _p(h))

    def set_external_force(self, force):
        """force [N,3] world-frame newtons on the trunk COM, or None to clear."""
        f = None if force is None else self._arr(force, (self.N, 3))
        self._f("set_external_force")(self._h, _p(f))

    def set_motor_strength(self, ratios, mask=None):
        """motor strength ratios [N,12] (laikago_motor.py:67-76; None = all ones) of the masked robots"""
        a = None if ratios is None else self._arr(ratios, (self.N, 12))
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._f("set_motor_strength")(self._h, _p(a), _p(mask))

    def set_sensor_noise(self, stdev, seed=0):
        """stdev[5]: motor angle, velocity, torque, rpy, rpy rate (minitaur.py:102 order); None switches it off."""
        a = None if stdev is None else np.ascontiguousarray(stdev, dtype=np.float32)
        self._f("set_sensor_noise")(self._h, _p(a), C.c_uint64(int(seed)))

    def set_reset_offsets(self, xy, mask=None):
        """start offsets [N,2] (m) of the following resets of the masked robots; None = zero."""
        a = None if xy is None else self._arr(xy, (self.N, 2))
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._f("set_reset_offsets")(self._h, _p(a), _p(mask))

    def reset(self, mask=None, obs=None):
        if obs is None:
            obs = np.zeros((self.N, A.OBS_DIM), dtype=self.dtype)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._f("reset")(self._h, _p(mask), _p(obs), self.threads)
        return obs

    def step(self, action, donef=None, want_info=True):
        action = self._arr(action, (self.N, 60 if self.cfg.motor_mode == 2 else 12))
        obs = np.zeros((self.N, A.OBS_DIM), dtype=self.dtype)
        rew = np.zeros(self.N, dtype=self.dtype)
        done = np.zeros(self.N, dtype=np.uint8)
        info = np.zeros((self.N, A.INFO_DIM), dtype=self.dtype) if want_info else None
        if donef is not None:
            donef = np.ascontiguousarray(donef, dtype=np.uint8)
        self._f("step")(self._h, _p(action), _p(donef), _p(obs), _p(rew), _p(done), _p(info),
                        self.threads)
        return obs, rew, done, info

    def run_steps(self, nsteps, action=None, threads=None, stop_at_done=True, alive=None, want_obs=False):
        """nsteps control steps with a constant action (None = zeros), every thread running its own slice of the
        robots through all the steps (persistent workers: the all-core CPU baseline).  stop_at_done: a robot's loop ends with
        its episode (the reference's loops; the fused rollouts of the HIP library) -- its state stays the terminal state; alive
        [N] uint8: robots that were finished before the call (0) are not stepped at all.  -> (return[N], length[N]) of THIS call
        (+ the last observation row each robot produced, with want_obs)"""
        ret = np.zeros(self.N, dtype=self.dtype)
        ln = np.zeros(self.N, dtype=np.int32)
        a = None if action is None else self._arr(action, (self.N, 60 if self.cfg.motor_mode == 2 else 12))
        al = None if alive is None else np.array(alive, dtype=np.uint8)     # (a copy: the call writes the flags after it back)
        self.alive_after = al
        obs = np.zeros((self.N, A.OBS_DIM), dtype=self.dtype) if want_obs else None
        self._f("run_steps")(self._h, _p(a), int(nsteps), int(threads or self.threads), _p(ret), _p(ln), int(bool(stop_at_done)), _p(al), _p(obs))
        return (ret, ln, obs) if want_obs else (ret, ln)

    def sweep_hist(self, mask=None, clear=True):
        """ticks by the number of PGS sweeps they ran, summed over the (masked) robots since the last clear -> int64[64]"""
        out = np.zeros(64, dtype=np.int64)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._f("sweep_hist")(self._h, _p(out), _p(mask), int(bool(clear)))
        return out

    def body_stats(self, clear=True):
        """per robot: ticks with a body row inside the margin, ticks with a loaded body row, all ticks -> int64[N,3]"""
        out = np.zeros((self.N, 3), dtype=np.int64)
        self._f("body_stats")(self._h, _p(out), int(bool(clear)))
        return out

    TRACE_W = 64

    def trace(self, env, cap_ticks=64):
        """start recording robot `env`'s physics ticks (cap_ticks of them; see Env::trace in etgsim_oracle.cpp for the columns)"""
        self._trace_buf = np.zeros((cap_ticks, self.TRACE_W), dtype=np.float64)
        self._trace_env = env
        self._f("set_trace")(self._h, int(env), _p(self._trace_buf), int(cap_ticks))

    def trace_all(self, cap_ticks=16):
        """record EVERY robot's physics ticks from now on into one buffer [N, cap_ticks, 64] (each call rewinds every robot's
        tick count to 0: call it before the control step of interest); the rows are read from the returned array, the number
        of ticks a robot recorded from trace_counts()"""
        if getattr(self, "_trace_all", None) is None or self._trace_all.shape[1] != cap_ticks:
            self._trace_all = np.zeros((self.N, cap_ticks, self.TRACE_W), dtype=np.float64)
        for i in range(self.N):
            self._f("set_trace")(self._h, i, _p(self._trace_all[i]), int(cap_ticks))
        return self._trace_all

    def trace_counts(self):
        return np.array([self._f("set_trace")(self._h, i, None, -1) for i in range(self.N)])

    def trace_rows(self, stop=True):
        """-> the ticks recorded since trace() as float64 [n, 64] (and stop recording)"""
        n = self._f("set_trace")(self._h, int(self._trace_env), None, 0 if stop else -1)
        return self._trace_buf[:max(n, 0)].copy()

    def get_state(self):
        st = np.zeros((self.N, A.STATE_DIM), dtype=self.dtype)
        self._f("get_state")(self._h, _p(st))
        return st

    def set_state(self, st):
        st = self._arr(st, (self.N, A.STATE_DIM))
        self._f("set_state")(self._h, _p(st))

    def tick(self, tau, nticks=1):
        tau = self._arr(tau, (self.N, 12))
        self._f("tick")(self._h, _p(tau), int(nticks))

    def set_solve_noise(self, rel, seed=0):
        """TEST KNOB: every impulse a tick's contact solve returns is multiplied by 1 + rel * U(-1, 1) before it is applied -- the
        rounding noise of an fp32 solve, for the members of tests/parity_util.OracleEnsemble (Sim::solve_noise in the C++)"""
        f = self._f("set_solve_noise")
        f.argtypes = [C.c_void_p, C.c_double, C.c_uint64]
        f(self._h, float(rel), int(seed))

    def get_lambda(self, full=False):
        """the contact impulses of the last tick: [N,12] the feet's, per leg (n, t1, t2); full=True: [N,16] = per leg (foot n, t1,
        t2, body contact's normal) -- everything the solver warm-starts from (etg_get_contact_impulses' layout)"""
        lam = np.zeros((self.N, 16), dtype=self.dtype)
        self._f("get_lambda")(self._h, _p(lam))
        return lam if full else np.ascontiguousarray(lam.reshape(self.N, 4, 4)[:, :, :3]).reshape(self.N, 12)

    def set_lambda(self, lam):
        """install the solver's warm start: [N,16] = per leg (foot n, t1, t2, body contact's normal impulse) of the last tick
        (etg_set_contact_impulses), or [N,12] = the feet's only (body contacts start cold); set_state zeroes them"""
        lam = np.asarray(lam, dtype=self.dtype)
        if lam.shape == (self.N, 12):
            lam = np.concatenate([lam.reshape(self.N, 4, 3), np.zeros((self.N, 4, 1), dtype=self.dtype)], axis=2).reshape(self.N, 16)
        lam = self._arr(lam, (self.N, 16))
        self._f("set_lambda")(self._h, _p(lam))

    def dynamics_terms(self, env=0):
        M = np.zeros((18, 18), dtype=self.dtype)
        Cb = np.zeros(18, dtype=self.dtype)
        self._f("dynamics_terms")(self._h, int(env), _p(M), _p(Cb))
        return M, Cb

    def etg_rbf(self, t):
        r = np.zeros(A.RBF_H, dtype=self.dtype)
        ct = C.c_double if self.sfx == "64" else C.c_float
        self._f("etg_rbf")(self._h, ct(t), _p(r))
        return r

    def etg_action(self, t, env=0):
        r = np.zeros(12, dtype=self.dtype)
        ct = C.c_double if self.sfx == "64" else C.c_float
        self._f("etg_action")(self._h, int(env), ct(t), _p(r))
        return r


def _ct(dtype):
    return C.c_double if np.dtype(dtype) == np.float64 else C.c_float


def leg_ik(foot, sign, dtype=np.float64):
    sfx = "64" if np.dtype(dtype) == np.float64 else "32"
    foot = np.ascontiguousarray(foot, dtype=dtype)
    out = np.zeros(3, dtype=dtype)
    getattr(lib(), "etgo_leg_ik" + sfx)(_p(foot), _ct(dtype)(sign), _p(out))
    return out


def leg_fk(ang, sign, dtype=np.float64):
    sfx = "64" if np.dtype(dtype) == np.float64 else "32"
    ang = np.ascontiguousarray(ang, dtype=dtype)
    out = np.zeros(3, dtype=dtype)
    getattr(lib(), "etgo_leg_fk" + sfx)(_p(ang), _ct(dtype)(sign), _p(out))
    return out


def leg_jacobian(ang, leg, dtype=np.float64):
    sfx = "64" if np.dtype(dtype) == np.float64 else "32"
    ang = np.ascontiguousarray(ang, dtype=dtype)
    out = np.zeros((3, 3), dtype=dtype)
    getattr(lib(), "etgo_leg_jacobian" + sfx)(_p(ang), int(leg), _p(out))
    return out


def pd_torque(qdes, q, qd, kp, kd, dtype=np.float64):
    sfx = "64" if np.dtype(dtype) == np.float64 else "32"
    arrs = [np.ascontiguousarray(a, dtype=dtype) for a in (qdes, q, qd, kp, kd)]
    out = np.zeros(len(arrs[0]), dtype=dtype)
    getattr(lib(), "etgo_pd_torque" + sfx)(*[_p(a) for a in arrs], len(out), _p(out))
    return out


def motor_torque(qdes, q, qd, kp, kd, strength, limit=0.0, torque_mode=False, dtype=np.float64):
    """the motor model with strength ratios and torque limit (laikago_motor.py:103-175)"""
    sfx = "64" if np.dtype(dtype) == np.float64 else "32"
    arrs = [np.ascontiguousarray(a, dtype=dtype) for a in (qdes, q, qd, kp, kd, strength)]
    out = np.zeros(len(arrs[0]), dtype=dtype)
    ct = C.c_double if sfx == "64" else C.c_float
    getattr(lib(), "etgo_motor_torque" + sfx)(*[_p(a) for a in arrs], ct(limit), int(bool(torque_mode)), len(out), _p(out))
    return out


def pd_torque_hybrid(cmd, q, qd, dtype=np.float64):
    """HYBRID motor command (laikago_motor.py:152-167): cmd[60] = 12 x (q_des, kp, qd_des, kd, tau_ff)."""
    sfx = "64" if np.dtype(dtype) == np.float64 else "32"
    arrs = [np.ascontiguousarray(a, dtype=dtype) for a in (cmd, q, qd)]
    out = np.zeros(12, dtype=dtype)
    getattr(lib(), "etgo_pd_torque_hybrid" + sfx)(*[_p(a) for a in arrs], 12, _p(out))
    return out


def mlp_forward(obs, w1, b1, w2, b2, w3, b3, scale=1.0, dtype=np.float64):
    sfx = "64" if np.dtype(dtype) == np.float64 else "32"
    obs = np.ascontiguousarray(obs, dtype=dtype)
    ws = [np.ascontiguousarray(a, dtype=dtype) for a in (w1, b1, w2, b2, w3, b3)]
    n, in_dim = obs.shape
    hid, out_dim = ws[0].shape[0], ws[4].shape[0]
    act = np.zeros((n, out_dim), dtype=dtype)
    getattr(lib(), "etgo_mlp_forward" + sfx)(_p(obs), n, in_dim, hid, out_dim,
                                              *[_p(a) for a in ws], _ct(dtype)(scale), _p(act))
    return act


def mlp_sample(obs, w1, b1, w2, b2, w3, b3, ws, bs, noise, scale=1.0):
    """SAC.sample restated in numpy fp64 (alg/sac.py:65-76 over Actor.forward, model/mujoco_model.py:53-60):
    -> (tanh(mean + exp(clamp(log_std, -20, 2)) * noise) * scale, log-probability [n,1])."""
    f = np.float64
    obs, noise = np.asarray(obs, f), np.asarray(noise, f)
    h = np.maximum(obs @ np.asarray(w1, f).T + np.asarray(b1, f), 0)
    h = np.maximum(h @ np.asarray(w2, f).T + np.asarray(b2, f), 0)
    mean = h @ np.asarray(w3, f).T + np.asarray(b3, f)
    log_std = np.clip(h @ np.asarray(ws, f).T + np.asarray(bs, f), -20.0, 2.0)
    x = mean + np.exp(log_std) * noise
    a = np.tanh(x)
    logp = (-0.5 * noise ** 2 - log_std - 0.5 * np.log(2 * np.pi)) - np.log((1 - a ** 2) + 1e-6)
    return a * scale, logp.sum(1, keepdims=True)
