"""ctypes wrapper of the TEST-ONLY host emulation of the HIP kernels (etg_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from paddlerobotics_amd import a1_model as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        # ETG_EMU_FLAGS: extra compiler flags (e.g. -DETG_TRUNK_ON_AUX) to emulate a build variant of the kernel source
        # (the per-tick trace of physics_tick16 -- set_trace(), tests/divergence.py -- is always compiled into the emulation)
        flags = [f for f in os.environ.get("ETG_EMU_FLAGS", "").split() if f != "-DETG_TRACE_TICKS"]
        so = os.path.join(_HERE, "libetg_emu%s.so" % ("_" + "".join(ch for ch in "".join(flags) if ch.isalnum()) if flags else ""))
        flags = flags + ["-DETG_TRACE_TICKS"]
        srcs = [os.path.join(_HERE, "etg_emu.cpp"), os.path.join(_HERE, "emu_lanes.h")] + [
            os.path.join(_HERE, "..", "..", "paddlerobotics_amd", "csrc", f)
            for f in ("etg_core.h", "etg_core16.h", "etg_layout.h") if os.path.exists(
                os.path.join(_HERE, "..", "..", "paddlerobotics_amd", "csrc", f))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"] + flags +
                                  ["-o", so, srcs[0]])
        _LIB = C.CDLL(so)
        _LIB.emu_create.restype = C.c_void_p
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class EmuSim:
    def __init__(self, cfg, model=None, lanes=4):
        """lanes: 4 = etg_core.h (one leg per lane), 16 = etg_core16.h (one robot per DPP row)."""
        self.cfg, self.model = cfg, model if model is not None else A.default_model()
        self.N = cfg.num_envs
        self.lanes = lanes
        self._l = lib()
        self._h = C.c_void_p(self._l.emu_create(C.byref(cfg), C.byref(self.model)))
        self._l.emu_set_lanes(self._h, int(lanes))
        self.set_params(dyn=np.tile(A.default_dynamic_row(), (self.N, 1)), etg_w=np.zeros((3, 20)), etg_b=np.zeros(3))

    def __del__(self):
        try:
            self._l.emu_destroy(self._h)
        except Exception:
            pass

    def set_params(self, dyn=None, etg_w=None, etg_b=None, mask=None):
        per_env = 0
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        dyn, etg_w, etg_b = f(dyn), f(etg_w), f(etg_b)
        if etg_w is not None:
            per_env = int(etg_w.ndim == 3)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._l.emu_set_params(self._h, _p(dyn), _p(etg_w), _p(etg_b), per_env, _p(mask))

    def set_heightfield(self, h):
        self._hf = np.ascontiguousarray(h, dtype=np.float32)
        self._l.emu_set_heightfield(self._h, _p(self._hf))

    def set_external_force(self, force):
        f = None if force is None else np.ascontiguousarray(force, dtype=np.float32)
        self._l.emu_set_external_force(self._h, _p(f))

    def set_motor_strength(self, ratios):
        a = None if ratios is None else np.ascontiguousarray(ratios, dtype=np.float32)
        self._l.emu_set_motor_strength(self._h, _p(a))

    def set_sensor_noise(self, stdev, seed=0):
        import ctypes
        a = None if stdev is None else np.ascontiguousarray(stdev, dtype=np.float32)
        self._l.emu_set_sensor_noise(self._h, _p(a), ctypes.c_uint64(seed))

    def set_reset_offsets(self, xy):
        a = None if xy is None else np.ascontiguousarray(xy, dtype=np.float32)
        self._l.emu_set_reset_offsets(self._h, _p(a))

    def reset(self, mask=None):
        obs = np.zeros((self.N, A.OBS_DIM), dtype=np.float32)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._l.emu_reset(self._h, _p(mask), _p(obs))
        return obs

    def step(self, action, donef=None):
        action = np.ascontiguousarray(action, dtype=np.float32)
        obs = np.zeros((self.N, A.OBS_DIM), dtype=np.float32)
        rew = np.zeros(self.N, dtype=np.float32)
        done = np.zeros(self.N, dtype=np.uint8)
        info = np.zeros((self.N, A.INFO_DIM), dtype=np.float32)
        if donef is not None:
            donef = np.ascontiguousarray(donef, dtype=np.uint8)
        self._l.emu_step(self._h, _p(action), _p(donef), _p(obs), _p(rew), _p(done), _p(info))
        return obs, rew, done, info

    def rollout_openloop(self, n_steps, stop_at_done=True, obs=None):
        """the fused open-loop rollout (rollout_steps16 / rollout_steps) -> (obs [N,49] rows the rollout wrote, ret [N], len [N]);
        obs: the buffer the rows land in (rows of robots that were finished before the call keep their content)"""
        if obs is None:
            obs = np.zeros((self.N, A.OBS_DIM), dtype=np.float32)
        ret = np.zeros(self.N, dtype=np.float32)
        ln = np.zeros(self.N, dtype=np.int32)
        self._l.emu_rollout_openloop(self._h, int(n_steps), int(bool(stop_at_done)), _p(obs), _p(ret), _p(ln))
        return obs, ret, ln

    def get_state(self):
        st = np.zeros((self.N, A.STATE_DIM), dtype=np.float32)
        self._l.emu_get_state(self._h, _p(st))
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, dtype=np.float32)
        self._l.emu_set_state(self._h, _p(st))

    def set_contact_impulses(self, lam):
        lam = np.ascontiguousarray(lam, dtype=np.float32)
        self._l.emu_set_contact_impulses(self._h, _p(lam))

    def get_contact_impulses(self):
        lam = np.zeros((self.N, 16), dtype=np.float32)
        self._l.emu_get_contact_impulses(self._h, _p(lam))
        return lam

    def set_trace(self, on=True):
        """tick trace of an ETG_EMU_FLAGS=-DETG_TRACE_TICKS build: [N,16,16,10] float32 (tools/first_divergence.py)"""
        self._trace = np.zeros((self.N, 16, 16, 10), dtype=np.float32) if on else None
        self._l.emu_debug_set_trace(self._h, _p(self._trace))
        return self._trace

    @staticmethod
    def set_extra_sweeps(n):
        """every emulated robot sits through n more sweeps after its own convergence, frozen (what a robot on the GPU does
        while its wave neighbours still sweep); process-wide"""
        lib().emu_set_extra_sweeps(int(n))

    @staticmethod
    def set_force_body(on):
        """every emulated robot takes the body-row paths as if a wave neighbour had a body sphere in the margin / under load
        (16-lane mapping); process-wide"""
        lib().emu_set_force_body(int(bool(on)))

    def replication_check(self, env=0, nticks=50):
        f = self._l.emu16_tick_replication_check if self.lanes == 16 else self._l.emu_tick_replication_check
        return f(self._h, int(env), int(nticks))
