"""ctypes binding of the C-ABI in include/etgsim.h (paddlerobotics_amd/csrc/libetgsim.so).

There is no CPU fallback: if the library is missing this module raises, and if no HIP
device is visible etg_create() fails with ETG_ERR_NO_DEVICE.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

SYMBOLS = [
    "etg_create", "etg_destroy", "etg_last_error", "etg_version", "etg_lanes_per_robot", "etg_set_params",
    "etg_set_heightfield", "etg_set_external_force", "etg_set_motor_strength", "etg_random_pushes", "etg_clear_pushes", "etg_set_reset_offsets", "etg_set_sensor_noise", "etg_reset", "etg_step", "etg_episode_stats", "etg_rollout_openloop",
    "etg_get_state",
    "etg_set_state", "etg_policy_create", "etg_policy_load", "etg_policy_forward", "etg_policy_load_std", "etg_policy_sample",
    "etg_policy_destroy", "etg_rollout_policy", "etg_fit_etg", "etg_leg_kinematics", "etg_extra_sensors", "etg_step_autoreset",
    "etg_replay_begin", "etg_replay_end", "etg_rollout_policy_record", "etg_rollout_actions",
    "etg_prepare_next_dynamics", "etg_next_dynamics_pending",
    "etg_config_size", "etg_model_size", "etg_get_contact_impulses", "etg_set_contact_impulses", "etg_set_rollout_mode", "etg_rollout_wave_cycles", "etg_step_range",
]
ABI_VERSION = 2      # include/etgsim.h: etg_version()


class EtgError(RuntimeError):
    pass


def lib_path():
    # ETG_LIB selects an alternative build of the same HIP sources (A/B experiments in tools/)
    return os.environ.get("ETG_LIB", _build.LIB)


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise EtgError(
            "paddlerobotics_amd: %s is missing -- build it with `python -m paddlerobotics_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    lib.etg_last_error.restype = C.c_char_p
    vp, i32 = C.c_void_p, C.c_int
    lib.etg_create.argtypes = [vp, vp, i32, C.POINTER(vp)]
    lib.etg_destroy.argtypes = [vp]
    lib.etg_destroy.restype = None
    lib.etg_lanes_per_robot.argtypes = [vp]
    lib.etg_set_params.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    lib.etg_set_heightfield.argtypes = [vp, vp, vp]
    lib.etg_set_external_force.argtypes = [vp, vp, vp]
    lib.etg_set_motor_strength.argtypes = [vp, vp, vp, vp]
    lib.etg_random_pushes.argtypes = [vp, C.c_uint64, C.c_float, i32, C.c_float, C.c_float, vp]
    lib.etg_clear_pushes.argtypes = [vp, vp, vp]
    lib.etg_set_reset_offsets.argtypes = [vp, vp, vp, vp]
    lib.etg_set_sensor_noise.argtypes = [vp, vp, C.c_uint64]
    lib.etg_reset.argtypes = [vp, vp, vp, vp]
    lib.etg_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.etg_step_autoreset.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.etg_step_range.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.etg_episode_stats.argtypes = [vp, vp, vp, vp]
    lib.etg_rollout_openloop.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.etg_get_state.argtypes = [vp, vp, vp]
    lib.etg_set_state.argtypes = [vp, vp, vp]
    lib.etg_set_rollout_mode.argtypes = [vp, i32]
    lib.etg_rollout_wave_cycles.argtypes = [vp, vp, i32, C.POINTER(i32), C.POINTER(i32), vp]
    lib.etg_get_contact_impulses.argtypes = [vp, vp, vp]
    lib.etg_set_contact_impulses.argtypes = [vp, vp, vp]
    lib.etg_leg_kinematics.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.etg_extra_sensors.argtypes = [vp, vp, vp, vp]
    lib.etg_policy_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    lib.etg_policy_load.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.etg_policy_forward.argtypes = [vp, vp, i32, C.c_float, i32, vp, vp]
    lib.etg_policy_load_std.argtypes = [vp, vp, vp, vp]
    lib.etg_policy_sample.argtypes = [vp, vp, i32, vp, C.c_float, i32, vp, vp, vp]
    lib.etg_policy_destroy.argtypes = [vp]
    lib.etg_rollout_policy.argtypes = [vp, vp, i32, C.c_float, i32, i32, vp, vp, vp, vp]
    lib.etg_rollout_policy_record.argtypes = [vp, vp, i32, C.c_float, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.etg_rollout_actions.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.etg_prepare_next_dynamics.argtypes = [vp, vp, vp, vp]
    lib.etg_next_dynamics_pending.argtypes = [vp, vp, vp]
    lib.etg_policy_destroy.restype = None
    dbl = C.c_double
    lib.etg_fit_etg.argtypes = [vp, i32, vp, vp, dbl, dbl, dbl, dbl, dbl, i32, vp, vp, vp]
    ll = C.c_longlong
    lib.etg_replay_begin.argtypes = [vp, i32, ll, vp, vp, vp, i32, vp, i32, vp, vp, C.c_float, vp, vp]
    lib.etg_replay_end.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    # a library built from another revision of include/etgsim.h would read past (or short of) the structs this binding
    # passes to etg_create: refuse it here, with the reason, instead of simulating with garbage parameters
    from . import a1_model as _A
    if lib.etg_version() != ABI_VERSION or lib.etg_config_size() != C.sizeof(_A.EtgConfig) or lib.etg_model_size() != C.sizeof(_A.EtgRobotModel):
        raise EtgError("paddlerobotics_amd: %s has ABI version %d with EtgConfig / EtgRobotModel of %d / %d bytes; this binding is "
                       "version %d with %d / %d bytes -- rebuild the library (python -m paddlerobotics_amd.build)" % (
                           path, lib.etg_version(), lib.etg_config_size(), lib.etg_model_size(), ABI_VERSION,
                           C.sizeof(_A.EtgConfig), C.sizeof(_A.EtgRobotModel)))
    _LIB = lib
    return lib


ETG_ERR_STATE = -5   # include/etgsim.h


def check(code):
    if code != 0:
        raise EtgError("etgsim error %d: %s" % (code, load().etg_last_error().decode()))
