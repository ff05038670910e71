"""CPU suite: the oracle and the host emulation of the kernel math under AddressSanitizer + UndefinedBehaviorSanitizer
(SURVEY section 5 asks for a sanitizer build of the CPU side).  A driver linked from the SAME sources runs both precisions of
the oracle and both lane mappings of the emulation through a set of configurations -- default rule, fixed sweeps, heightfield,
body contacts + joint limits, every motor mode, filter / interpolation -- and must finish without a report."""
import ctypes as C
import os
import subprocess

import numpy as np

from paddlerobotics_amd import a1_model as A

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sanitize")


def test_oracle_and_kernel_math_run_clean_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-C", HERE, "sanitize_driver"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(0)
    hf = dict(heights=rng.uniform(0, 0.05, size=(64, 64)).astype(np.float32), cell=0.05, origin=(-1.6, -1.6))
    scen = [
        (dict(), 3),
        (dict(solver_iters=3), 2),
        (dict(terrain=1, heightfield=hf), 3),
        (dict(terrain=1, heightfield=hf, body_contacts=2, joint_limits=1, motor_mode=1), 3),
        (dict(body_contacts=1, friction_model=1), 2),
        (dict(body_contacts=3, motor_mode=1, joint_limits=0), 3),
        (dict(motor_mode=2, enable_action_filter=True, enable_action_interp=True, torque_limit=30.0, clip_motor_commands=0.2), 2),
        (dict(enable_etg=0, solver_iters=2, solver_residual=1e-5), 2),
        (dict(pd_latency=0.0013), 2),
        (dict(foot_restitution=0.5, warmstart=0.85, warmstart_friction=0.85, contact_slop=0.0, clip_motor_commands=0.1), 2),
    ]
    model = A.default_model()
    path = tmp_path / "scenarios.bin"
    with open(path, "wb") as f:
        f.write(np.int32(len(scen)).tobytes())
        for kw, steps in scen:
            cfg = A.default_config(3, settle_ticks=40, **kw)
            f.write(bytes(cfg))
            f.write(bytes(model))
            f.write(np.int32(steps).tobytes())
            if cfg.terrain == 1:
                f.write(np.ascontiguousarray(kw["heightfield"]["heights"], dtype=np.float32).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(HERE, "sanitize_driver"), str(path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "SANITIZE OK" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
