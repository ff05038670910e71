"""CPU suite: an ended episode is not simulated any more (fused rollouts; KCfg.stop_at_done).

The reference's episode loops leave at `done` (pretrain.py:137-153, train.py:226-247).  The fused rollouts of the HIP library do
the same per robot since round 6: the robot's state stays the terminal state, its observation row the one of the step that ended
the episode, return / length what they were, its control variables (ETG phase, last command, ...) frozen.  Checked here without
a GPU:
  * the oracle's run_steps (the CPU baseline's loop) against its own stepping loop with break-at-done, per robot;
  * the kernel source compiled for the host (tests/emu: rollout_steps16 / rollout_steps, i.e. what k_rollout16 / k_rollout run)
    against the emulation's stepping loop with break-at-done -- BIT-identical (one source, no contraction on the host), both
    lane mappings, with and without body rows; and, with simulate_finished, against stepping every robot through all the steps
    (the behaviour of rounds 1-5);
  * a second rollout call does not touch a robot that finished in the first;
  * the CPU build of the C-ABI (oracle/libetgsim_cpu.so): etg_rollout_openloop / etg_set_rollout_mode.
The -m gpu counterparts (a finished robot in a wave of running ones, the closed-loop and tape kernels, sensor noise):
tests/test_gpu_stop_at_done.py."""
import ctypes as C

import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points


def _gait(n, seed=0):
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    rng = np.random.default_rng(seed)
    W, B = np.zeros((n, 3, 20)), np.zeros((n, 3))
    for i in range(n):
        W[i], B[i], _ = Opt_with_points(layer, ETG_T=0.5, w0=w0, b0=b0, points=prior + 0.02 * rng.normal(size=(6, 2)))
    return W, B


def _pushes(n):
    """lateral forces on the trunk that throw some robots over within 5..40 control steps and leave others standing"""
    f = np.zeros((n, 3))
    f[:, 1] = np.linspace(0.0, 70.0, n)
    return f


def _break_at_done(sim, steps, n):
    """the reference's loop per robot on a stepping simulator: states / rows / accumulators as they were at the step that ended
    each robot's episode (robots are independent: the others being stepped on does not matter)"""
    st_end, obs_end = None, None
    ret, ln, alive = np.zeros(n), np.zeros(n, int), np.ones(n, bool)
    for k in range(steps):
        obs, r, d, _ = sim.step(np.zeros((n, 12)))
        st = np.asarray(sim.get_state()).copy()
        if st_end is None:
            st_end, obs_end = st.copy(), np.asarray(obs).copy()
        st_end[alive], obs_end[alive] = st[alive], np.asarray(obs)[alive]
        ret += alive * r
        ln += alive
        alive &= ~np.asarray(d).astype(bool)
    return st_end, obs_end, ret, ln, alive


def test_oracle_run_steps_leaves_the_loop_at_done():
    from oracle.oracle import OracleSim
    n, steps = 8, 45
    W, B = _gait(n)
    a, b = OracleSim(A.default_config(n, settle_ticks=200)), OracleSim(A.default_config(n, settle_ticks=200))
    for o in (a, b):
        o.set_params(etg_w=W, etg_b=B)
        o.reset()
        o.set_external_force(_pushes(n))
    st_end, obs_end, ret, ln, alive = _break_at_done(b, steps, n)
    assert 2 <= int((~alive).sum()) <= n - 2, ln                  # some fell, some walk on: the test means something
    ret_a, ln_a, obs_a = a.run_steps(steps, want_obs=True)
    assert np.array_equal(ln_a, ln) and np.allclose(ret_a, ret, rtol=0, atol=1e-12)
    assert np.array_equal(a.get_state(), st_end)                  # the terminal state, bit for bit
    assert np.array_equal(obs_a, obs_end)
    # a second call does not touch the finished robots and carries the others on
    st_before = a.get_state().copy()
    ret2, ln2 = a.run_steps(5, alive=alive.astype(np.uint8))
    assert np.array_equal(a.get_state()[~alive], st_before[~alive]) and np.all(ln2[~alive] == 0)
    assert not np.array_equal(a.get_state()[alive], st_before[alive])
    # stop_at_done = False: the behaviour of rounds 1-5 (stepped on, accumulators masked)
    c = OracleSim(A.default_config(n, settle_ticks=200))
    c.set_params(etg_w=W, etg_b=B); c.reset(); c.set_external_force(_pushes(n))
    ret_c, ln_c = c.run_steps(steps, stop_at_done=False)
    assert np.array_equal(ln_c, ln) and np.allclose(ret_c, ret, rtol=0, atol=1e-12)
    assert np.array_equal(c.get_state(), b.get_state())


@pytest.mark.parametrize("lanes,body", [(16, 2), (16, 0), (4, 2), (4, 0)])
def test_kernel_source_rollout_stops_at_done_bit_for_bit(lanes, body):
    from tests.emu.emu import EmuSim
    n, steps = 6, 40
    W, B = _gait(n, seed=3)
    kw = dict(settle_ticks=150, body_contacts=body)
    fused, stepped, full = (EmuSim(A.default_config(n, **kw), lanes=lanes) for _ in range(3))
    for e in (fused, stepped, full):
        e.set_params(etg_w=W, etg_b=B)
        e.reset()
        e.set_external_force(_pushes(n))
    st_end, obs_end, ret, ln, alive = _break_at_done(stepped, steps, n)
    assert int((~alive).sum()) >= 1 and (body == 0 or alive.any()), ln     # (toe spheres only: the shins pass through the floor, every robot ends)
    # two launches (25 + 15 steps): a robot that ends in the first is not touched by the second
    marker = np.full((n, A.OBS_DIM), 7.0, dtype=np.float32)
    obs_f, ret_f, ln_f = fused.rollout_openloop(25, obs=marker)
    obs_f, ret_f, ln_f = fused.rollout_openloop(15, obs=obs_f)
    assert np.array_equal(ln_f, ln)
    assert np.array_equal(fused.get_state(), st_end.astype(np.float32))           # the terminal state, bit for bit
    assert np.array_equal(obs_f, obs_end.astype(np.float32))                      # every robot's LAST row (none still the marker)
    assert np.allclose(ret_f, ret, rtol=0, atol=1e-4)                             # (fp32 sums accumulated in the same order)
    # control variables of a finished robot are frozen: stepping it on from here equals stepping the reference on from its end
    # simulate_finished: every robot goes through all the steps, accumulators masked (rounds 1-5)
    obs_g, ret_g, ln_g = full.rollout_openloop(steps, stop_at_done=False)
    assert np.array_equal(ln_g, ln) and np.allclose(ret_g, ret, rtol=0, atol=1e-4)
    assert np.array_equal(full.get_state(), stepped.get_state())
    assert not np.array_equal(full.get_state()[~alive], st_end[~alive].astype(np.float32))


def test_cpu_abi_rollout_openloop_stops_at_done():
    from oracle import oracle as O
    lib = C.CDLL(O.build_cpu_abi())
    lib.etg_last_error.restype = C.c_char_p
    n, steps = 6, 40
    W, B = _gait(n, seed=5)
    cfg, model = A.default_config(n, settle_ticks=150), A.default_model()
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def make():
        h = C.c_void_p()
        assert lib.etg_create(C.byref(cfg), C.byref(model), -1, C.byref(h)) == 0, lib.etg_last_error()
        w, b, obs, force = f32(W), f32(B), np.zeros((n, A.OBS_DIM), np.float32), f32(_pushes(n))
        assert lib.etg_set_params(h, None, p(w), p(b), 1, None, None) == 0
        assert lib.etg_reset(h, None, p(obs), None) == 0
        assert lib.etg_set_external_force(h, p(force), None) == 0
        return h, obs

    h1, obs1 = make()
    h2, obs2 = make()
    ret1, ln1 = np.zeros(n, np.float32), np.zeros(n, np.int32)
    assert lib.etg_rollout_openloop(h1, steps, p(obs1), p(ret1), p(ln1), None) == 0, lib.etg_last_error()
    # the stepping loop with break-at-done on the second handle
    st_end, obs_end = np.zeros((n, A.STATE_DIM), np.float32), np.zeros((n, A.OBS_DIM), np.float32)
    alive, ln = np.ones(n, bool), np.zeros(n, int)
    rew, done, st = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros((n, A.STATE_DIM), np.float32)
    for k in range(steps):
        assert lib.etg_step(h2, None, None, p(obs2), p(rew), p(done), None, None) == 0
        assert lib.etg_get_state(h2, p(st), None) == 0
        st_end[alive], obs_end[alive] = st[alive], obs2[alive]
        ln += alive
        alive &= ~done.astype(bool)
    assert 1 <= int((~alive).sum()) <= n - 1
    st1 = np.zeros((n, A.STATE_DIM), np.float32)
    assert lib.etg_get_state(h1, p(st1), None) == 0
    assert np.array_equal(ln1, ln) and np.array_equal(st1, st_end) and np.array_equal(obs1, obs_end)
    # etg_set_rollout_mode(h, 1): finished robots are stepped on
    h3, obs3 = make()
    assert lib.etg_set_rollout_mode(h3, 1) == 0
    ret3, ln3 = np.zeros(n, np.float32), np.zeros(n, np.int32)
    assert lib.etg_rollout_openloop(h3, steps, p(obs3), p(ret3), p(ln3), None) == 0
    st3 = np.zeros((n, A.STATE_DIM), np.float32)
    lib.etg_get_state(h3, p(st3), None)
    assert np.array_equal(ln3, ln) and np.array_equal(st3, st)
    for h in (h1, h2, h3):
        lib.etg_destroy(h)
