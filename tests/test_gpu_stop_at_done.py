"""-m gpu: an ended episode is not simulated any more (fused rollouts; KCfg.stop_at_done, include/etgsim.h "fused rollouts").

CPU counterpart: tests/test_stop_at_done.py (oracle, the kernel source on the host bit for bit, the CPU build of the ABI).  What
only the device can show:
  * a finished robot shares its wavefront with running ones: the running robots' results do not depend on it, BIT for bit (the
    same robots next to neighbours that never fall);
  * stop_at_done and simulate_finished give bit-identical returns / lengths and bit-identical states of the robots still running;
  * a finished robot's state / observation row / accumulators are not touched by later launches (chunks of 50 steps, later calls);
  * against the stepping loop with break-at-done (the reference's loop per robot: pretrain.py:137-153, train.py:226-247) -- another
    kernel, so to rounding noise: open loop, closed loop, action tape, recorded closed loop; both lane mappings;
  * sensor noise: a robot's last row carries the noise of the stream position of the step that wrote it.
"""
import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_gpu_parity import _need_gpu, _etg_params, _make   # noqa: E402
from tests.test_gpu_parity2 import _policy                       # noqa: E402


def _say(*a):
    print("[parity]", *a, flush=True)


def _pushes(n, top=70.0, every=1):
    """lateral trunk forces: robot i gets top * (i % k) / k ... some fall within 5..40 steps, some never"""
    f = np.zeros((n, 3), dtype=np.float32)
    f[:, 1] = np.linspace(0.0, top, n)
    if every > 1:
        f[np.arange(n) % every != 0] = 0.0
    return torch.as_tensor(f)


def _stepping_break_at_done(env, steps, act_fn=None):
    """the reference's loop per robot on the stepping kernels: state / row / accumulators at the step that ended each episode"""
    n = env.num_envs
    alive = torch.ones(n, dtype=torch.bool, device=env.device)
    st_end, obs_end = env.get_state().clone(), env.obs.clone()
    ln = torch.zeros(n, dtype=torch.int32, device=env.device)
    ret = torch.zeros(n, device=env.device)
    for k in range(steps):
        a = None if act_fn is None else act_fn(env, k)
        _, r, d, _ = env.step(a, want_info=False)
        st = env.get_state()
        st_end[alive], obs_end[alive] = st[alive], env.obs[alive]
        ret += alive.float() * r
        ln += alive.int()
        alive &= ~d.view(-1).bool()
    return st_end, obs_end, ret, ln, alive


def _compare_with_stepping(tag, fused_state, fused_obs, ln_f, ret_f, st_end, obs_end, ln, ret, floor=2e-4):
    """fused vs stepping are two kernels of one source: equal to rounding noise amplified by the contacts -- a borderline fall may
    move by a step, so: lengths equal for >= 95 % of the robots, and on those the terminal state / last row agree"""
    same = (ln_f == ln).cpu().numpy()
    gq = (fused_state - st_end)[:, 13:25].abs().max(1).values.cpu().numpy()
    go = (fused_obs - obs_end).abs().max(1).values.cpu().numpy()
    gr = (ret_f - ret).abs().cpu().numpy()
    _say("%s: episode lengths equal %d / %d | on those: joint gap median %.1e q95 %.1e, last-row gap median %.1e, return gap median %.1e"
         % (tag, same.sum(), len(same), np.median(gq[same]), np.quantile(gq[same], 0.95), np.median(go[same]), np.median(gr[same])))
    assert same.mean() >= 0.95, tag
    assert np.median(gq[same]) < 0.1 * floor and np.quantile(gq[same], 0.9) < floor, tag
    assert np.median(go[same]) < 50 * floor, tag            # (normalised rows: x10 angles, x38 displacement)
    assert np.median(gr[same]) < 5e-3, tag


@pytest.fixture(autouse=True)
def _launches_of_50_steps(monkeypatch):
    """every env of this file cuts its fused rollouts into launches of 50 control steps (ETG_ROLLOUT_CHUNK, read by etg_create;
    the default is 400): what a later LAUNCH does to a finished robot is part of what is tested here"""
    monkeypatch.setenv("ETG_ROLLOUT_CHUNK", "50")


@pytest.mark.parametrize("lanes", [16, 4])
def test_open_loop_rollout_stops_at_done(lanes):
    _need_gpu()
    n, steps = 256, 80                                         # 80 steps = two launches (50 + 30)
    W, B = _etg_params(64, seed=3)
    W, B = np.tile(W, (n // 64, 1, 1)), np.tile(B, (n // 64, 1))
    f = _pushes(n)
    envs = [_make(n, lanes_per_robot=lanes) for _ in range(3)]
    fused, stepped, full = envs
    for e in envs:
        e.reset(ETG_w=W, ETG_b=B)
        e.set_external_force(f)
    full.set_rollout_mode(simulate_finished=True)
    st_end, obs_end, ret, ln, alive = _stepping_break_at_done(stepped, steps)
    n_done = int((~alive).sum())
    assert 0.2 * n < n_done < 0.9 * n, n_done                   # finished and running robots share wavefronts
    ret_f, ln_f = fused.rollout_openloop(steps)
    _compare_with_stepping("open loop lanes %d" % lanes, fused.get_state(), fused.obs, ln_f, ret_f, st_end, obs_end, ln, ret)
    # both modes: identical accumulators, identical states of the robots still running -- bit for bit (same kernel, the finished
    # robots' lanes are the only difference)
    ret_g, ln_g = full.rollout_openloop(steps)
    assert torch.equal(ln_g, ln_f) and torch.equal(ret_g, ret_f)
    run = ln_f == steps
    assert torch.equal(full.get_state()[run], fused.get_state()[run])
    fin = ~run
    assert not torch.equal(full.get_state()[fin], fused.get_state()[fin])       # (they were simulated on there)
    # a later call leaves the finished robots alone and carries the others on
    st0, obs0 = fused.get_state().clone(), fused.obs.clone()
    ret2, ln2 = fused.rollout_openloop(7)
    assert torch.equal(fused.get_state()[fin], st0[fin]) and torch.equal(fused.obs[fin], obs0[fin])
    assert torch.equal(ln2[fin], ln_f[fin]) and torch.equal(ret2[fin], ret_f[fin])
    still = ln2 == steps + 7
    assert still.any() and not torch.equal(fused.get_state()[still], st0[still])
    for e in envs:
        e.close()


@pytest.mark.parametrize("lanes", [16, 4])
@pytest.mark.parametrize("mode", ["open", "policy"])
def test_running_robots_do_not_depend_on_finished_wave_neighbours(lanes, mode):
    """The same robots twice: once next to neighbours with the same sane gait (nobody ends early), once with three of every four
    robots given a gait that folds them up within a few steps (ETG offset: feet pulled 0.2 m up), so that every wavefront holds
    finished robots next to a running one -- on the DEFAULT kernel instantiation (plain robot layer + body rows).  The sane
    robots' states, rows, returns and lengths are bit-identical in the two runs."""
    _need_gpu()
    n, steps = 256, 60
    W, B = _etg_params(64, seed=5)
    W, B = np.tile(W, (n // 64, 1, 1)), np.tile(B, (n // 64, 1))
    keep = np.arange(n) % 4 == 0
    Wd, Bd = W.copy(), B.copy()
    Wd[~keep] = 0.0
    Bd[~keep] = np.array([0.0, 0.0, 0.2])
    pol, _ = _policy()
    out = []
    for w, b in ((W, B), (Wd, Bd)):
        env = _make(n, lanes_per_robot=lanes)
        env.reset(ETG_w=w, ETG_b=b)
        ret, ln = env.rollout_openloop(steps) if mode == "open" else env.rollout_policy(pol, steps, 0.3, fused=True)
        out.append((env.get_state().clone(), env.obs.clone(), ret.clone(), ln.clone()))
        env.close()
    k = torch.as_tensor(keep, device="cuda:0")
    fell = (out[1][3][~k] < steps).float().mean().item()
    _say("wave neighbours, lanes %d %s: %.0f %% of the folded robots finished early (after %.1f steps on average); sane robots still "
         "running %.0f %%" % (lanes, mode, 100 * fell, out[1][3][~k].float().mean().item(), 100 * (out[1][3][k] == steps).float().mean().item()))
    assert fell > 0.9
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a[k], b[k])


@pytest.mark.parametrize("lanes", [16, 4])
def test_closed_loop_rollout_stops_at_done(lanes):
    _need_gpu()
    n, steps = 128, 60
    W, B = _etg_params(64, seed=7)
    W, B = np.tile(W, (n // 64, 1, 1)), np.tile(B, (n // 64, 1))
    f = _pushes(n)
    pol, _ = _policy()
    fused, stepped, full = (_make(n, lanes_per_robot=lanes) for _ in range(3))
    for e in (fused, stepped, full):
        e.reset(ETG_w=W, ETG_b=B)
        e.set_external_force(f)
    full.set_rollout_mode(simulate_finished=True)
    st_end, obs_end, ret, ln, alive = _stepping_break_at_done(stepped, steps, lambda e, k: pol.predict(e.obs, 0.3))
    assert 0.2 * n < int((~alive).sum()) < 0.95 * n
    ret_f, ln_f = fused.rollout_policy(pol, steps, 0.3, fused=True)
    _compare_with_stepping("closed loop lanes %d" % lanes, fused.get_state(), fused.obs, ln_f, ret_f, st_end, obs_end, ln, ret)
    ret_g, ln_g = full.rollout_policy(pol, steps, 0.3, fused=True)
    assert torch.equal(ln_g, ln_f) and torch.equal(ret_g, ret_f)
    run = ln_f == steps
    assert torch.equal(full.get_state()[run], fused.get_state()[run]) and torch.equal(full.obs[run], fused.obs[run])
    st0, obs0 = fused.get_state().clone(), fused.obs.clone()
    fused.rollout_policy(pol, 5, 0.3, fused=True)
    assert torch.equal(fused.get_state()[~run], st0[~run]) and torch.equal(fused.obs[~run], obs0[~run])
    for e in (fused, stepped, full):
        e.close()


@pytest.mark.parametrize("lanes", [16, 4])
def test_action_tape_rollout_stops_at_done(lanes):
    _need_gpu()
    n, T = 128, 60
    W, B = _etg_params(64, seed=9)
    W, B = np.tile(W, (n // 64, 1, 1)), np.tile(B, (n // 64, 1))
    f = _pushes(n)
    rng = np.random.default_rng(4)
    tape = torch.as_tensor(rng.uniform(-0.1, 0.1, size=(T, n, 12)), dtype=torch.float32, device="cuda:0")
    fused, stepped = _make(n, lanes_per_robot=lanes), _make(n, lanes_per_robot=lanes)
    for e in (fused, stepped):
        e.reset(ETG_w=W, ETG_b=B)
        e.set_external_force(f)
    st_end, obs_end, ret, ln, alive = _stepping_break_at_done(stepped, T, lambda e, k: tape[k])
    assert 0.2 * n < int((~alive).sum()) < 0.95 * n
    ret_f, ln_f, rec = fused.rollout_actions(tape, record=("obs", "reward", "done", "joint_angle"))
    _compare_with_stepping("action tape lanes %d" % lanes, fused.get_state(), fused.obs, ln_f, ret_f, st_end, obs_end, ln, ret)
    done, rew = rec["done"], rec["reward"]
    li = ln_f.long()
    idx = torch.arange(T, device="cuda:0")[:, None]
    after = idx >= li[None, :]                                  # steps after the robot's episode
    assert bool(done[after].all()) and bool((rew[after] == 0).all())
    before = idx < (li - 1)[None, :]
    assert not bool(done[before].any())
    fin = li < T
    last_rows = rec["obs"][(li - 1).clamp(min=0), torch.arange(n, device="cuda:0")]
    assert torch.equal(last_rows[fin], fused.obs[fin])          # the tape's row of the ending step == the robot's final row
    assert bool(done[(li - 1).clamp(min=0), torch.arange(n, device="cuda:0")][fin].all())
    # sum of the tape's rewards == the accumulator
    assert torch.allclose(rew.sum(0), ret_f, atol=1e-3)
    fused.close(); stepped.close()


def test_recorded_closed_loop_stops_at_done_and_feeds_the_replay_memory():
    _need_gpu()
    from paddlerobotics_amd.replay import DeviceReplayMemory, store_recorded
    n, T = 64, 50
    W, B = _etg_params(n, seed=11)
    pol, _ = _policy()
    env = _make(n)
    env.reset(ETG_w=W, ETG_b=B)
    env.set_external_force(_pushes(n))
    ret, ln, rec = env.rollout_policy_record(pol, T, 0.3)
    li = ln.long()
    idx = torch.arange(T, device="cuda:0")[:, None]
    after = idx >= li[None, :]
    assert 0.2 * n < int((li < T).sum()) < 0.95 * n
    assert bool(rec["done"][after].all()) and bool((rec["reward"][after] == 0).all())
    assert torch.allclose(rec["reward"].sum(0), ret, atol=1e-3)
    rpm = DeviceReplayMemory(T * n, A.OBS_DIM, 12, device="cuda:0")
    store_recorded(rpm, rec)
    assert rpm.size() == int(li.sum().item())                  # exactly the steps the episodes ran
    env.close()


@pytest.mark.parametrize("lanes", [16, 4])
def test_last_row_carries_the_noise_of_the_step_that_wrote_it(lanes):
    """open loop + sensor noise: stepping with break-at-done draws every row's noise at the stream position of its step; the fused
    rollout's rows (written at different steps: a robot's last one when its episode ends) must carry the same draws"""
    _need_gpu()
    n, steps = 128, 70
    W, B = _etg_params(64, seed=13)
    W, B = np.tile(W, (n // 64, 1, 1)), np.tile(B, (n // 64, 1))
    f = _pushes(n)
    noise = [0.02, 0.3, 0.0, 0.01, 0.05]
    fused, stepped = (_make(n, lanes_per_robot=lanes, observation_noise_stdev=noise, seed=21) for _ in range(2))
    for e in (fused, stepped):
        e.reset(ETG_w=W, ETG_b=B)
        e.set_external_force(f)
    st_end, obs_end, ret, ln, alive = _stepping_break_at_done(stepped, steps)
    ret_f, ln_f = fused.rollout_openloop(steps)
    same = (ln_f == ln)
    assert same.float().mean().item() > 0.9 and 0.2 * n < int((~alive).sum()) < 0.95 * n
    gap = (fused.obs - obs_end)[same].abs()
    # a wrong stream position shows as the noise itself: 0.02 rad x 10 on the angle columns, 0.3 on the velocity columns
    _say("sensor noise lanes %d: last-row gap to stepping, median %.1e q95 %.1e (a wrong draw would be ~0.2)" % (
        lanes, gap.max(1).values.median().item(), gap.max(1).values.quantile(0.95).item()))
    assert gap.max(1).values.median().item() < 5e-3 and gap.max(1).values.quantile(0.9).item() < 5e-2
    fused.close(); stepped.close()
