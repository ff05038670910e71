"""-m gpu: the batch as G sub-batches on G streams (etg_step_range; env.step(groups=G), rollout_policy(fused=False, groups=G)).

train.py:129-178's loop has one barrier per control step: the slowest wavefront of the whole batch.  Sub-batches are
independent, so group g's step k + 1 may start when ITS slowest wavefront has finished.  Checked here: whatever G, every robot's
observations, rewards, done flags, info rows, state and episode statistics are BIT-identical to G = 1 -- both lane mappings,
a batch size that leaves a ragged last group, sensor noise on (every robot draws from its own counter-based stream)."""
import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_gpu_parity import _need_gpu, _etg_params, _make   # noqa: E402

NOISE = [0.02, 0.3, 0.0, 0.01, 0.05]
STEPS = 25


@pytest.mark.parametrize("lanes", [16, 4])
@pytest.mark.parametrize("n,noise", [(256, False), (112, True)])
def test_step_in_groups_is_step(lanes, n, noise):
    _need_gpu()
    W, B = _etg_params(n, seed=31)
    kw = dict(observation_noise_stdev=NOISE) if noise else {}
    rng = np.random.default_rng(3)
    acts = rng.uniform(-0.15, 0.15, size=(STEPS, n, 12)).astype(np.float32)
    f = np.zeros((n, 3), np.float32); f[:, 1] = np.linspace(0, 90, n)      # some robots fall: done flags, body rows
    res = {}
    for G in (1, 2, 4, 7):
        env = _make(n, lanes_per_robot=lanes, seed=5, **kw)
        env.reset(ETG_w=W, ETG_b=B)
        env.set_external_force(torch.as_tensor(f))
        rg = env.group_ranges(G)
        assert rg[0][0] == 0 and rg[-1][1] == n and all(a % 16 == 0 for a, _ in rg) and all(x[1] == y[0] for x, y in zip(rg, rg[1:]))
        rows = []
        for k in range(STEPS):
            obs, r, d, info = env.step(torch.as_tensor(acts[k]), groups=G)
            rows.append((obs.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy(), env.info_buf.cpu().numpy().copy()))
        ret, ln = env.episode_stats()
        res[G] = (rows, env.get_state().cpu().numpy().copy(), ret.cpu().numpy().copy(), ln.cpu().numpy().copy())
        env.close()
    assert res[1][1].shape == (n, A.STATE_DIM) and (res[1][3] < STEPS).any()      # someone's episode ended on the way
    for G in (2, 4, 7):
        for k in range(STEPS):
            for x, y in zip(res[1][0][k], res[G][0][k]):
                assert np.array_equal(x, y), (G, k)
        for x, y in zip(res[1][1:], res[G][1:]):
            assert np.array_equal(x, y), G


@pytest.mark.parametrize("student", [False, True])
def test_grouped_stepping_closed_loop_is_the_stepping_closed_loop(student):
    """predict() + step() per control step, per sub-batch on its own stream, against the one-batch loop: bit-identical"""
    _need_gpu()
    from paddlerobotics_amd.policy import MfmaPolicy
    n = 512
    W, B = _etg_params(n, seed=33)
    kw = dict(sensor_mode={"dis": 0}) if student else {}      # the 46-float observation of BCtrain.py:53-59 (columns 3..48)
    res = {}
    for G in (1, 4, 8):
        env = _make(n, seed=2, **kw)
        obs_dim = env.observation_space.shape[0]
        pol = MfmaPolicy(obs_dim, 12, device="cuda:0")
        pol.load_state_dict(MfmaPolicy.init_like_reference(obs_dim, 12, seed=4))
        env.reset(ETG_w=W, ETG_b=B)
        ret, ln = env.rollout_policy(pol, 25, 0.3, fused=False, groups=G)
        torch.cuda.synchronize()
        res[G] = (env.get_state().cpu().numpy().copy(), ret.cpu().numpy().copy(), ln.cpu().numpy().copy(), env.obs.cpu().numpy().copy())
        env.close()
    for G in (4, 8):
        for x, y in zip(res[1], res[G]):
            assert np.array_equal(x, y), G


def test_groups_refuse_what_they_do_not_cover():
    _need_gpu()
    env = _make(64, auto_reset=True)
    env.reset()
    with pytest.raises(ValueError):
        env.step(None, groups=2)
    env.close()
