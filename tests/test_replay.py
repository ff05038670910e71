"""The transition hand-off of the step boundary (paddlerobotics_amd/replay.py): the reference's ReplayMemory call sites
(train.py:141,159,163-165,240-241,323-324) served from a ring of transitions on the device, and the batched
run_train_episode / run_EStrain_episode collection loop (train.py:129-179,213-249)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from paddlerobotics_amd.replay import DeviceReplayMemory, collect_transitions


def _batch(n, od, ad, base):
    obs = torch.arange(n * od, dtype=torch.float32).view(n, od) + base
    return obs, obs[:, :ad] * 0.5, obs[:, 0] * 2.0, obs + 0.25, (obs[:, 0] % 2 == 0).float()


def test_memory_append_mask_wrap_sample_and_files(tmp_path):
    od, ad = 5, 3
    m = DeviceReplayMemory(10, od, ad, device="cpu")
    assert m.size() == 0
    # single transition, the reference's call (1-D numpy rows, python scalars)
    m.append(np.arange(od, dtype=np.float64), np.ones(ad), 1.5, np.arange(od) + 1.0, 1.0)
    assert m.size() == 1 and m.obs[0].tolist() == [0, 1, 2, 3, 4] and m.reward[0] == 1.5 and m.terminal[0] == 1.0
    # masked batch: rows 0, 2, 3 of 4 are stored, in order, after the first transition
    o, a, r, no, t = _batch(4, od, ad, 100.0)
    m.append_batch(o, a, r, no, t, mask=torch.tensor([1, 0, 1, 1], dtype=torch.uint8))
    assert m.size() == 4
    assert torch.equal(m.obs[1:4], o[[0, 2, 3]]) and torch.equal(m.action[1:4], a[[0, 2, 3]])
    assert torch.equal(m.reward[1:4], r[[0, 2, 3]]) and torch.equal(m.next_obs[1:4], no[[0, 2, 3]])
    assert torch.equal(m.terminal[1:4], t[[0, 2, 3]])
    # wrap around: 8 more rows into a ring of 10 -> slots 4..9, 0, 1; size saturates
    o2, a2, r2, no2, t2 = _batch(8, od, ad, 1000.0)
    m.append_batch(o2, a2, r2, no2, t2)
    assert m.size() == 10 and int(m._pos) == 2
    assert torch.equal(m.obs[4:10], o2[:6]) and torch.equal(m.obs[0:2], o2[6:]) and torch.equal(m.obs[2:4], o[[2, 3]])
    # an all-false mask stores nothing
    before = m.obs.clone()
    m.append_batch(o, a, r, no, t, mask=torch.zeros(4, dtype=torch.bool))
    assert torch.equal(m.obs[:10], before[:10]) and int(m._pos) == 2
    # sampling: rows come from the stored set, consistent across the five arrays, roughly uniform
    g = torch.Generator().manual_seed(0)
    so, sa, sr, sno, st = m.sample_batch(4000, generator=g)
    assert so.shape == (4000, od) and sa.shape == (4000, ad) and sr.shape == (4000,) and st.shape == (4000,)
    assert torch.equal(sno, so + 0.25) and torch.equal(sr, so[:, 0] * 2.0)
    _, counts = torch.unique(so[:, 0], return_counts=True)
    assert counts.numel() == 10 and counts.min() > 300
    # a partly filled memory only yields what it holds
    p = DeviceReplayMemory(100, od, ad, device="cpu")
    p.append_batch(*_batch(3, od, ad, 7.0))
    assert set(p.sample_batch(200, generator=g)[0][:, 0].tolist()) <= {7.0, 12.0, 17.0}
    # files
    m.save(str(tmp_path / "rpm"))
    q = DeviceReplayMemory(10, od, ad, device="cpu")
    q.load(str(tmp_path / "rpm"))
    assert q.size() == 10 and int(q._pos) == 2 and torch.equal(q.obs[:10], m.obs[:10]) and torch.equal(q.terminal[:10], m.terminal[:10])
    with pytest.raises(ValueError):
        DeviceReplayMemory(5, od, ad, device="cpu").load(str(tmp_path / "rpm"))
    with pytest.raises(ValueError):
        m.append_batch(*_batch(11, od, ad, 0.0))
    with pytest.raises(ValueError):
        m.append_batch(torch.zeros(2, od + 1), a[:2], r[:2], no[:2], t[:2])


class _ToyEnv:
    """deterministic stand-in with the env surface collect_transitions uses: robot i terminates at step first_done + i + 1; obs is
    ONE buffer overwritten in place by every step (as the real env's is).  On a GPU device it also carries the `info_buf` the
    fused storage path reads (torso = 2 * reward, velx alternating 0.5 / 0.1)."""

    class _Space:
        shape = (2,)

    def __init__(self, n, first_done=0, device="cpu", obs_dim=3):
        self.num_envs, self.device, self.action_space = n, torch.device(device), self._Space()
        self.obs = torch.zeros(n, obs_dim, device=self.device)
        self.idx = torch.arange(n, device=self.device)
        self.t = 0
        self.first_done = first_done
        if self.device.type == "cuda":
            from paddlerobotics_amd import a1_model as A
            self.info_buf = torch.zeros(n, A.INFO_DIM, device=self.device)
            self._torso, self._velx = A.INFO_SLICES["torso"][0], A.INFO_SLICES["velx"][0]

    def reset(self, **kw):
        self.t = 0
        self.obs[:] = self.idx.to(torch.float32)[:, None]
        self.ret = torch.zeros(self.num_envs, device=self.device); self.len = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self.alive = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        return self.obs, {}

    def step(self, action, donef=False):
        self.t += 1
        self.obs += 100.0 + action.sum(1, keepdim=True)
        done = (self.idx + 1 + self.first_done <= self.t) | bool(donef)
        rew = torch.full((self.num_envs,), float(self.t), device=self.device)
        self.ret += self.alive * rew; self.len += self.alive.int(); self.alive &= ~done
        info = {"torso": rew * 2, "velx": torch.where(self.idx % 2 == 0, 0.5, 0.1)}
        if hasattr(self, "info_buf"):
            self.info_buf[:, self._torso] = info["torso"]
            self.info_buf[:, self._velx] = info["velx"]
        return self.obs, rew, done, info

    def episode_stats(self):
        return self.ret, self.len


def test_collect_transitions_stores_live_rows_with_the_bootstrap_mask():
    n, max_step = 6, 3
    env, rpm = _ToyEnv(n), DeviceReplayMemory(64, 3, 2, device="cpu")
    g = torch.Generator().manual_seed(1)
    ret, ln, infos = collect_transitions(env, rpm, max_step, mode="uniform", action_bound=0.0, generator=g)
    # robot i lives i + 1 steps, capped by the forced done of step max_step + 1
    want_len = torch.tensor([1, 2, 3, 4, 4, 4], dtype=torch.int32)
    assert torch.equal(ln, want_len) and rpm.size() == int(want_len.sum())
    assert torch.equal(ret, torch.tensor([1.0, 3, 6, 10, 10, 10]))
    assert torch.equal(infos["torso"], 2 * ret) and torch.equal(infos["success_rate"], torch.tensor([1.0, 0, 1, 0, 1, 0]))
    k = rpm.size()
    o, no, r, t, a = rpm.obs[:k], rpm.next_obs[:k], rpm.reward[:k], rpm.terminal[:k], rpm.action[:k]
    # every stored row is one step of a live robot: next = obs + 100 (action_bound 0), obs is the PRE-step observation
    assert torch.equal(no, o + 100.0) and (a.abs() <= 1).all() and a.abs().max() > 0.5
    robot = (o[:, 0] % 100).long()
    step = (o[:, 0] // 100).long() + 1
    for i in range(n):
        rows = step[robot == i]
        assert rows.tolist() == list(range(1, int(want_len[i]) + 1))
    # terminal is the bootstrap mask 1 - done: 0 exactly on each robot's last stored row (forced done included)
    assert torch.equal(t == 0, step == want_len[robot].long())
    assert torch.equal(r, step.float())
    with pytest.raises(ValueError):
        collect_transitions(env, rpm, max_step, mode="predict")


@pytest.mark.gpu
def test_device_collection_keeps_bootstrapping_after_step_2000():
    """GPU, the etg_replay_* kernels under collect_transitions, on a toy env whose robots end at control steps 1999, 2000 and 2001:
    the kernels store 1 - done; from episode step 2000 on the stored flag is put back to 1 (train.py:148), exactly as the host
    path does (test_bootstrap_mask_stays_on_from_episode_step_2000)."""
    from tests.test_gpu_parity import _need_gpu
    _need_gpu()
    n = 3
    env = _ToyEnv(n, first_done=1998, device="cuda:0")
    rpm = DeviceReplayMemory(8192, 3, 2)
    assert rpm.fused
    ret, ln, infos = collect_transitions(env, rpm, 2005, mode="uniform", action_bound=0.0, generator=torch.Generator(device="cuda:0").manual_seed(0),
                                         info_keys=("torso", "no_such_key"))
    assert ln.tolist() == [1999, 2000, 2001] and rpm.size() == 6000 and int(rpm.size_tensor()) == 6000
    assert set(infos) == {"torso", "success_rate"} and torch.equal(infos["torso"], 2 * ret)
    k = rpm.size()
    robot, t = (rpm.obs[:k, 0] % 100).long(), rpm.terminal[:k]
    assert [(t[robot == i] == 0).sum().item() for i in range(n)] == [1, 0, 0]
    assert (rpm.reward[:k] > 0).all()


@pytest.mark.gpu
def test_collect_transitions_on_the_device_matches_a_stepping_loop_and_the_fused_rollout():
    """GPU: the collection loop on the real env + MFMA policy.  (1) `predict` mode: the stored rows are exactly the
    alive-masked trajectory of an identical env stepped by hand (step-major, robot-minor order), terminal = 1 - done, and
    the episode returns equal the fused closed-loop kernel's (etg_rollout_policy).  (2) `sample` mode (SAC.sample,
    alg/sac.py:65-76): stored actions are the squashed Gaussian draws, reproducible from the generator."""
    from tests.test_gpu_parity import _need_gpu, _make
    from tests.test_gpu_parity2 import _policy
    _need_gpu()
    n, max_step, bound = 256, 59, 0.3
    pol, _ = _policy()
    env, ref = _make(n, seed=5, body_contacts=0), _make(n, seed=5, body_contacts=0)   # (toe spheres only: two kernels of one source, equal to rounding; a gripping knee sphere amplifies it)
    rpm = DeviceReplayMemory(n * (max_step + 1) + 7, 49, 12)
    ret, ln, infos = collect_transitions(env, rpm, max_step, policy=pol, action_bound=bound)
    # the same episode by hand
    obs, _ = ref.reset()
    alive = torch.ones(n, dtype=torch.bool, device="cuda:0")
    rows = {k: [] for k in ("obs", "action", "reward", "next_obs", "terminal")}
    torso = torch.zeros(n, device="cuda:0")
    for steps in range(1, max_step + 2):
        a = pol.predict(obs)
        o0 = obs.clone()
        obs, r, d, info = ref.step(a * bound, donef=(steps > max_step))
        for k, v in (("obs", o0), ("action", a), ("reward", r), ("next_obs", obs), ("terminal", 1.0 - d.float())):
            rows[k].append(v[alive].clone())
        torso += alive.float() * info["torso"]
        alive &= ~d
    k = rpm.size()
    assert k == int(ln.sum().item()) and 0 < k < n * (max_step + 1) + 7
    for name in rows:
        want = torch.cat(rows[name])
        assert want.shape[0] == k and torch.equal(getattr(rpm, name)[:k], want), name
    assert torch.equal(infos["torso"], torso) and ((infos["success_rate"] >= 0) & (infos["success_rate"] <= 1)).all()
    assert int((rpm.terminal[:k] == 0).sum().item()) == n             # one closing row per robot
    ref.reset()
    ret_f, ln_f = ref.rollout_policy(pol, max_step + 1, bound)
    assert torch.equal(ln_f, ln) and (ret_f - ret).abs().max().item() < 2e-3 * max(1.0, ret.abs().max().item())
    # sampling from the memory on the device
    so, sa, sr, sno, st = rpm.sample_batch(512)
    assert so.is_cuda and so.shape == (512, 49) and sa.abs().max().item() <= 1.0 and torch.isfinite(sno).all()
    # (2) stochastic actions
    g = torch.Generator(device="cuda:0"); g.manual_seed(9)
    rpm2 = DeviceReplayMemory(n * 12, 49, 12)
    collect_transitions(env, rpm2, 9, policy=pol, action_bound=bound, mode="sample", generator=g)
    g.manual_seed(9)
    first = pol.sample(env.reset()[0], 1.0, generator=g, return_logp=False)
    assert torch.equal(rpm2.action[:n], first) and (rpm2.action[:n] - pol.predict(env.obs)).abs().max().item() > 1e-2
    # (3) warm-up collection with uniform actions (train.py:141-142)
    rpm3 = DeviceReplayMemory(n * 12, 49, 12)
    collect_transitions(env, rpm3, 9, mode="uniform")
    a3 = rpm3.action[:rpm3.size()]
    assert a3.min().item() < -0.9 and a3.max().item() > 0.9 and abs(a3.mean().item()) < 0.05
    env.close(); ref.close()


def _np_memory(max_size, od, ad):
    z = lambda *s: np.zeros(s, dtype=np.float32)
    return dict(obs=z(max_size + 1, od), next_obs=z(max_size + 1, od), action=z(max_size + 1, ad), reward=z(max_size + 1),
                terminal=z(max_size + 1), pc=np.zeros(2, dtype=np.int64))


def test_cpu_abi_statement_of_the_replay_entry_points_matches_the_torch_definition():
    """oracle/libetgsim_cpu.so restates etg_replay_begin / etg_replay_end sequentially on host pointers; DeviceReplayMemory's
    torch indexing (the definition used on host tensors) gives the same memory for random masked batches with wrap-around,
    and the info sums / success counter / alive update follow train.py:150-156."""
    import ctypes as C
    from tests.test_abi_and_emu import _cpu_abi
    lib = _cpu_abi()
    ll = C.c_longlong
    vp = C.c_void_p
    lib.etg_replay_begin.argtypes = [vp, C.c_int, ll, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_float, vp, vp]
    lib.etg_replay_end.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    p = lambda a: a.ctypes.data_as(vp)
    rng = np.random.default_rng(0)
    n, od, ad, cap, idim = 37, 5, 3, 100, 9
    ref = DeviceReplayMemory(cap, od, ad, device="cpu")
    mem = _np_memory(cap, od, ad)
    alive = np.ones(n, dtype=np.uint8)
    sums, want_sums = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32)
    for step in range(9):
        obs, act = rng.normal(size=(n, od)).astype(np.float32), rng.normal(size=(n, ad)).astype(np.float32)
        nxt, rew = rng.normal(size=(n, od)).astype(np.float32), rng.normal(size=n).astype(np.float32)
        done = (rng.random(n) < 0.15).astype(np.uint8)
        info = rng.normal(size=(n, idim)).astype(np.float32)
        slot = np.zeros(n, dtype=np.int32)
        before = alive.copy()
        scaled = np.zeros((n, ad), np.float32)
        assert lib.etg_replay_begin(p(alive), n, cap, p(mem["pc"]), p(slot), p(obs), od, p(act), ad, p(mem["obs"]), p(mem["action"]), 0.3, p(scaled), None) == 0
        assert np.array_equal(scaled, np.float32(0.3) * act)
        assert lib.etg_replay_end(p(slot), n, p(rew), p(done), p(nxt), od, p(mem["reward"]), p(mem["terminal"]), p(mem["next_obs"]),
                                  p(info), idim, 7, 8, p(sums), p(alive), None) == 0
        ref.append_batch(torch.from_numpy(obs), torch.from_numpy(act), torch.from_numpy(rew), torch.from_numpy(nxt),
                         torch.from_numpy(1.0 - done.astype(np.float32)), mask=torch.from_numpy(before))
        want_sums[:, :7] += before[:, None] * info[:, :7]
        want_sums[:, 7] += before * (info[:, 8] >= 0.3)
        assert np.array_equal(alive, before & (1 - done))
    for name in ("obs", "next_obs", "action", "reward", "terminal"):
        assert np.array_equal(mem[name][:cap], getattr(ref, name)[:cap].numpy()), name
    assert mem["pc"][0] == int(ref._pos) and mem["pc"][1] == int(ref._count) and mem["pc"][1] > cap   # wrapped
    assert np.allclose(sums, want_sums, atol=1e-5)
    assert lib.etg_replay_begin(p(alive), cap + 1, cap, p(mem["pc"]), p(slot), p(obs), od, p(act), ad, p(mem["obs"]), p(mem["action"]), 1.0, None, None) == -1


@pytest.mark.gpu
def test_fused_replay_kernels_match_the_torch_definition():
    """GPU: etg_replay_begin / etg_replay_end (prefix sum + scattered rows) against the torch indexing on the same device:
    random masked batches of 4096 and of a ragged size, wrap-around, all-dead and all-alive masks."""
    from tests.test_gpu_parity import _need_gpu
    _need_gpu()
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    # > 8192 rows: the three-launch slot computation; > 65536 rows: the bounded grid that skips dead rows 64 at a time
    for n, cap in ((4096, 10000), (1000, 2999), (1, 3), (20000, 50000), (8193, 8200), (70001, 150000)):
        a, b = DeviceReplayMemory(cap, 49, 12), DeviceReplayMemory(cap, 49, 12, fused=False)
        assert a.fused and not b.fused
        for step in range(7):
            r = lambda *s: torch.randn(*s, device="cuda:0", generator=g)
            obs, act, rew, nxt = r(n, 49), r(n, 12), r(n), r(n, 49)
            term = (torch.rand(n, device="cuda:0", generator=g) < 0.8).float()
            mask = None if step == 0 else (torch.zeros(n, dtype=torch.bool, device="cuda:0") if step == 1 else
                                           torch.rand(n, device="cuda:0", generator=g) < 0.6)
            for m in (a, b):
                m.append_batch(obs, act, rew, nxt, term, mask=mask)
        for name in ("obs", "next_obs", "action", "reward", "terminal"):
            assert torch.equal(getattr(a, name)[:cap], getattr(b, name)[:cap]), (n, name)
        assert int(a._pos) == int(b._pos) and int(a._count) == int(b._count) and a.size() == b.size()
    # rows wider than a wave (history-stacked observations: 3 x 49 columns): the element-per-thread form of the copies
    a, b = DeviceReplayMemory(5000, 147, 12), DeviceReplayMemory(5000, 147, 12, fused=False)
    for step in range(4):
        r = lambda *s: torch.randn(*s, device="cuda:0", generator=g)
        obs, act, rew, nxt = r(1500, 147), r(1500, 12), r(1500), r(1500, 147)
        term = (torch.rand(1500, device="cuda:0", generator=g) < 0.8).float()
        mask = torch.rand(1500, device="cuda:0", generator=g) < 0.7
        for m in (a, b):
            m.append_batch(obs, act, rew, nxt, term, mask=mask)
    for name in ("obs", "next_obs", "action", "reward", "terminal"):
        assert torch.equal(getattr(a, name)[:5000], getattr(b, name)[:5000]), name
    assert int(a._pos) == int(b._pos) and int(a._count) == int(b._count)
    with pytest.raises(Exception):
        DeviceReplayMemory(10, 49, 12).append_batch(torch.zeros(11, 49, device="cuda:0"), torch.zeros(11, 12, device="cuda:0"),
                                                   torch.zeros(11, device="cuda:0"), torch.zeros(11, 49, device="cuda:0"),
                                                   torch.ones(11, device="cuda:0"))


def test_bc_pair_memory_and_noise_levels():
    """BCtrain.py: rpm.append(agent_obs, ref_obs) / sample_batch_by_index on the pair memory, and obs2noise's per-column
    levels (BCtrain.py:53-59)."""
    from paddlerobotics_amd.replay import obs2noise
    m = DeviceReplayMemory(20, 4, 7, device="cpu")
    a, b = torch.arange(12.0).view(3, 4), torch.arange(21.0).view(3, 7) + 100
    m.append_pairs(a, b, mask=torch.tensor([True, False, True]))
    m.append_pairs(a + 50, b + 50)
    assert m.size() == 5
    o, r = m.sample_batch_by_index([0, 1, 4])
    assert torch.equal(o, torch.stack([a[0], a[2], a[2] + 50])) and torch.equal(r, torch.stack([b[0], b[2], b[2] + 50]))
    g = torch.Generator().manual_seed(0)
    obs = torch.zeros(20000, 49)
    d = obs2noise(obs, g)
    assert d[:, :7].abs().max() == 0 and d[:, 37:].abs().max() == 0 and obs.abs().max() == 0
    for (lo, hi, sd) in ((7, 10, 0.6), (10, 13, 0.2), (13, 25, 0.1), (25, 37, 0.5)):
        assert abs(d[:, lo:hi].std().item() - sd) < 0.03 * sd and abs(d[:, lo:hi].mean().item()) < 0.02 * sd


@pytest.mark.gpu
def test_collect_bc_pairs_on_the_device():
    """GPU: the behaviour-cloning collection loop (BCtrain.py:87-131) with a 46-input student: stored pairs are (noisy
    observation without the displacement columns, clean teacher observation) of live robots only."""
    from tests.test_gpu_parity import _need_gpu, _make
    from paddlerobotics_amd.policy import MfmaPolicy
    from paddlerobotics_amd.replay import collect_bc_pairs
    _need_gpu()
    n, max_step = 128, 29
    student = MfmaPolicy(46, 12)
    student.load_state_dict(MfmaPolicy.init_like_reference(46, 12, seed=1))
    env = _make(n, seed=2)
    rpm = DeviceReplayMemory(n * (max_step + 1), 46, 49)
    g = torch.Generator(device="cuda:0"); g.manual_seed(4)
    ret, ln = collect_bc_pairs(env, rpm, max_step, student=student, generator=g)
    k = rpm.size()
    assert k == int(ln.sum().item()) and torch.isfinite(ret).all()
    stu, tea = rpm.sample_batch_by_index(torch.arange(k))
    assert stu.shape == (k, 46) and tea.shape == (k, 49)
    diff = stu - tea[:, 3:]
    assert diff[:, :4].abs().max().item() == 0 and diff[:, 34:].abs().max().item() == 0      # contacts and ETG columns: clean
    assert abs(diff[:, 4:7].std().item() - 0.6) < 0.06 and abs(diff[:, 22:34].std().item() - 0.5) < 0.05
    # without the noise the student sees the teacher's row minus the displacement
    rpm2 = DeviceReplayMemory(n * (max_step + 1), 46, 49)
    collect_bc_pairs(env, rpm2, max_step, mode="uniform", sensor_noise=False)
    s2, t2 = rpm2.sample_batch_by_index(torch.arange(rpm2.size()))
    assert torch.equal(s2, t2[:, 3:])
    with pytest.raises(ValueError):
        collect_bc_pairs(env, DeviceReplayMemory(100, 49, 12), max_step, mode="uniform")
    env.close()


def test_bootstrap_mask_stays_on_from_episode_step_2000():
    """train.py:148-149: `terminal = float(done) if episode_steps < 2000 else 0; terminal = 1 - terminal` -- a robot that ends at
    control step 2000 or later is stored with the bootstrap mask still 1, on the stepping path and for recorded episodes."""
    from paddlerobotics_amd.replay import bootstrap_mask, store_recorded, BOOTSTRAP_ALWAYS_FROM
    d = torch.tensor([True, False, True])
    assert torch.equal(bootstrap_mask(d, 1999), torch.tensor([0.0, 1.0, 0.0]))
    assert torch.equal(bootstrap_mask(d, 2000), torch.ones(3)) and BOOTSTRAP_ALWAYS_FROM == 2000
    n = 3
    env, rpm = _ToyEnv(n, first_done=1998), DeviceReplayMemory(8192, 3, 2, device="cpu")   # ends at steps 1999, 2000, 2001
    ret, ln, infos = collect_transitions(env, rpm, 2005, mode="uniform", action_bound=0.0, generator=torch.Generator().manual_seed(0),
                                         info_keys=("torso", "no_such_key", "base_position"))   # unknown / vector keys: skipped
    assert ln.tolist() == [1999, 2000, 2001] and rpm.size() == 6000 and int(rpm.size_tensor()) == 6000
    assert set(infos) == {"torso", "success_rate"}
    k = rpm.size()
    robot, t = (rpm.obs[:k, 0] % 100).long(), rpm.terminal[:k]
    assert (t[robot == 0] == 0).sum() == 1 and (t[robot == 1] == 0).sum() == 0 and (t[robot == 2] == 0).sum() == 0
    # the recorded-episode path: the same rule per row index (episode step = row + 1)
    T, od, ad = 2002, 3, 2
    done = torch.zeros(T, n, dtype=torch.bool)
    done[1998, 0] = done[1999, 1] = done[2000, 2] = True
    rec = {"obs": torch.zeros(T, n, od), "action": torch.zeros(T, n, ad), "reward": torch.zeros(T, n), "final_obs": torch.zeros(n, od), "done": done}
    rec["obs"][..., 0] = torch.arange(n, dtype=torch.float32)
    m = DeviceReplayMemory(8192, od, ad, device="cpu")
    store_recorded(m, rec)
    k = m.size()
    assert k == 6000
    robot, t = m.obs[:k, 0].long(), m.terminal[:k]
    assert [(t[robot == i] == 0).sum().item() for i in range(n)] == [1, 0, 0]


def test_store_recorded_masks_rows_after_the_first_done():
    """replay.store_recorded on host tensors: [T, N] recordings -> the rows up to each robot's first done, step-major,
    next_obs = the following step's observation (the final one for the last step), terminal = 1 - done."""
    from paddlerobotics_amd.replay import store_recorded
    T, N, od, ad = 5, 3, 4, 2
    obs = torch.arange(T * N * od, dtype=torch.float32).view(T, N, od)
    rec = {"obs": obs, "action": obs[..., :ad] * 0.1, "reward": obs[..., 0] * 2, "final_obs": torch.full((N, od), -1.0),
           "done": torch.tensor([[0, 0, 0], [0, 1, 0], [0, 0, 0], [1, 1, 0], [0, 0, 1]], dtype=torch.bool)}
    m = DeviceReplayMemory(100, od, ad, device="cpu")
    assert store_recorded(m, rec) == T * N
    # robot 0 lives steps 0..3, robot 1 steps 0..1, robot 2 steps 0..4
    want = [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 2), (3, 0), (3, 2), (4, 2)]
    assert m.size() == len(want)
    for k, (t, i) in enumerate(want):
        assert torch.equal(m.obs[k], obs[t, i]) and torch.equal(m.action[k], obs[t, i, :ad] * 0.1) and m.reward[k] == obs[t, i, 0] * 2
        assert torch.equal(m.next_obs[k], obs[t + 1, i] if t + 1 < T else rec["final_obs"][i])
        assert m.terminal[k] == (0.0 if rec["done"][t, i] else 1.0)
    # a memory smaller than the episode: appended in slices of whole steps, the ring keeps the newest rows
    small = DeviceReplayMemory(7, od, ad, device="cpu")
    store_recorded(small, rec)
    assert small.size() == 7 and int(small._count) == len(want)
    newest = want[-7:]
    stored = {tuple(small.obs[k].tolist()) for k in range(7)}
    assert stored == {tuple(obs[t, i].tolist()) for t, i in newest}


@pytest.mark.gpu
def test_recorded_fused_rollout_fills_the_memory_like_the_stepping_loop():
    """GPU: etg_rollout_policy_record (the fused closed-loop kernel writing every step's observation / action / reward / done)
    + store_recorded == collect_transitions(mode="predict") stepping the same robots: same number of rows in the same order,
    observations / actions / rewards equal to the two kernels' rounding, identical terminal flags, identical episode lengths."""
    from tests.test_gpu_parity import _need_gpu, _make
    from tests.test_gpu_parity2 import _policy
    from paddlerobotics_amd.replay import collect_recorded
    _need_gpu()
    n, max_step, bound = 256, 39, 0.3
    pol, _ = _policy()
    a, b = _make(n, seed=5, body_contacts=0), _make(n, seed=5, body_contacts=0)   # (toe spheres only: see above)
    ra, rb = DeviceReplayMemory(n * (max_step + 1), 49, 12), DeviceReplayMemory(n * (max_step + 1), 49, 12)
    ret_a, ln_a, _ = collect_transitions(a, ra, max_step, policy=pol, action_bound=bound)
    ret_b, ln_b = collect_recorded(b, rb, max_step, pol, action_bound=bound)
    assert torch.equal(ln_a, ln_b) and ra.size() == rb.size() == int(ln_a.sum().item())
    k = ra.size()
    assert torch.equal(ra.terminal[:k], rb.terminal[:k])
    # two kernels of the same source (FMA contraction differs) in closed loop through an actor with gains of ~10: rows agree to
    # rounding level, except downstream of the few contact / joint-stop events that the last bit decides (a calf joint resting
    # on its stop: tests/test_gpu_parity4.py) -- the typical row is held tight, 2 % of the rows may sit further out
    def close(x, y, tol):
        d = (x - y).abs() / (1 + y.abs())
        # (the rows allowed outside `tol` -- robots at a contact bifurcation -- are capped too: finite, and off by less than their own size)
        return bool(torch.isfinite(d).all()) and bool(d.median() <= 0.1 * tol) and float((d > tol).float().mean()) < 0.02 and float(d.max()) <= 1.0
    assert close(ra.obs[:k], rb.obs[:k], 2e-3) and close(ra.next_obs[:k], rb.next_obs[:k], 2e-3)
    assert close(ra.action[:k], rb.action[:k], 1e-4) and close(ra.reward[:k], rb.reward[:k], 1e-3)
    assert close(ret_a, ret_b, 2e-3)
    # the first step's rows are the reset observation and the actor's answer to it, bit for bit
    assert torch.equal(rb.obs[:n], ra.obs[:n]) and (rb.action[:n] - pol.predict(rb.obs[:n].contiguous())).abs().max().item() < 1e-6
    with pytest.raises(ValueError):
        _make(32, lanes_per_robot=4).rollout_policy_record(pol, 5)
    a.close(); b.close()


@pytest.mark.gpu
def test_es_generation_with_an_actor_keeps_the_candidates_episodes():
    """One ES generation over ETG control points with a fixed actor and a replay memory (train.py:398-418 with --es_rpm):
    fitness = the fused closed-loop returns, and every live step of every candidate lands in the memory."""
    from tests.test_gpu_parity import _need_gpu, _make
    from tests.test_gpu_parity2 import _policy
    from paddlerobotics_amd import rollout as R
    from paddlerobotics_amd.es import SimpleGA
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    _need_gpu()
    n, max_step = 256, 49
    pol, _ = _policy()
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    solver = SimpleGA(12, sigma_init=0.02, popsize=n, param=np.zeros(12), device="cuda:0")
    env, twin = _make(n), _make(n)
    rpm = DeviceReplayMemory(n * (max_step + 1), 49, 12)
    fit = R.es_generation(solver, R.make_etg_evaluator(env, layer, 0.5, prior, w0, b0, max_step=max_step, policy=pol, rpm=rpm))
    solver2 = SimpleGA(12, sigma_init=0.02, popsize=n, param=np.zeros(12), device="cuda:0")
    fit2 = R.es_generation(solver2, R.make_etg_evaluator(twin, layer, 0.5, prior, w0, b0, max_step=max_step, policy=pol))
    assert torch.allclose(fit, fit2, rtol=1e-4, atol=1e-3)                      # same candidates, same fused kernel arithmetic
    _, ln = env.episode_stats()
    assert rpm.size() == int(ln.sum().item()) and int((rpm.terminal[:rpm.size()] == 0).sum().item()) == n
    env.close(); twin.close()


@pytest.mark.gpu
def test_recorded_stochastic_rollout_matches_the_sampling_loop():
    """GPU: etg_rollout_policy_record with the caller's N(0,1) draws = the squashed-Gaussian actor of run_train_episode
    (agent.sample, alg/sac.py:65-76) fused into the closed-loop kernel; the same draws fed to the stepping loop
    (policy.sample + env.step) fill the memory with the same rows."""
    from tests.test_gpu_parity import _need_gpu, _make
    from tests.test_gpu_parity2 import _policy
    from paddlerobotics_amd.replay import collect_recorded
    _need_gpu()
    n, max_step, bound = 128, 29, 0.3
    pol, _ = _policy()
    g = torch.Generator(device="cuda:0"); g.manual_seed(12)
    noise = torch.randn(max_step + 1, n, 12, device="cuda:0", generator=g)
    a, b = _make(n, seed=6, body_contacts=0), _make(n, seed=6, body_contacts=0)   # (toe spheres only: see above)
    ra, rb = DeviceReplayMemory(n * (max_step + 1), 49, 12), DeviceReplayMemory(n * (max_step + 1), 49, 12)
    ret_a, ln_a, _ = collect_transitions(a, ra, max_step, policy=pol, action_bound=bound, mode="sample", noise=noise)
    ret_b, ln_b = collect_recorded(b, rb, max_step, pol, action_bound=bound, mode="sample", noise=noise)
    assert torch.equal(ln_a, ln_b) and ra.size() == rb.size()
    k = ra.size()
    close = lambda x, y, tol: bool(((x - y).abs() <= tol * (1 + y.abs())).all())
    assert torch.equal(ra.terminal[:k], rb.terminal[:k])
    assert close(ra.action[:k], rb.action[:k], 2e-4) and close(ra.obs[:k], rb.obs[:k], 2e-3) and close(ra.reward[:k], rb.reward[:k], 2e-3)
    # the stochastic actions are not the deterministic ones, and they are what policy.sample gives for the first rows
    det = pol.predict(rb.obs[:n].contiguous())
    assert (rb.action[:n] - det).abs().max().item() > 1e-2
    first = pol.sample(rb.obs[:n].contiguous(), 1.0, noise=noise[0], return_logp=False)
    assert (rb.action[:n] - first).abs().max().item() < 1e-5
    # drawn inside collect_recorded when no noise is passed
    rc = DeviceReplayMemory(n * (max_step + 1), 49, 12)
    collect_recorded(b, rc, max_step, pol, action_bound=bound, mode="sample", generator=g)
    assert rc.size() > 0 and (rc.action[:n] - det).abs().max().item() > 1e-2
    a.close(); b.close()
