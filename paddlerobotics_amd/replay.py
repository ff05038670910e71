"""The transition side of the step boundary: what the reference's episode loops hand to its off-policy learner.

  ReplayMemory(max_size, obs_dim, act_dim)          train.py:323-324 (parl.utils.ReplayMemory; parl is not vendored in
  rpm.append(obs, action, reward, next_obs, terminal)  the reference tree: its interface is taken from these call sites)
  rpm.size(), rpm.sample_batch(BATCH_SIZE)          train.py:141,159,163-165,240-241
  terminal = 1 - float(done), 1 from episode step 2000 on   train.py:148-149,229-230 (the stored flag is the BOOTSTRAP mask;
                                                    bootstrap_mask() below)

One env here is thousands of robots, so one `step()` yields a BATCH of transitions: `append_batch` writes the rows of the
robots whose episode is still running into a ring of transitions that lives in HBM next to the simulator -- no host copy,
no host synchronisation (the write position is a device scalar; rows of finished robots go to a scratch slot).
`collect_transitions` is the batched run_train_episode / run_EStrain_episode collection loop.

On a HIP device the appends go through etg_replay_begin / etg_replay_end (csrc/etg_replay.hip: prefix sum over the alive
bytes + scattered rows, three launches per control step, the info sums and the alive update folded in); the torch
indexing below is the definition of the same operation (host tensors, tests) and what the kernels are tested against.
"""
import ctypes as C

import numpy as np
import torch


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DeviceReplayMemory:
    def __init__(self, max_size, obs_dim, act_dim, device="cuda:0", fused=None):
        self.max_size, self.obs_dim, self.act_dim = int(max_size), int(obs_dim), int(act_dim)
        self.device = torch.device(device)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        # row max_size of every array is the scratch slot masked-out rows are written to
        self.obs, self.next_obs = z(self.max_size + 1, self.obs_dim), z(self.max_size + 1, self.obs_dim)
        self.action = z(self.max_size + 1, self.act_dim)
        self.reward, self.terminal = z(self.max_size + 1), z(self.max_size + 1)
        self._pc = torch.zeros(2, dtype=torch.int64, device=self.device)       # [next slot to write, transitions ever appended]
        # HIP kernels (etg_replay.hip) on a GPU; fused=False keeps the torch definition there too (what the tests compare with)
        self.fused = (self.device.type == "cuda") if fused is None else bool(fused)
        if self.fused:
            from . import _lib
            self._lib, self._check = _lib.load(), _lib.check

    @property
    def _pos(self):
        return self._pc[0]

    @_pos.setter
    def _pos(self, v):
        self._pc[0] = v

    @property
    def _count(self):
        return self._pc[1]

    @_count.setter
    def _count(self, v):
        self._pc[1] = v

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- fused path (HIP device): alive = uint8 [n] device tensor (or None); returns the int32 slot tensor
    def begin(self, obs, action, alive=None, act_scale=1.0, act_scaled=None):
        obs, action = self._rows(obs, self.obs_dim).contiguous(), self._rows(action, self.act_dim).contiguous()
        n = obs.shape[0]
        if action.shape[0] != n or (alive is not None and (alive.numel() != n or alive.dtype != torch.uint8)):
            raise ValueError("begin(): obs / action / alive (uint8) must have one row per robot")
        slot = torch.empty(n, dtype=torch.int32, device=self.device)
        self._check(self._lib.etg_replay_begin(_ptr(alive), n, self.max_size, _ptr(self._pc), _ptr(slot), _ptr(obs), self.obs_dim,
                                               _ptr(action), self.act_dim, _ptr(self.obs), _ptr(self.action), C.c_float(act_scale),
                                               _ptr(act_scaled), self._stream()))
        return slot

    def end(self, slot, reward, done, next_obs, info_buf=None, n_sum=0, velx_col=-1, info_sum=None, alive=None):
        next_obs = self._rows(next_obs, self.obs_dim).contiguous()
        n = next_obs.shape[0]
        reward = reward.to(torch.float32).contiguous().view(-1)
        done = done.view(torch.uint8) if done.dtype == torch.bool else done.to(torch.uint8)
        done = done.contiguous().view(-1)
        if reward.numel() != n or done.numel() != n or slot.numel() != n:
            raise ValueError("end(): one reward / done / slot per robot")
        info_dim = 0 if info_buf is None else int(info_buf.shape[1])
        self._check(self._lib.etg_replay_end(_ptr(slot), n, _ptr(reward), _ptr(done), _ptr(next_obs), self.obs_dim, _ptr(self.reward),
                                             _ptr(self.terminal), _ptr(self.next_obs), _ptr(info_buf), info_dim, int(n_sum), int(velx_col),
                                             _ptr(info_sum), _ptr(alive), self._stream()))

    # ---- writing
    def slots(self, n, mask=None):
        """ring slots for the next batch of `n` rows (mask [n] bool/uint8: only those rows are stored) and advance the
        write position; all on the device."""
        if n > self.max_size:
            raise ValueError("a batch of %d transitions does not fit a memory of %d" % (n, self.max_size))
        if mask is None:
            k = torch.arange(n, device=self.device)
            slot = (self._pos + k) % self.max_size
            took = n
        else:
            m = mask.to(self.device).view(-1).to(torch.int64)
            if m.numel() != n:
                raise ValueError("mask has %d entries for a batch of %d" % (m.numel(), n))
            k = torch.cumsum(m, 0) - 1
            slot = torch.where(m > 0, (self._pos + k) % self.max_size, torch.full_like(k, self.max_size))
            took = m.sum()
        self._pos = (self._pos + took) % self.max_size
        self._count = self._count + took
        return slot

    def write_before(self, slot, obs, action):
        self.obs.index_copy_(0, slot, self._rows(obs, self.obs_dim))
        self.action.index_copy_(0, slot, self._rows(action, self.act_dim))

    def write_after(self, slot, reward, next_obs, terminal):
        self.reward.index_copy_(0, slot, self._vec(reward))
        self.next_obs.index_copy_(0, slot, self._rows(next_obs, self.obs_dim))
        self.terminal.index_copy_(0, slot, self._vec(terminal))

    def append_batch(self, obs, action, reward, next_obs, terminal, mask=None):
        if self.fused:   # terminal is the bootstrap mask 1 - done
            alive = None if mask is None else (mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)).contiguous().view(-1)
            slot = self.begin(obs, action, alive)
            done = (self._vec(terminal) == 0).view(torch.uint8)
            self.end(slot, self._vec(reward), done, next_obs)
            return
        slot = self.slots(self._rows(obs, self.obs_dim).shape[0], mask)
        self.write_before(slot, obs, action)
        self.write_after(slot, reward, next_obs, terminal)

    def append(self, obs, action, reward, next_obs, terminal):
        """one transition (the reference's call) or a batch of them"""
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self.device)
        if obs.dim() == 1:
            r = lambda x: torch.as_tensor(x, dtype=torch.float32, device=self.device).reshape(1, -1)
            self.append_batch(r(obs), r(action), r(reward).view(1), r(next_obs), r(terminal).view(1))
        else:
            self.append_batch(obs, action, reward, next_obs, terminal)

    # ---- reading
    def size(self):
        """number of stored transitions as a Python int (the reference's rpm.size()): a host synchronisation; hot loops use
        size_tensor()"""
        return int(min(int(self._count.item()), self.max_size))

    def size_tensor(self):
        """the same count as a 0-d tensor on the memory's device: no host synchronisation (e.g.
        torch.where(rpm.size_tensor() < WARMUP_STEPS, uniform_action, sampled_action) in a loop ported from train.py:141)"""
        return torch.clamp(self._count, max=self.max_size)

    def __len__(self):
        return self.size()

    def sample_batch(self, batch_size, generator=None):
        """uniform over the stored transitions, with replacement (as the reference's memory draws indices):
        (obs, action, reward, next_obs, terminal) on the device"""
        n = torch.clamp(self._count, max=self.max_size)
        u = torch.rand(int(batch_size), device=self.device, generator=generator)
        idx = torch.clamp((u * n).to(torch.int64), max=self.max_size - 1)
        return self.obs[idx], self.action[idx], self.reward[idx], self.next_obs[idx], self.terminal[idx]

    def sample_batch_by_index(self, batch_idx):
        """(obs, action) rows at the given indices: the pair memory of the behaviour-cloning loop (BCtrain.py:144-147,
        rpm.append(agent_obs, ref_obs) / rpm.sample_batch_by_index)"""
        idx = torch.as_tensor(batch_idx, dtype=torch.int64, device=self.device)
        return self.obs[idx], self.action[idx]

    def append_pairs(self, first, second, mask=None):
        """rpm.append(agent_obs, ref_obs) of BCtrain.py:130, batched and masked: `first` rows go to the obs field, `second`
        rows to the action field (the reference builds that memory with obs_dim = student dim, act_dim = teacher dim)"""
        n = self._rows(first, self.obs_dim).shape[0]
        if self.fused:
            alive = None if mask is None else (mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)).contiguous().view(-1)
            self.begin(first, second, alive)
        else:
            self.write_before(self.slots(n, mask), first, second)

    # ---- on-disk format: one .npz with the arrays of the filled part, oldest rows first are NOT reordered (ring order)
    def save(self, path):
        n = self.size()
        np.savez(path, obs=self.obs[:n].cpu().numpy(), action=self.action[:n].cpu().numpy(),
                 reward=self.reward[:n].cpu().numpy(), terminal=self.terminal[:n].cpu().numpy(),
                 next_obs=self.next_obs[:n].cpu().numpy(),
                 other=np.array([n, int(self._pos.item())], dtype=np.int64))

    def load(self, path):
        d = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        n = int(d["other"][0])
        if n > self.max_size or d["obs"].shape[1] != self.obs_dim or d["action"].shape[1] != self.act_dim:
            raise ValueError("stored memory (%d x obs %d, act %d) does not fit this one" %
                             (n, d["obs"].shape[1], d["action"].shape[1]))
        for name in ("obs", "action", "reward", "terminal", "next_obs"):
            getattr(self, name)[:n] = torch.as_tensor(d[name][:n], dtype=torch.float32, device=self.device)
        self._pos = torch.tensor(int(d["other"][1]) % self.max_size, dtype=torch.int64, device=self.device)
        self._count = torch.tensor(n, dtype=torch.int64, device=self.device)

    # ---- helpers
    def _rows(self, x, dim):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        if x.dim() != 2 or x.shape[1] != dim:
            raise ValueError("expected [n, %d], got %s" % (dim, tuple(x.shape)))
        return x

    def _vec(self, x):
        return torch.as_tensor(x, device=self.device).to(torch.float32).view(-1)


BOOTSTRAP_ALWAYS_FROM = 2000   # train.py:148: `terminal = float(done) if episode_steps < 2000 else 0`, then 1 - terminal


def bootstrap_mask(done, episode_steps):
    """the flag the reference stores with a transition (train.py:148-149, 229-230): 1 - done while the episode is younger than
    2000 control steps, 1 (keep bootstrapping) from step 2000 on, whatever `done` says.  done [N] bool / uint8 / float,
    episode_steps: the 1-based step count of the transition."""
    if episode_steps >= BOOTSTRAP_ALWAYS_FROM:
        return torch.ones(done.numel(), device=done.device)
    return 1.0 - done.view(-1).to(torch.float32)


def _scalar_info_columns(info_keys):
    """info keys that can be summed per episode: the scalar (one-column) entries of a1_model.INFO_SLICES; like the reference's
    `if key in info.keys()` (train.py:150-151) anything else is skipped, on both storage paths."""
    from . import a1_model as A
    keep = [k for k in info_keys if k in A.INFO_SLICES and A.INFO_SLICES[k][1] - A.INFO_SLICES[k][0] == 1]
    return keep, [A.INFO_SLICES[k][0] for k in keep]


def collect_transitions(env, rpm, max_step, policy=None, action_bound=0.3, mode="predict", ETG_w=None, ETG_b=None,
                        x_noise=0, precision=0, generator=None, info_keys=("torso", "feet", "up", "tau", "stand", "badfoot", "footcontact"),
                        noise=None):
    """One episode of every robot with its transitions stored in `rpm` (run_train_episode train.py:129-179,
    run_EStrain_episode train.py:213-249 with es_rpm), batched:

      reset(ETG_w, ETG_b, x_noise); for steps = 1 .. max_step + 1:
        action  = policy.predict(obs) | policy.sample(obs) | U(-1, 1)        (mode "predict" | "sample" | "uniform")
        next_obs, reward, done, info = step(action * action_bound, donef = steps > max_step)
        rpm.append(obs, action, reward, next_obs, 1 - done)   for the robots whose episode was still running
        infos[key] += info[key], success += (info["velx"] >= 0.3)            (alive-masked)

    The action stored is the UNSCALED one, as in the reference.  Returns (episode_return [N], episode_len [N], infos)
    with infos[key] [N] the per-episode sums and infos["success_rate"] [N]."""
    n, dev = env.num_envs, env.device
    adim = env.action_space.shape[0]
    if mode not in ("predict", "sample", "uniform"):
        raise ValueError("mode must be 'predict', 'sample' or 'uniform'")
    if mode != "uniform" and policy is None:
        raise ValueError("mode %r needs a policy" % mode)
    obs, _ = env.reset(ETG_w=ETG_w, ETG_b=ETG_b, x_noise=x_noise)

    step_no = [0]

    def act(obs):
        k = step_no[0]
        step_no[0] += 1
        if mode == "uniform":
            return torch.rand(n, adim, device=dev, generator=generator) * 2 - 1
        if mode == "sample":   # noise [max_step + 1, N, 12]: explicit N(0,1) draws (reproducible against collect_recorded)
            return policy.sample(obs, 1.0, precision, noise=None if noise is None else noise[k], generator=generator, return_logp=False)
        return policy.predict(obs, 1.0, precision)

    info_buf = getattr(env, "info_buf", None)
    if getattr(rpm, "fused", False) and info_buf is not None:
        # HIP path: three launches per control step next to predict + step; the summed info terms are the leading columns of
        # the step's info buffer (a1_model.INFO_SLICES order), so one kernel adds them all
        from . import a1_model as A
        info_keys, cols = _scalar_info_columns(info_keys)
        n_sum = max(cols) + 1 if cols else 0
        velx = A.INFO_SLICES["velx"][0]
        sums = torch.zeros(n, n_sum + 1, device=dev)
        alive = torch.ones(n, dtype=torch.uint8, device=dev)
        scaled = torch.empty(n, adim, device=dev)
        for steps in range(1, max_step + 2):
            action = act(obs)
            slot = rpm.begin(obs, action, alive, float(action_bound), scaled)    # also writes scaled = action * action_bound
            obs, reward, done, _ = env.step(scaled, donef=(steps > max_step))
            rpm.end(slot, reward, done, obs, info_buf, n_sum, velx, sums, alive)
            if steps >= BOOTSTRAP_ALWAYS_FROM:   # the kernel stored 1 - done; rows not stored carry slot -1 -> the spare row
                rpm.terminal.index_fill_(0, torch.where(slot < 0, rpm.max_size, slot).to(torch.int64), 1.0)
        ret, ln = env.episode_stats()
        infos = {k: sums[:, c] for k, c in zip(info_keys, cols)}
        infos["success_rate"] = sums[:, n_sum] / ln.to(torch.float32).clamp(min=1)
        return ret, ln, infos

    alive = torch.ones(n, dtype=torch.bool, device=dev)
    info_keys, _ = _scalar_info_columns(info_keys)
    infos = {k: torch.zeros(n, device=dev) for k in info_keys}
    success = torch.zeros(n, device=dev)
    for steps in range(1, max_step + 2):
        action = act(obs)
        slot = rpm.slots(n, alive)
        rpm.write_before(slot, obs, action)          # the observation buffer is overwritten by the step
        obs, reward, done, info = env.step(action * action_bound, donef=(steps > max_step))
        rpm.write_after(slot, reward, obs, bootstrap_mask(done, steps))
        af = alive.to(torch.float32)
        for k in info_keys:
            if k in info:
                infos[k] += af * info[k].view(-1)
        success += af * (info["velx"] >= 0.3).to(torch.float32)
        alive = alive & ~done.view(-1).to(torch.bool)
    ret, ln = env.episode_stats()
    infos["success_rate"] = success / ln.to(torch.float32).clamp(min=1)
    return ret, ln, infos


def obs2noise(obs, generator=None):
    """BCtrain.py:53-59 on a batch: Gaussian sensor noise on the normalised rpy (6e-2 / 0.1), rpy rate (1e-1 / 0.5), motor
    angle (1e-2 / 0.1) and motor velocity (0.5) columns of the 49-float observation; returns a new tensor."""
    out = obs.clone()
    for a, b, sd in ((7, 10, 6e-2 / 0.1), (10, 13, 1e-1 / 0.5), (13, 25, 1e-2 / 0.1), (25, 37, 0.5)):
        out[:, a:b] += torch.randn(obs.shape[0], b - a, device=obs.device, generator=generator) * sd
    return out


def collect_bc_pairs(env, rpm, max_step, student=None, action_bound=0.3, sensor_noise=True, mode="sample", x_noise=0,
                     precision=0, generator=None):
    """The collection half of BCtrain.py:87-131, batched: every control step the student acts on ITS observation
    (cal_agent_obs: the 49-float row, optionally with obs2noise, without the 3 displacement columns) and the pair
    (student observation [46], teacher observation [49]) of every robot whose episode is still running is stored
    (rpm = DeviceReplayMemory(max_size, 46, 49)); the learner later labels the pairs with the teacher's action
    (BClearn).  mode: "sample" | "predict" (student policy) | "uniform" (the warm-up phase, BCtrain.py:101-102).
    Returns (episode_return [N], episode_len [N])."""
    n, dev = env.num_envs, env.device
    adim = env.action_space.shape[0]
    if mode != "uniform" and student is None:
        raise ValueError("mode %r needs the student policy" % mode)
    obs, _ = env.reset(x_noise=x_noise)
    if obs.shape[1] != rpm.act_dim or obs.shape[1] - 3 != rpm.obs_dim:
        raise ValueError("the pair memory must be DeviceReplayMemory(max_size, %d, %d)" % (obs.shape[1] - 3, obs.shape[1]))
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    for steps in range(1, max_step + 2):
        agent_obs = (obs2noise(obs, generator) if sensor_noise else obs)[:, 3:].contiguous()
        rpm.append_pairs(agent_obs, obs, alive)                 # before the step overwrites the observation buffer
        if mode == "uniform":
            action = torch.rand(n, adim, device=dev, generator=generator) * 2 - 1
        elif mode == "sample":
            action = student.sample(agent_obs, 1.0, precision, generator=generator, return_logp=False)
        else:
            action = student.predict(agent_obs, 1.0, precision)
        obs, _, done, _ = env.step(action * action_bound, donef=(steps > max_step), want_info=False)
        alive = alive & ~done.view(-1).to(torch.bool)
    return env.episode_stats()


def store_recorded(rpm, rec):
    """Move a recorded episode (env.rollout_policy_record) into the replay memory: step-major, robot-minor, only the steps up
    to and including each robot's first `done` (what collect_transitions stores step by step), terminal = bootstrap_mask(done, step).
    Returns the number of rows offered (T * N); the memory must hold at least that many."""
    obs, act, rew, done = rec["obs"], rec["action"], rec["reward"], rec["done"]
    T, N = done.shape
    dn = done.to(torch.int32)
    alive = (torch.cumsum(dn, dim=0) - dn) == 0                       # no done BEFORE this step
    nxt = torch.cat([obs[1:], rec["final_obs"][None]], dim=0)
    term = 1.0 - done.to(torch.float32)
    term[BOOTSTRAP_ALWAYS_FROM - 1:] = 1.0                            # episode step = row + 1 (bootstrap_mask)
    if N > rpm.max_size:
        raise ValueError("a step of %d robots does not fit a memory of %d" % (N, rpm.max_size))
    per = max(1, rpm.max_size // N)                                   # steps per append: a batch must fit the ring
    for t0 in range(0, T, per):
        t1 = min(T, t0 + per)
        k = (t1 - t0) * N
        rpm.append_batch(obs[t0:t1].reshape(k, -1), act[t0:t1].reshape(k, -1), rew[t0:t1].reshape(-1), nxt[t0:t1].reshape(k, -1),
                         term[t0:t1].reshape(-1), mask=alive[t0:t1].reshape(-1))
    return T * N


def collect_recorded(env, rpm, max_step, policy, action_bound=0.3, ETG_w=None, ETG_b=None, x_noise=0, precision=0, mode="predict",
                     generator=None, noise=None):
    """run_EStrain_episode with es_rpm (train.py:213-249; mode "predict": agent.predict) or the collection of run_train_episode
    (train.py:129-161; mode "sample": agent.sample, the squashed-Gaussian actor on N(0,1) draws made here or passed as `noise`
    [max_step + 1, N, 12]) for all robots through the FUSED closed-loop kernel: reset, one recorded rollout of max_step + 1
    control steps, rows into `rpm`.  The forced `done` of the last step (donef = steps > max_step) is applied to the recorded
    flags.  Returns (ret [N], len [N])."""
    env.reset(ETG_w=ETG_w, ETG_b=ETG_b, x_noise=x_noise)
    if mode == "sample" and noise is None:
        noise = torch.randn(max_step + 1, env.num_envs, 12, device=env.device, generator=generator)
    elif mode not in ("predict", "sample"):
        raise ValueError("mode must be 'predict' or 'sample'")
    ret, ln, rec = env.rollout_policy_record(policy, max_step + 1, action_bound, precision, noise=noise if mode == "sample" else None)
    rec["done"] = rec["done"].clone()
    rec["done"][-1] = True                                            # donef of the last step ends every running episode
    store_recorded(rpm, rec)
    return ret, ln
