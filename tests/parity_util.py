"""Per-robot parity criteria for the -m gpu tests (test infrastructure; imports the oracle).

Contact dynamics are discontinuous: a sphere that enters the 0.02 m margin, or a friction row that starts to grip, one tick
earlier moves a robot's trajectory by 1e-4 .. 1e-2 rad within a control step, and whether it happens a tick earlier is decided
by the last bits of the arithmetic.  A fixed bound therefore only holds on smooth stretches.  Round 5 excused 10 % of the
robots for that reason without looking at them; here every robot is held to its OWN trajectory's sensitivity:

  * `OracleEnsemble` steps, next to the fp64 oracle, INDEPENDENT fp32-LEVEL EVALUATIONS of the same trajectory: the fp32 build of
    the oracle, E more fp32 oracles whose actions are the fp32 actions the GPU receives nudged by +-1 fp32 ulp (the smallest
    input change an fp32 implementation cannot tell from the original; it also re-shuffles every later rounding), and E64 fp64
    oracles with nudged actions (the pure input sensitivity).  `spread()` is, per robot, the largest distance of any member
    from the nominal fp64 trajectory.
  * `sens_robots(err_gpu, spread, floor, what)`: a robot is INSIDE when err_gpu <= floor + 4 * spread.  A robot on a smooth
    stretch has spread ~ 1e-7 and is held to the floor; a robot whose members part from each other is held to 4 x the distance by
    which they part.  The printed [parity] line says how many robots needed the allowance and how large their spread was
    (profiles/r06_parity_report.txt).  How many robots may be OUTSIDE is measured, not chosen: every member of the ensemble is
    judged the same way against the REST of the ensemble (leave-one-out: member m's gap against floor + 4 x the spread of the
    others) -- an independent fp32 evaluation that parts alone on a robot is outside on it, exactly like a GPU that parts alone.
    The GPU may be outside on as many robots as the worst member is, plus one per 64 robots: a robot at a bifurcation that every
    evaluation passes with probability p of parting is parted by the GPU ALONE with probability p (1 - p)^M -- a few per cent
    for M = 8..10 members and p ~ 0.1, so over the ~60 criterion lines of a run one such robot is expected, and a count of
    zero cannot be demanded of any evaluation, the oracle's own fp32 build included (profiles/r06_outlier_ws085_lanes4.txt: the
    one robot of `warmstart_085 lanes 4` -- restarted from the GPU's own states, the GPU stays on the fp64 oracle's branch at
    every step and it is the fp32 ORACLE that parts, by the same 1.1e-3 rad).  Every outside robot is listed in the [parity]
    line.  Without the ensemble's per-member record (a spread array the test built some other way) no robot may be outside.
  * `nearest_member(...)`: for ONE control step from a synchronised state (tests/test_gpu_parity5.py) the criterion is sharper:
    the GPU's result must lie within the floor of SOME member's result -- the step map is discontinuous, the GPU has to be on
    one of its branches.
"""
import os

import numpy as np

from paddlerobotics_amd import a1_model as A

NCPU = os.cpu_count() or 1
REPORT = []          # (what, n_robots, n_allowance, text): the tests' own tally (printed per line, summarised by conftest)
ENSEMBLES = []       # every OracleEnsemble made (sens_robots finds the per-member record behind a spread array here)


def _member_record(spread):
    """the per-member worst gaps [M, n] whose maximum over the members IS the given spread array (the test accumulated
    ens.spread(cols) over the same steps), or None"""
    for ens in reversed(ENSEMBLES[-4:]):
        for per in ens._acc.values():
            if per.shape[1:] == np.shape(spread) and np.array_equal(per.max(0), spread):
                return per
    return None


def _oracle(n, dtype=np.float64, **kw):
    from oracle.oracle import OracleSim
    return OracleSim(A.default_config(n, **kw), dtype=dtype)


def ulp_nudge(a32, rng):
    """fp32 array -> the same values moved by -1, 0 or +1 fp32 ulp, as float64"""
    a32 = np.asarray(a32, dtype=np.float32)
    d = rng.integers(-1, 2, size=a32.shape)
    up, dn = np.nextafter(a32, np.float32(np.inf)), np.nextafter(a32, np.float32(-np.inf))
    return np.where(d > 0, up, np.where(d < 0, dn, a32)).astype(np.float64)


class OracleEnsemble:
    """The fp64 oracle (`nominal`), the fp32 oracle, E fp32 oracles and E64 fp64 oracles with +-1 ulp actions, driven together.

    Construct from keyword arguments of a1_model.default_config, or from an EtgConfig (cfg=..., e.g. the env's own).  Every
    OracleSim method that installs something (set_params, set_heightfield, set_reset_offsets, set_motor_strength,
    set_external_force, set_sensor_noise, set_state) is forwarded to all members; reset / step return the nominal member's
    outputs."""

    def __init__(self, n, E=3, E64=1, seed=1234, cfg=None, threads=NCPU, rel_noise=0.0, res_nudge=2e-3, solve_noise=1e-5, **kw):
        """rel_noise > 0: the nudged members' actions are scaled by 1 + U(-1, 1) * rel_noise instead of moved by one fp32 ulp
        (2^-8 for a policy evaluated on bf16 operands: the input uncertainty of THAT arithmetic)."""
        from oracle.oracle import OracleSim
        self.n = n
        self.rel_noise = rel_noise
        mk = (lambda dt: OracleSim(type(cfg).from_buffer_copy(cfg), dtype=dt)) if cfg is not None else (lambda dt: _oracle(n, dtype=dt, **kw))
        self.nominal = mk(np.float64)
        self.o32 = mk(np.float32)
        self.nudged = [mk(np.float32) for _ in range(E)] + [mk(np.float64) for _ in range(E64)]
        # the stopping rule is a last-bit decision too (one more sweep or not: max_r (d lambda_r A_rr)^2 <= solver_residual): two
        # fp64 members whose threshold is moved by -+res_nudge (relative) take the other branch where the test is that close
        self.res_members = []
        if res_nudge and self.nominal.cfg.solver_residual > 0:
            for sgn in (-1.0, 1.0):
                c = type(self.nominal.cfg).from_buffer_copy(self.nominal.cfg)
                c.solver_residual = c.solver_residual * (1.0 + sgn * res_nudge)
                self.res_members.append(OracleSim(c, dtype=np.float64))
        # ... and the rounding noise of the SOLVE: the kernels track every row's velocity incrementally through tens of sweeps
        # in fp32; on a hard landing (impulse 1.8, 14 sweeps) the kernel source's impulses are 5e-6 relative off the fp64 solve,
        # the fp32 oracle's 10x less, and a +-1 ulp action reaches the solve as ~1e-7.  A chattering body contact amplifies
        # that x 300 within a control step (tools/first_divergence.py), so members that only carry 1-ulp noise under-state
        # what ANY fp32 solve of this shape does.  The nudged members therefore also perturb every solved impulse by
        # 1 + solve_noise * U(-1, 1) (OracleSim.set_solve_noise, a test knob of the oracle).
        if solve_noise:
            for k, o in enumerate(self.nudged):
                o.set_solve_noise(solve_noise, seed + 101 * k)
        self.members = [self.o32] + self.nudged + self.res_members
        self._acc = {}                        # spread(cols) -> running per-member worst gap [M, n] (sens_robots' leave-one-out)
        ENSEMBLES.append(self)
        self.rng = np.random.default_rng(seed)
        for o in [self.nominal] + self.members:
            o.threads = threads
        self.cfg = self.nominal.cfg

    def __getattr__(self, name):
        if name in ("set_params", "set_heightfield", "set_reset_offsets", "set_motor_strength", "set_external_force",
                    "set_sensor_noise", "set_state", "set_lambda"):
            def forward(*a, **k):
                for o in [self.nominal] + self.members:
                    getattr(o, name)(*a, **k)
            return forward
        raise AttributeError(name)

    def reset(self, mask=None):
        self._obs = [np.asarray(o.reset(mask=mask), dtype=np.float64) for o in self.members]
        out = self.nominal.reset(mask=mask)
        self._obs0 = np.asarray(out, dtype=np.float64)
        return out

    def _nudge(self, a32):
        if self.rel_noise > 0.0:
            return a32.astype(np.float64) * (1.0 + self.rel_noise * self.rng.uniform(-1.0, 1.0, size=a32.shape))
        return ulp_nudge(a32, self.rng)

    def step(self, action, donef=None, want_info=True):
        a32 = np.asarray(action, dtype=np.float32)           # what the GPU receives
        self._obs = [np.asarray(self.o32.step(a32, donef, want_info=False)[0], dtype=np.float64)]
        for o in self.nudged:
            self._obs.append(np.asarray(o.step(self._nudge(a32), donef, want_info=False)[0], dtype=np.float64))
        for o in self.res_members:
            self._obs.append(np.asarray(o.step(a32.astype(np.float64), donef, want_info=False)[0], dtype=np.float64))
        out = self.nominal.step(a32.astype(np.float64), donef, want_info=want_info)
        self._obs0 = np.asarray(out[0], dtype=np.float64)
        return out

    def closed_loop_step(self, ws, scale=0.3, col0=0):
        """one step of run_EStrain_episode (train.py:213-249) with a fixed actor: EVERY member acts on its OWN observation
        (a = tanh(mean(obs[:, col0:])) * scale through the oracle's mlp_forward; the nudged members' actions moved as in step())"""
        from oracle import oracle as O
        obs_new = []
        for o, ob in zip(self.members, self._obs):
            a = O.mlp_forward(ob[:, col0:], *ws, scale=scale)
            if o is not self.o32 and o not in self.res_members:
                a = self._nudge(a.astype(np.float32))
            obs_new.append(np.asarray(o.step(a, want_info=False)[0], dtype=np.float64))
        self._obs = obs_new
        out = self.nominal.step(O.mlp_forward(self._obs0[:, col0:], *ws, scale=scale), want_info=False)
        self._obs0 = np.asarray(out[0], dtype=np.float64)
        return out

    def get_state(self):
        return self.nominal.get_state()

    def member_states(self):
        return [np.asarray(o.get_state(), dtype=np.float64) for o in self.members]

    def spread(self, cols):
        """per robot: the largest |member - nominal| over the state columns `cols` (slice or index array)"""
        s0 = self.nominal.get_state()[:, cols]
        per = np.stack([np.abs(s[:, cols] - s0).max(1) for s in self.member_states()])
        key = repr(cols)
        self._acc[key] = per if key not in self._acc else np.maximum(self._acc[key], per)
        return per.max(0)

    def spread_o32(self, cols):
        return np.abs(np.asarray(self.o32.get_state(), dtype=np.float64)[:, cols] - self.nominal.get_state()[:, cols]).max(1)


def sens_tally(err_gpu, spread, floor, factor=4.0):
    """the counts behind sens_robots, without asserting -> dict(n, need, bad (indices), allowed, loo, text)"""
    err_gpu, spread = np.asarray(err_gpu, dtype=np.float64), np.asarray(spread, dtype=np.float64)
    n = len(err_gpu)
    need = err_gpu > floor
    bad = err_gpu > floor + factor * spread
    txt = "robots %d, within the floor %d, needed the sensitivity allowance %d" % (n, int((~need).sum()), int(need.sum()))
    if need.any():
        txt += " (their gaps %.1e .. %.1e, their ensemble spread %.1e .. %.1e)" % (err_gpu[need].min(), err_gpu[need].max(),
                                                                                   spread[need].min(), spread[need].max())
    per = _member_record(spread)
    allowed, loo, loo_txt = 0, None, ""
    if per is not None and per.shape[0] >= 3:
        loo = []
        for m in range(per.shape[0]):
            rest = np.delete(per, m, axis=0).max(0)
            loo.append(int((per[m] > floor + factor * rest).sum()))
        allowed = max(loo) + max(1, int(round(n / 64.0)))
        loo_txt = " | the members judged alike against the rest of the ensemble: %s outside -> allowed %d" % (loo, allowed)
    if bad.any():
        loo_txt += " | the GPU's outside robots: " + ", ".join("#%d gap %.1e spread %.1e" % (i, err_gpu[i], spread[i]) for i in np.nonzero(bad)[0][:6])
    return dict(n=n, need=int(need.sum()), bad=np.nonzero(bad)[0], allowed=allowed, loo=loo, text=txt, loo_text=loo_txt)


def sens_robots(err_gpu, spread, floor, what, factor=4.0):
    """A robot is inside when err_gpu <= floor + factor * spread; the GPU may be outside on as many robots as the ensemble's
    worst member is when judged the same way against the other members, plus one per 64 robots (module docstring).  Prints the
    tally of the robots that needed the allowance and of those outside."""
    err_gpu, spread = np.asarray(err_gpu, dtype=np.float64), np.asarray(spread, dtype=np.float64)
    t = sens_tally(err_gpu, spread, floor, factor)
    n = t["n"]
    print("[parity] %-70s median %.3e max %.3e (floor %.1e) | %s | outside floor + %g x spread: %d%s"
          % (what, float(np.median(err_gpu)), float(err_gpu.max()), floor, t["text"], factor, len(t["bad"]), t["loo_text"]), flush=True)
    REPORT.append((what, n, t["need"], t["text"]))
    assert np.isfinite(err_gpu).all(), what
    assert len(t["bad"]) <= t["allowed"], (what, t["bad"].tolist(), err_gpu[t["bad"]].tolist(), spread[t["bad"]].tolist())
    # a wrong kernel moves EVERY robot: the quarter of the robots whose ensemble stayed closest together is held to half the floor
    # (or, where even those part -- torque commands, joints riding their stops --, to 4 x their own median spread)
    calm = np.argsort(spread)[:max(4, n // 4)]
    assert np.median(err_gpu[calm]) < max(0.5 * floor, factor * float(np.median(spread[calm]))), (what, float(np.median(err_gpu[calm])), float(np.median(spread[calm])))


def one_step_verdict(gpu, nominal, members, floor, factor=4.0):
    """ONE control step from a synchronised state, per robot over the given columns -> (on_nominal, on_any, gap to the nominal,
    gap to the nearest member).  The step map has branches (a row inside / outside the margin, a grip, one sweep more) and, on a
    branch, it can be steep (a chattering contact amplifies the solve's rounding noise x 300 within the step):
      * on the nominal branch: |gpu - nominal| <= floor + factor x the spread of the members that stayed on that branch
        (members further than 50 x floor from the nominal took another branch and do not count as spread);
      * on another member's branch: within the same allowance of a member that parted."""
    g = np.abs(gpu - nominal).max(1)
    dev = np.stack([np.abs(m - nominal).max(1) for m in members])                  # [M, n]
    near_dev = np.where(dev < 50.0 * floor, dev, 0.0).max(0)
    allow = floor + factor * near_dev
    d = np.stack([g] + [np.abs(gpu - m).max(1) for m in members])
    on_nominal = g <= allow
    on_any = on_nominal | (d.min(0) <= allow)
    return on_nominal, on_any, g, d.min(0)


def nearest_member(gpu, nominal, members, floor):
    """per robot: min over {nominal} + members of max |gpu - member| over the given columns -> (distance, index of the nearest:
    0 = the nominal fp64 oracle)"""
    d = np.stack([np.abs(gpu - m).max(1) for m in [nominal] + list(members)], axis=0)
    return d.min(0), d.argmin(0)
