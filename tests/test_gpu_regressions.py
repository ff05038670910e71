"""-m gpu, round 6: the named outliers of round 5, each with a first-divergence report.

Round 5 left three cases whose worst robots sat 1e-3 .. 1e-2 rad off the fp64 oracle with nothing on record about WHY:
  * test_residual_rule_matches_oracle[flat-16] (gait seed 21, action seed 2): one robot 8.5e-3 rad off after 20 steps;
  * tools/fuzz_parity.py --seed 505, trial 83 and --seed 606, trial 11: fewer than 90 % of the robots within their sensitivity
    (both: per-robot dynamics with foot friction up to 3.2, body contacts with friction).

What they were (gpurun_out/r06_b -> profiles/r06_first_divergence_*.txt, found with tools/first_divergence.py on a tracing build
of the library): every one of them starts at a control step in which a shin lies nearly flat, the knee sphere and the
shin-midpoint sphere are at the same height to ~1e-7 m, and the HARD deepest-of-three choice of the round-5 body contact
(EtgConfig.body_blend = 0) takes the other sphere on the GPU than in the oracle: the contact point jumps by half a shin
(0.1 m) and the step lands 1e-4 .. 1e-2 rad elsewhere.  The kernel SOURCE run on the host (tests/emu: plain C++, no FMA
contraction) sides with the GPU in 7 of 8, the fp32 and fp64 oracles side with each other: not a defect of either side, but a
contact model that is discontinuous exactly where a fallen robot rests.  Round 6 replaces the hard choice by the blended
body contact (body_blend = 1e-3 m, the default: include/etgsim.h), which has no such tie.

Here, per case and without a tracing build:
  * [hard]    body_blend = 0: the one-step consistency run along the GPU's own trajectory (tests/test_gpu_parity5.py); every
              (robot, step) pair off every branch of the oracle ensemble must be EXPLAINED: the emulation reproduces the GPU's
              result and its tick trace first differs from the fp64 oracle's in the sphere of a body contact, or the oracle's
              own trace shows the sphere of an active body contact changing within that step (the tie is crossed in the step);
  * [default] the same scenario under the default model: no pair may be off every branch.
The printed lines are the report (profiles/r06_parity_report.txt keeps them)."""
import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_gpu_parity import _need_gpu, _etg_params, _make   # noqa: E402
from tests.parity_util import OracleEnsemble   # noqa: E402
from tests.divergence import lockstep_offenders, oracle_pick_changes   # noqa: E402
from tests import fuzz_cases   # noqa: E402


class EnvAsOracle:
    """the installers tests/fuzz_cases.setup_trial calls on an oracle, forwarded to a second GPU env (the probe)"""

    def __init__(self, env):
        self.env, self._wb = env, None

    def set_heightfield(self, h): pass                      # (make_env built the same terrain from the same keyword arguments)

    def set_sensor_noise(self, stdev, seed=0): pass         # (observation_noise_stdev is a make_env keyword: already on)

    def set_reset_offsets(self, xy): self.env.set_reset_offsets(torch.as_tensor(np.asarray(xy), dtype=torch.float32))

    def set_params(self, dyn=None, etg_w=None, etg_b=None):
        if dyn is not None:
            self.env.set_dynamic_param(torch.as_tensor(np.asarray(dyn), dtype=torch.float32, device="cuda:0"))
        if etg_w is not None:
            self._wb = (etg_w, etg_b)

    def set_motor_strength(self, sr): self.env.set_motor_strength_ratios(torch.as_tensor(np.asarray(sr), dtype=torch.float32))

    def set_external_force(self, f): self.env.set_external_force(torch.as_tensor(np.asarray(f), dtype=torch.float32))

    def reset(self):
        self.env.reset(ETG_w=self._wb[0], ETG_b=self._wb[1]) if self._wb else self.env.reset()


def _fuzz_case(seed, trial, n, blend_kw):
    """the trial's env (trajectory), a probe env, the ensemble and the emulation -- every installer of the trial on all of them"""
    from tests.emu.emu import EmuSim
    from paddlerobotics_amd.env import make_env
    # the probe env takes the trial's keywords, which setup_trial draws: the same draw from an identical generator
    lanes, kw, ex = fuzz_cases.draw(fuzz_cases.trial_rng(seed, trial))
    kw.update(blend_kw)
    if ex["noise"]:
        kw["observation_noise_stdev"] = fuzz_cases.NOISE
    assert lanes == 16
    probe_env = make_env("Quadrupedal", num_envs=n, device="cuda:0", lanes_per_robot=lanes, seed=trial, **kw)
    made = {}

    def make_oracles(cfg):
        c = lambda: type(cfg).from_buffer_copy(cfg)
        made["ens"] = OracleEnsemble(n, E=4, seed=seed + trial, cfg=c())
        made["emu"] = EmuSim(c(), lanes=16)
        return [made["ens"], made["emu"], EnvAsOracle(probe_env)]

    T = fuzz_cases.setup_trial(seed, trial, n, make_oracles, kw_extra=blend_kw)
    adim = T["env"].action_space.shape[0]
    action_fn = lambda k: fuzz_cases.trial_action(T["rng"], T["mode"], n, adim)
    return T["env"], probe_env, made["ens"], made["emu"], action_fn, fuzz_cases.short(T["kw"])


def _plain_case(n, gait_seed, act_seed, amp, blend_kw):
    """test_residual_rule_matches_oracle[flat-16]'s scenario: default config, per-robot gaits, U(-amp, amp) residual actions"""
    from tests.emu.emu import EmuSim
    W, B = _etg_params(n, seed=gait_seed)
    run, probe = _make(n, lanes_per_robot=16, **blend_kw), _make(n, lanes_per_robot=16, **blend_kw)
    cfg = lambda: type(run.cfg).from_buffer_copy(run.cfg)
    ens = OracleEnsemble(n, E=4, seed=gait_seed, cfg=cfg())
    emu = EmuSim(cfg(), lanes=16)
    for e in (run, probe):
        e.reset(ETG_w=W, ETG_b=B)
    for o in (ens, emu):
        o.set_params(etg_w=W, etg_b=B)
        o.reset()
    rng = np.random.default_rng(act_seed)
    return run, probe, ens, emu, (lambda k: rng.uniform(-amp, amp, size=(n, 12))), "default config"


CASES = {
    "residual_rule_flat16_seed21": lambda bk: _plain_case(64, 21, 2, 0.1, bk) + (20,),
    "fuzz_seed505_trial83": lambda bk: _fuzz_case(505, 83, 32, bk) + (12,),
    "fuzz_seed606_trial11": lambda bk: _fuzz_case(606, 11, 32, bk) + (12,),
}


@pytest.mark.parametrize("model", ["hard", "default"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_named_outlier_of_round5(case, model):
    _need_gpu()
    blend_kw = dict(body_blend=0.0) if model == "hard" else {}
    run, probe, ens, emu, action_fn, text, steps = CASES[case](blend_kw)
    assert abs(float(run.cfg.body_blend) - (0.0 if model == "hard" else 1e-3)) < 1e-9
    what = "%s [%s body contact]" % (case, "hard deepest-of-three" if model == "hard" else "blended (default)")
    print("[parity] %s: %d robots x %d steps, %s" % (what, run.num_envs, steps, text), flush=True)
    tally, off = lockstep_offenders(run, probe, ens, emu, steps, action_fn, what=what)
    run.close(); probe.close()
    assert tally["nominal"] >= 0.9 * tally["pairs"]
    if model == "default":
        assert tally["none"] == 0, off[:5]
        return
    # hard choice: every pair off the ensemble is a body-sphere tie (the kernel source's first differing decision, or -- where the
    # host build of the source rounds the tie the oracle's way -- the oracle's own sphere changing within the step)
    unexplained = [o for o in off if not (o["kind"] == "sphere" or o["tie_in_step"])]
    assert not unexplained, unexplained[:5]
