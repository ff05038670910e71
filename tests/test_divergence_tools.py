"""CPU suite: the first-divergence machinery of tests/divergence.py (used on the GPU by tests/test_gpu_regressions.py), driven
here with the host build of the kernel source (tests/emu) standing in for the device: two emulated envs take the place of the
GPU's trajectory env and probe env, so the run exercises the state / contact-impulse hand-over, the oracle ensemble, the tick
traces of both sides and the classification -- not the device code."""
import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A

torch = pytest.importorskip("torch")


class EmuAsEnv:
    """the few VecEnv methods lockstep_offenders() uses, on an EmuSim"""

    def __init__(self, emu):
        self.emu, self.num_envs, self.cfg = emu, emu.N, emu.cfg

    def get_state(self): return torch.as_tensor(self.emu.get_state())

    def get_contact_impulses(self): return torch.as_tensor(self.emu.get_contact_impulses())

    def set_state(self, st): self.emu.set_state(st.numpy())

    def set_contact_impulses(self, lam): self.emu.set_contact_impulses(lam.numpy())

    def step(self, a, want_info=False): self.emu.step(a.numpy())


def _gaits(n, seed):
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    rng = np.random.default_rng(seed)
    W, B = np.zeros((n, 3, 20)), np.zeros((n, 3))
    for i in range(n):
        W[i], B[i], _ = Opt_with_points(layer, ETG_T=0.5, w0=w0, b0=b0, points=prior + 0.02 * rng.normal(size=(6, 2)))
    return W, B


@pytest.mark.parametrize("blend", [0.0, 1e-3])
def test_lockstep_run_and_classification_on_the_host_build(blend):
    from tests.emu.emu import EmuSim
    from tests.parity_util import OracleEnsemble
    from tests.divergence import lockstep_offenders, emu_tick_rows, first_decision_gap
    n, steps = 8, 6
    W, B = _gaits(n, 4)
    cfg = lambda: A.default_config(n, settle_ticks=150, body_blend=blend)
    run, probe, emu = (EmuSim(cfg(), lanes=16) for _ in range(3))
    ens = OracleEnsemble(n, E=2, seed=3, cfg=cfg(), threads=4)
    f = np.zeros((n, 3)); f[:, 1] = np.linspace(0.0, 60.0, n)        # (some robots are thrown over: body contacts load)
    for o in (run, probe, emu, ens):
        o.set_params(etg_w=W, etg_b=B)
        o.reset()
        o.set_external_force(f)
    for k in range(25):                                               # into the interesting part: some on their side by now
        a = np.zeros((n, 12), np.float32)
        run.step(a); ens.step(a, want_info=False); emu.step(a); probe.step(a)
    rng = np.random.default_rng(0)
    lines = []
    tally, off = lockstep_offenders(EmuAsEnv(run), EmuAsEnv(probe), ens, emu, steps, lambda k: rng.uniform(-0.1, 0.1, size=(n, 12)),
                                    say=lines.append, what="host build")
    assert tally["pairs"] == n * steps and tally["nominal"] + tally["other"] + tally["none"] == tally["pairs"]
    assert tally["nominal"] >= 0.8 * tally["pairs"], lines
    for o in off:                                                     # the "device" here IS the emulation: reproduced exactly
        assert o["emu_gap"] == 0.0 and o["kind"] != "?"
    if blend > 0:
        assert tally["none"] == 0, lines
    # the trace decoders on the last step: the emulation's and the oracle's row sets agree for a robot on the nominal branch
    etr, otr, cnt = emu._trace, ens.nominal._trace_all, ens.nominal.trace_counts()
    assert np.all(cnt == 13)
    g = emu_tick_rows(etr, 0, 12)
    assert g["act"] == int(otr[0, 12, 0]) and g["sweeps_cum"] == int(otr[0, :13, 3].sum())
    assert first_decision_gap(etr, 0, otr[0, :13], float(ens.cfg.contact_margin)) is None
