"""-m gpu, round 4: the contact model with Bullet's sweep order and in-loop joint limits (the library default), and the
robot-layer gaps VERDICT r03 listed -- all through the C-ABI against the fp64 oracle, both lane mappings.

  * joint-limit rows INSIDE the sweeps (btMultiBodyJointLimitConstraint rows before the contact rows): a walking robot whose calf
    joints ride their upper stop, against the oracle, sweep counts included;
  * motor strength ratios (laikago_motor.py:67-76,138,167), the command clip around the delayed reading (a1.py:439-457),
    foot restitution (minitaur.py:1112-1122);
  * the settings the default follows: warm start 0.1 on the normal rows only, contact slop, friction rows skipped while the
    normal impulse is zero -- each switched to its other value changes the trajectory AND still matches the oracle.
"""
import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_gpu_parity import _need_gpu, _etg_params, _make, _oracle, _lt   # noqa: E402
from tests.test_gpu_parity2 import _say                                       # noqa: E402
from tests.parity_util import OracleEnsemble, sens_robots                     # noqa: E402


def _run_pair(env, ens, acts, strength=None, W=None, B=None):
    """step the env and the oracle ensemble (tests/parity_util.OracleEnsemble: fp64 nominal, fp32, +-1 ulp actions, nudged stopping
    threshold) through `acts`; per robot: the GPU's worst joint gap to the fp64 oracle, the ensemble's own worst spread, and the
    GPU's worst base-position gap overall"""
    if strength is not None:
        env.set_motor_strength_ratios(torch.as_tensor(strength, dtype=torch.float32))
        ens.set_motor_strength(strength)
    if W is not None:
        ens.set_params(etg_w=W, etg_b=B)
    ens.reset()
    env.reset(ETG_w=W, ETG_b=B) if W is not None else env.reset()
    n = env.num_envs
    eg, sp, ep = np.zeros(n), np.zeros(n), 0.0
    for a in acts:
        env.step(torch.as_tensor(a, dtype=torch.float32))
        ens.step(a, want_info=False)
        sg, so = env.get_state().cpu().numpy(), ens.get_state()
        eg = np.maximum(eg, np.abs(sg - so)[:, 13:25].max(1))
        sp = np.maximum(sp, ens.spread(slice(13, 25)))
        ep = max(ep, np.abs(sg - so)[:, :3].max())
    return eg, sp, ep


@pytest.mark.parametrize("lanes", [4, 16])
@pytest.mark.parametrize("option", ["strength", "strength_torque_mode", "clip_delayed", "restitution", "warmstart_085", "pyramid_no_slop"])
def test_round4_options_match_oracle(lanes, option):
    _need_gpu()
    n = 64
    rng = np.random.default_rng(31)
    W, B = _etg_params(n, seed=31)
    mk, ok, strength, scale = dict(lanes_per_robot=lanes), {}, None, 0.15
    if option == "strength":
        mk.update(motor_torque_limits=12.0); ok.update(torque_limit=12.0)
        strength = rng.uniform(0.4, 1.0, size=(n, 12))
    elif option == "strength_torque_mode":
        # commands of up to 9 N m against a 6 N m limit: the TORQUE branch of the motor model returns before the clip
        # (laikago_motor.py:137-139).  Random torques on every joint are chaotic within ~6 steps: 4 steps are compared.
        mk.update(motor_control_mode="torque", motor_torque_limits=6.0, body_contacts=0); ok.update(motor_mode=1, torque_limit=6.0, body_contacts=0)   # (the robots collapse: toe spheres only keeps 4 steps comparable)
        strength = rng.uniform(0.4, 1.0, size=(n, 12))
        scale, W, B = 9.0, None, None
    elif option == "clip_delayed":
        mk.update(enable_clip_motor_commands=True); ok.update(clip_motor_commands=0.2)
        scale = 0.6
    elif option == "restitution":
        mk.update(foot_restitution=0.6); ok.update(foot_restitution=0.6)
    elif option == "warmstart_085":      # Bullet's own default factor, on the friction rows too
        mk.update(warmstart=0.85, warmstart_friction=0.85); ok.update(warmstart=0.85, warmstart_friction=0.85)
    else:
        mk.update(friction_model=1, contact_slop=0.0); ok.update(friction_model=1, contact_slop=0.0)
    env, ens = _make(n, **mk), OracleEnsemble(n, E=5, E64=3, seed=31, **ok)
    orc = ens.nominal
    acts = [rng.uniform(-scale, scale, size=(n, 12)) for _ in range(4 if option == "strength_torque_mode" else 12)]
    eg, sp, wp = _run_pair(env, ens, acts, strength, W, B)
    _say("round-4 option %-22s lanes %2d: joints vs the fp64 oracle median %.2e max %.2e rad (ensemble spread: %.2e / %.2e), base %.2e m"
         % (option, lanes, np.median(eg), eg.max(), np.median(sp), sp.max(), wp))
    _lt(np.median(eg), 2e-5, "round-4 option %s lanes %d: median joint gap" % (option, lanes))
    # EVERY robot within the floor + 4 x its own ensemble spread (torque commands drive joints onto their stops, restitution
    # switches on at an approach speed of exactly 0.2 m/s, a grip starts or stops: where the last bit decides, the ensemble parts
    # too, and says by how much)
    sens_robots(eg, sp, 5e-5, "round-4 option %s lanes %d: joint angles, %d steps" % (option, lanes, len(acts)))
    # and the option matters: the default configuration moves differently
    ref = _oracle(n, **({"motor_mode": 1, "body_contacts": 0} if option == "strength_torque_mode" else {}))
    if W is not None:
        ref.set_params(etg_w=W, etg_b=B)
    ref.reset()
    for a in acts:
        ref.step(a)
    assert np.abs(ref.get_state() - orc.get_state())[:, :25].max() > 1e-4
    env.close()


@pytest.mark.parametrize("lanes", [4, 16])
def test_joint_limit_rows_inside_the_sweeps_match_oracle(lanes):
    """A gait whose calf joints ride their upper stop (-0.916 rad: residual actions straighten the knees) while the feet carry
    the robot: joint-limit rows and contact rows are solved in the same sweeps (the rare branch of the tick).  State, sweep
    counts and the stops themselves against the oracle under the default stopping rule."""
    _need_gpu()
    n = 64
    W, B = _etg_params(n, seed=9)
    env, ens = _make(n, lanes_per_robot=lanes), OracleEnsemble(n, E=3, seed=9)
    orc = ens.nominal
    env.reset(ETG_w=W, ETG_b=B)
    ens.set_params(etg_w=W, etg_b=B)
    ens.reset()
    rng = np.random.default_rng(12)
    eg = np.zeros(n)
    e32 = np.zeros(n)
    at_stop = 0
    per_wave = 64 // lanes
    for k in range(16):
        act = rng.uniform(-0.05, 0.05, size=(n, 12))
        act[:, 2::3] += 0.95                      # knees towards straight: the PD target is past the calf joint's upper bound
        _, _, _, info = env.step(torch.as_tensor(act, dtype=torch.float32))
        _, _, _, io = ens.step(act)
        sg, so = env.get_state().cpu().numpy(), ens.get_state()
        eg = np.maximum(eg, np.abs(sg - so)[:, 13:25].max(1))
        e32 = np.maximum(e32, ens.spread(slice(13, 25)))
        at_stop += int((so[:, 15:25:3] >= A.JOINT_UPPER[2] - 1e-3).sum())
        sw_g = info["solver_sweeps"].cpu().numpy().reshape(-1, per_wave)
        sw_o = io[:, A.INFO_SWEEPS].reshape(-1, per_wave)
        assert np.all(sw_g[:, 0] >= sw_o.max(1) - 3) and np.all(sw_g[:, 0] <= sw_o.sum(1) + 3), (k, sw_g[:, 0], sw_o.max(1))
    q = env.get_state()[:, 13:25].cpu().numpy().reshape(n, 4, 3)
    _say("joint-limit rows in the sweeps, lanes %d: joints vs the fp64 oracle median %.2e max %.2e (ensemble spread: %.2e / %.2e); %d "
         "joint-steps at the calf stop" % (lanes, np.median(eg), eg.max(), np.median(e32), e32.max(), at_stop))
    assert at_stop > 100                                             # the rows were there
    assert (q[:, :, 2] <= A.JOINT_UPPER[2] + 0.02).all()              # and held
    # A joint RESTING on its stop sits exactly at the bound (the violation decays geometrically under erp), where the row's
    # activation test q >= upper is decided by the last bit: the fp64 oracle and any fp32 evaluation -- its own fp32 build
    # included -- drop the row on different ticks, and the driven joint then jumps a few mrad before the row is back (Bullet
    # has the same test: btMultiBodyJointLimitConstraint skips a row whose position error is > 0).  So the GPU is held to the
    # trajectory's own sensitivity: EVERY robot within the floor + 4 x its own ensemble spread, and tight where that one is tight.
    _lt(np.median(eg), 2e-5, "joint-limit rows inside the sweeps, lanes %d: median joint gap" % lanes)
    sens_robots(eg, e32, 1e-4, "joint-limit rows inside the sweeps, lanes %d: joint angles, 16 steps" % lanes)
    env.close()


@pytest.mark.parametrize("lanes", [4, 16])
def test_prepared_next_dynamics_under_the_all_options_kernels(lanes):
    """etg_prepare_next_dynamics settles the next episodes on scratch state with the NEXT dynamic rows -- and with the robots' own
    motor strength ratios, which are not part of those rows.  Found by the round-4 soak run: with any robot-layer option on
    (random pushes here: the non-PLAIN kernels read the ratios) the scratch settle ran on zeroed ratios, left the robot limp, and
    every restarted episode ended after one step.  Restarted robots must stand, and their episodes must last."""
    _need_gpu()
    n = 128
    env = _make(n, auto_reset=True, lanes_per_robot=lanes, seed=4, random_dynamics_refresh=1000,
                random_param={"random_dynamics": 1, "random_force": 1})
    env.reset()
    for _ in range(env._nx_first):                     # the env's first prepare call
        env.step(None, want_info=False)
    obs, _, d, _ = env.step(None, donef=True)           # everybody restarts on the prepared rows and the prepared settle
    assert bool(d.all())
    z = env.get_state()[:, 2]
    assert float(z.min()) > 0.2, float(z.min())         # standing (a limp settle leaves the trunk at ~0.06 m)
    alive = torch.ones(n, dtype=torch.bool, device="cuda:0")
    for _ in range(6):
        _, _, d, _ = env.step(None, want_info=False)
        alive &= ~d.view(-1).bool()
    assert float(alive.float().mean()) > 0.9            # the zero-residual gait walks on for the next steps
    env.close()


def test_rollout_policy_takes_the_faster_route_and_says_so_when_forced():
    """env.rollout_policy(fused=None) runs the fused closed-loop kernel where it is ahead (the 16-lane mapping; the 4-lane one
    without body rows) and predict() + step() where the kernel's register budget spills (4 lanes with body rows: 25 % behind,
    tools/closed_loop_probe.py); fused=True insists on the kernel -- same trajectory to rounding -- and raises outside it."""
    _need_gpu()
    from paddlerobotics_amd.env import FusedKernelUnavailable
    from paddlerobotics_amd.policy import MfmaPolicy
    n = 64
    pol = MfmaPolicy(49, 12)
    pol.load_state_dict(MfmaPolicy.init_like_reference(49, 12, seed=2))
    auto, forced, stepped = (_make(n, lanes_per_robot=4, body_contacts=2) for _ in range(3))
    for e in (auto, forced, stepped):
        e.reset()
    ra, la = auto.rollout_policy(pol, 8, 0.3)
    rf, lf = forced.rollout_policy(pol, 8, 0.3, fused=True)
    for _ in range(8):
        stepped.step(pol.predict(stepped.obs, 0.3), want_info=False)
    assert torch.equal(auto.get_state(), stepped.get_state())                      # the auto route IS the stepping loop here
    gap = (forced.get_state() - stepped.get_state())[:, 13:25].abs().max().item()
    _lt(gap, 5e-5, "4 lanes + body rows, fused closed loop forced vs predict + step: joint gap after 8 steps")
    assert torch.equal(la, lf) and torch.allclose(ra, rf, rtol=1e-3, atol=1e-3)
    hyb = _make(n, motor_control_mode="hybrid")
    hyb.reset()
    with pytest.raises(FusedKernelUnavailable):
        hyb.rollout_policy(pol, 2, 0.3, fused=True)
    for e in (auto, forced, stepped, hyb):
        e.close()


@pytest.mark.parametrize("lanes", [4, 16])
def test_collapsed_robots_joint_stops_and_body_rows_match_oracle(lanes):
    """The heaviest tail of the tick on the device: robots that collapse onto folded legs (torque mode, small random torques from a
    crouch) run joint-limit rows, foot rows and the body contacts' normal + friction rows in one solve, up to the 50-sweep cap, on
    every tick -- through the all-options kernels, whose body tail tracks the second rows' velocity through the joint phase too.
    Same scenario as the CPU suite's test_joint_stops_under_loaded_body_rows_match_oracle (the emulation), here through the C-ABI;
    64 robots so that waves mix robots with and without such rows."""
    _need_gpu()
    n = 64
    env = _make(n, lanes_per_robot=lanes, motor_control_mode="torque")
    ens = OracleEnsemble(n, E=3, seed=5, motor_mode=1)
    orc = ens.nominal
    env.reset()
    ens.reset()
    rng = np.random.default_rng(5)
    st = orc.get_state().copy()
    st[:, 2] = 0.16 + 0.01 * rng.uniform(size=n)
    st[:, 7:13] = 0.0
    st[:, 13:25] = np.tile([0.0, 1.2, -2.55], 4)[None, :] + 0.03 * rng.normal(size=(n, 12))
    st[:, 25:37] = 0.0
    st[n // 2:, 2] += 0.12                                           # half of the robots start higher: they land a step or two later
    st = st.astype(np.float32).astype(np.float64)                    # (what the GPU holds)
    env.set_state(torch.as_tensor(st, dtype=torch.float32))
    ens.set_state(st)
    orc.body_stats()
    lo, hi = np.array(A.JOINT_LOWER * 4), np.array(A.JOINT_UPPER * 4)
    eg, e32, both = np.zeros(n), np.zeros(n), 0
    for k in range(6):
        a = rng.uniform(-1.0, 1.0, size=(n, 12))
        env.step(torch.as_tensor(a, dtype=torch.float32))
        ens.step(a, want_info=False)
        sg, so = env.get_state().cpu().numpy(), ens.get_state()
        eg = np.maximum(eg, np.abs(sg - so)[:, 13:25].max(1))
        e32 = np.maximum(e32, ens.spread(slice(13, 25)))
        at_stop = ((so[:, 13:25] >= hi - 1e-9) | (so[:, 13:25] <= lo + 1e-9)).any(1)
        both += int((at_stop & (orc.body_stats()[:, 1] > 0)).sum())
    _say("collapsed robots, lanes %2d: joints vs the fp64 oracle median %.2e q90 %.2e max %.2e rad (ensemble spread: %.2e / %.2e / %.2e), %d robot-steps with a joint at a stop AND a loaded body row"
         % (lanes, np.median(eg), np.quantile(eg, 0.9), eg.max(), np.median(e32), np.quantile(e32, 0.9), e32.max(), both))
    assert both >= 40, both
    _lt(np.median(eg), 3e-5, "collapsed robots lanes %d: median joint gap" % lanes)
    sens_robots(eg, e32, 1e-4, "collapsed robots lanes %d: joint angles, 6 steps of random torques" % lanes)
    env.close()
