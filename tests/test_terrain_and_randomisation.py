"""Terrain tasks, terrain bands, external force, sensor_mode and randomisation helpers,
on the CPU: the kernel source runs through the host emulation (tests/emu) against the oracle."""
import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A
from paddlerobotics_amd import terrain as T


def _params(n, seed=0):
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    from paddlerobotics_amd.etg_fit import opt_with_points_batched
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    pts = prior[None] + 0.02 * np.random.default_rng(seed).normal(size=(n, 6, 2))
    w, b = opt_with_points_batched(layer, 0.5, pts, b0, w0, device="cpu")
    return w.numpy(), b.numpy()


def test_stair_and_slope_profiles():
    """stairs: n treads of the given width/height up, plateau, down to zero; ramps reach the same top."""
    x = np.linspace(-1, 12, 2601)
    for task in ("stairstair", "stairslope", "slopestair", "slopeslope"):
        z, end = T.profile(task, x, step_height=0.09, step_width=0.3, slope=0.3, n_steps=5)
        assert z[x < 0.99].max() == 0.0 and abs(z.max() - 0.45) < 1e-12
        assert np.all(z[x > end + 1e-9] == 0.0)
        assert z.min() >= 0.0
    z, _ = T.profile("stairstair", x, 0.09, 0.3, 0.3, 5)
    levels = np.unique(np.round(z, 9))
    assert np.allclose(levels, 0.09 * np.arange(6))                 # only tread heights occur
    zs, _ = T.profile("slopeslope", x, 0.09, 0.3, 0.3, 5)
    assert np.abs(np.diff(zs) / np.diff(x)).max() <= 0.3 + 1e-9     # ramp gradient = SLOPE


def test_task_heightfield_bands_use_reference_ranges():
    hf = T.make_task_heightfield("stairstair", variants=8, seed=3)
    rows = hf["heights"].shape[0] // 8
    assert hf["bands"] == 8 and hf["heights"].shape[0] == 8 * rows
    p = hf["params"]
    assert np.all((p[:, 0] >= 0.08 - 1e-9) & (p[:, 0] <= 0.1 + 1e-9))      # STEP_HEIGHT, train.py:48
    assert np.all((p[:, 1] >= 0.26 - 1e-9) & (p[:, 1] <= 0.4 + 1e-9))      # STEP_WIDTH, train.py:50
    for v in range(8):
        band = hf["heights"][v * rows:(v + 1) * rows]
        assert np.all(band == band[:1])                                     # a pure x-profile
        assert abs(band.max() - 5 * p[v, 0]) < 1e-6
    beam = T.make_task_heightfield("balancebeam", variants=1)
    mid = beam["heights"].shape[0] // 2
    assert beam["heights"][mid].max() == 0.0 and beam["heights"][0, -1] < -0.4


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_terrain_bands_match_oracle(lanes):
    """two robots on two different stair variants of one banded heightfield"""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 2
    hf = T.make_task_heightfield("stairstair", variants=2, seed=1, cell=0.05)
    # give band 1 a pedestal under the start position so that the two robots see different ground
    rows = hf["heights"].shape[0] // 2
    hf["heights"][rows:, :] += 0.03
    cfg = A.default_config(n, solver_iters=4, terrain=1, heightfield=hf)
    assert cfg.hf_bands == 2
    W, B = _params(n, seed=4)
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    for s in (orc, emu):
        s.set_heightfield(hf["heights"])
        s.set_params(etg_w=W, etg_b=B)
    orc.reset(); emu.reset()
    so, se = orc.get_state(), emu.get_state()
    assert abs((so[1, 2] - so[0, 2]) - 0.03) < 2e-3         # robot 1 stands on its own (raised) band
    assert np.abs(se[:, :7] - so[:, :7]).max() < 2e-3
    for k in range(5):
        orc.step(np.zeros((n, 12))); emu.step(np.zeros((n, 12)))
        assert np.abs(emu.get_state()[:, 13:25] - orc.get_state()[:, 13:25]).max() < 1e-2


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_external_force_matches_oracle(lanes):
    """a sideways / forward push on the trunk: same response as the oracle, and it really pushes"""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 3
    cfg = A.default_config(n, solver_iters=4)
    W, B = _params(n, seed=2)
    force = np.array([[0, 0, 0], [0, 30.0, 0], [25.0, 0, 5.0]])
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    for s in (orc, emu):
        s.set_params(etg_w=W[:1].repeat(n, 0), etg_b=B[:1].repeat(n, 0))
        s.reset()
        s.set_external_force(force)
    for k in range(8):
        orc.step(np.zeros((n, 12))); emu.step(np.zeros((n, 12)))
    so, se = orc.get_state(), emu.get_state()
    assert np.abs(se[:, :3] - so[:, :3]).max() < 3e-3 and np.abs(se[:, 13:25] - so[:, 13:25]).max() < 1e-2
    assert so[1, 1] - so[0, 1] > 0.01          # pushed to +y relative to the unforced twin
    assert so[2, 0] - so[0, 0] > 0.005         # pushed forward
    # clearing the force: robots 1 and 2 stop being accelerated (forces are part of the step, not the state)
    for s in (orc, emu):
        s.set_external_force(None)
    orc.step(np.zeros((n, 12))); emu.step(np.zeros((n, 12)))
    assert np.abs(emu.get_state()[:, :3] - orc.get_state()[:, :3]).max() < 3e-3


def test_sensor_mode_columns_follow_the_reference_obs_dim_rule():
    """deployment/test.py:26-46: motor 1/2 -> 24/12, dis -> 3, imu 1/2 -> 6/3, contact -> 4, ETG -> 12"""
    from paddlerobotics_amd.env import sensor_columns
    def dim(**kw):
        return len(sensor_columns(kw))
    assert dim() == 49
    assert dim(dis=0) == 46                      # the student observation (BCtrain.py:53-59)
    assert dim(motor=2) == 37 and dim(imu=2) == 46 and dim(contact=0) == 45 and dim(ETG=0) == 37
    assert dim(dis=0, motor=2, imu=2, contact=0, ETG=0) == 15
    cols = sensor_columns({"dis": 0})
    assert cols == list(range(3, 49))
    # imu == 2 is the rpy RATE alone (EnvWrapper.py:91-92 `sensors_dict["IMU"] = drpy`), not rpy
    assert sensor_columns({"dis": 0, "contact": 0, "motor": 0, "ETG": 0, "imu": 2}) == [10, 11, 12]
    assert sensor_columns({"dis": 0, "contact": 0, "motor": 0, "ETG": 0, "imu": 1}) == [7, 8, 9, 10, 11, 12]
    # the optional sensors of train.py:268-271 come from the etg_extra_sensors() row, appended after the 49-float row
    from paddlerobotics_amd.env import extra_sensor_columns
    assert extra_sensor_columns({}) == [] and len(extra_sensor_columns({"footpose": 1})) == 12
    assert len(extra_sensor_columns({"ETG_obs": 1, "footpose": 1, "dynamic_vec": 1, "force_vec": 1})) == 20 + 12 + 48 + 3
    assert extra_sensor_columns({"force_vec": 1}) == [80, 81, 82]


def test_param2dynamic_rows_equal_the_dict_mapping():
    rng = np.random.default_rng(0)
    P = rng.uniform(-1.5, 1.5, size=(5, 48))
    rows = A.param2dynamic_rows(P)
    for i in range(5):
        assert np.array_equal(rows[i], A.dynamic_dict_to_row(A.param2dynamic_dict(P[i])))
    assert np.array_equal(A.param2dynamic_rows(np.zeros(48))[0], A.default_dynamic_row())


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_torque_mode_matches_oracle(lanes):
    """motor_control_mode TORQUE (train.py mode_map; laikago_motor.py:140-143): the action is the torque."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 3
    cfg = A.default_config(n, solver_iters=4, motor_mode=1)
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    orc.reset(); emu.reset()
    assert np.abs(emu.get_state()[:, :7] - orc.get_state()[:, :7]).max() < 2e-3      # the settle runs under PD
    assert orc.get_state()[:, 2].min() > 0.2
    rng = np.random.default_rng(0)
    z0 = orc.get_state()[:, 2].copy()
    for k in range(4):
        tau = rng.uniform(-3.0, 3.0, size=(n, 12))
        tau[0] = 0.0                                         # robot 0: limp
        orc.step(tau); emu.step(tau)
        so, se = orc.get_state(), emu.get_state()
        assert np.abs(se[:, 13:25] - so[:, 13:25]).max() < 1e-2 and np.abs(se[:, :3] - so[:, :3]).max() < 3e-3
    # with zero torque the legs give way (in the position mode action 0 would hold the stance)
    assert orc.get_state()[0, 2] < z0[0] - 0.03


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_clip_motor_commands_matches_oracle(lanes):
    """A1._ClipMotorCommands (a1.py:439-457): the position command never leads the motor angle by more than
    0.2 rad, so a 0.6 rad command step moves the joints more gently than unclipped."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 2
    act = np.zeros((n, 12)); act[:, 1::3] = 0.6
    res = {}
    for clip in (0.0, 0.2):
        cfg = A.default_config(n, solver_iters=4, clip_motor_commands=clip)
        orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
        orc.reset(); emu.reset()
        q0 = orc.get_state()[:, 13:25].copy()
        orc.step(act); emu.step(act)
        so, se = orc.get_state(), emu.get_state()
        assert np.abs(se[:, 13:25] - so[:, 13:25]).max() < 2e-3
        res[clip] = np.abs(so[:, 13:25] - q0).max()
    assert res[0.2] < 0.8 * res[0.0]


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_hybrid_mode_matches_oracle(lanes):
    """motor mode HYBRID: per-motor (q_des, kp, qd_des, kd, tau_ff) commands, laikago_motor.py:152-167"""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 3
    cfg = A.default_config(n, solver_iters=4, motor_mode=2)
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    orc.reset(); emu.reset()
    assert np.abs(emu.get_state()[:, :7] - orc.get_state()[:, :7]).max() < 2e-3
    rng = np.random.default_rng(2)
    for k in range(4):
        cmd = np.zeros((n, 60))
        cmd[:, 0::5] = A.INIT_MOTOR_ANGLES + rng.normal(size=(n, 12)) * 0.05
        cmd[:, 1::5] = rng.uniform(60, 120, size=(n, 12))
        cmd[:, 2::5] = rng.normal(size=(n, 12)) * 0.2
        cmd[:, 3::5] = rng.uniform(0.5, 2.5, size=(n, 12))
        cmd[:, 4::5] = rng.normal(size=(n, 12)) * 0.5
        orc.step(cmd); emu.step(cmd)
        so, se = orc.get_state(), emu.get_state()
        assert np.abs(se[:, 13:25] - so[:, 13:25]).max() < 5e-3 and np.abs(se[:, :3] - so[:, :3]).max() < 3e-3
    assert orc.get_state()[:, 2].min() > 0.2          # the hybrid PD holds the stance
    # with the model's own gains and zero feed-forward the HYBRID command reproduces the POSITION mode
    pos = OracleSim(A.default_config(1, solver_iters=4)); hyb = OracleSim(A.default_config(1, solver_iters=4, motor_mode=2))
    pos.reset(); hyb.reset()
    for k in range(3):
        _, _, _, info = pos.step(np.zeros((1, 12)))
        qdes = info[0, A.INFO_SLICES["real_action"][0]:A.INFO_SLICES["real_action"][1]]
        cmd = np.zeros((1, 60)); cmd[0, 0::5] = qdes; cmd[0, 1::5] = 80.0; cmd[0, 3::5] = [1., 2., 2.] * 4
        hyb.step(cmd)
        assert np.abs(hyb.get_state() - pos.get_state()).max() < 1e-9


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_reset_offsets_match_oracle(lanes):
    """start offsets (the x_noise of env.reset): on flat ground the settle is translation invariant; on stairs the
    robot that starts closer to the first riser reaches it earlier -- same in the kernels' code and the oracle"""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 3
    xy = np.array([[0.0, 0.0], [0.08, 0.0], [-0.1, 0.05]])
    W, B = _params(n, seed=5)
    cfg = A.default_config(n, solver_iters=4)
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    for s in (orc, emu):
        s.set_params(etg_w=W[:1].repeat(n, 0), etg_b=B[:1].repeat(n, 0))
        s.set_reset_offsets(xy)
    oo, oe = orc.reset(), emu.reset()
    so, se = orc.get_state(), emu.get_state()
    assert np.abs(so[:, :2] - so[0, :2] - xy).max() < 1e-9          # flat: a pure translation of robot 0's settle
    assert np.abs(so[:, 2:] - so[0, 2:]).max() < 1e-9 and np.abs(oo - oo[0]).max() < 1e-9
    assert np.abs(se[:, :7] - so[:, :7]).max() < 2e-3 and np.abs(oe - oo).max() < 5e-2
    # stairs: a banded task heightfield, the flat approach ends at x = 1
    hf = T.make_task_heightfield("stairstair", variants=1, seed=1, cell=0.05)
    cfg = A.default_config(n, solver_iters=4, terrain=1, heightfield=hf)
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    for s in (orc, emu):
        s.set_heightfield(hf["heights"])
        s.set_params(etg_w=W[:1].repeat(n, 0), etg_b=B[:1].repeat(n, 0))
        s.set_reset_offsets(xy)
        s.reset()
    assert np.abs(emu.get_state()[:, :7] - orc.get_state()[:, :7]).max() < 2e-3
    assert np.abs(orc.get_state()[:, :2] - orc.get_state()[0, :2] - xy).max() < 1e-4
    # None clears the offsets again
    orc.set_reset_offsets(None); orc.reset()
    assert np.abs(orc.get_state()[:, 0] - orc.get_state()[0, 0]).max() < 1e-9


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_sensor_noise_matches_oracle(lanes):
    """_AddSensorNoise (minitaur.py:1206-1211): the counter-based Gaussian noise lands on the same observation columns
    with the same values in the kernels' code and in the oracle; the dynamics are untouched by it."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 3
    std = np.array([0.02, 0.3, 0.0, 0.01, 0.05], dtype=np.float32)
    cfg = A.default_config(n, solver_iters=4)
    W, B = _params(n, seed=6)
    clean, orc, emu = OracleSim(cfg), OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    for s in (clean, orc, emu):
        s.set_params(etg_w=W, etg_b=B)
    orc.set_sensor_noise(std, seed=11); emu.set_sensor_noise(std, seed=11)
    oc, oo, oe = clean.reset(), orc.reset(), emu.reset()
    devs = []
    for k in range(6):
        assert np.abs(oe - oo).max() < 5e-2
        d = oo - oc
        assert np.abs(d[:, :7]).max() == 0 and np.abs(d[:, 37:]).max() == 0     # displacement, contacts, ETG: no noise
        devs.append(d)
        assert np.abs(orc.get_state() - clean.get_state()).max() == 0            # the physics never sees the noise
        oc = clean.step(np.zeros((n, 12)))[0]
        oo = orc.step(np.zeros((n, 12)))[0]
        oe = emu.step(np.zeros((n, 12)))[0]
    d = np.stack(devs)                                                           # [6, n, 49] normalised units
    # column scales: rpy /0.1, rpy rate /0.5, angle /0.1, velocity /1
    for cols, sd in ((slice(7, 10), std[3] / 0.1), (slice(10, 13), std[4] / 0.5), (slice(13, 25), std[0] / 0.1),
                     (slice(25, 37), std[1])):
        got = d[:, :, cols].std()
        assert 0.6 * sd < got < 1.5 * sd, (cols, got, sd)
    assert np.abs(d[0] - d[1]).max() > 1e-3                                       # a fresh draw every observation
    # same seed -> same stream; another seed -> another one
    again = OracleSim(cfg); again.set_params(etg_w=W, etg_b=B); again.set_sensor_noise(std, seed=11)
    other = OracleSim(cfg); other.set_params(etg_w=W, etg_b=B); other.set_sensor_noise(std, seed=12)
    ref = OracleSim(cfg); ref.set_params(etg_w=W, etg_b=B); ref.set_sensor_noise(std, seed=11)
    assert np.array_equal(again.reset(), ref.reset()) and not np.array_equal(other.reset(), ref.reset())
    orc.set_sensor_noise(None)
    assert np.abs(orc.step(np.zeros((n, 12)))[0] - clean.step(np.zeros((n, 12)))[0]).max() == 0


@pytest.mark.parametrize("lanes", [16, 4])
def test_emulation_knee_contacts_match_oracle(lanes):
    """body_contacts: a limp robot (TORQUE mode, no torque) folds onto its knees; the knee spheres then carry it.  The kernel
    source of both mappings (16 lanes: the 4th lane of every leg owns the body row; 4 lanes: a 4th row on the leg's lane, 4 x 4
    Delassus blocks) against the oracle's 16-row formulation."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 2
    hf = {"heights": np.zeros((64, 64), dtype=np.float32), "cell": 0.1, "origin": (-3.2, -3.2)}
    act = np.zeros((n, 12)); act[1, 1::3] = 2.0                       # robot 1: a little thigh torque, lands differently
    finals = {}
    for bc in (0, 1):
        cfg = A.default_config(n, solver_iters=4, motor_mode=1, body_contacts=bc, terrain=1, heightfield=hf, joint_limits=0)   # (a limp robot folds past the stops)
        orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
        for s in (orc, emu):
            s.set_heightfield(hf["heights"]); s.reset()
        for k in range(13):
            orc.step(act); emu.step(act)
            so, se = orc.get_state(), emu.get_state()
            if k < 12:   # (from step 12 on the limp legs flop chaotically: the fp32 oracle is 1e-3 rad off the fp64 one by then)
                assert np.abs(so[:, 13:25] - se[:, 13:25]).max() < 2e-3 and np.abs(so[:, :3] - se[:, :3]).max() < 1e-4, (bc, k)
        finals[bc] = orc.get_state()
    # without the knee rows the trunk keeps sinking through the floor (only feet collide); with them it is caught
    assert finals[0][0, 2] < -0.3 and finals[1][0, 2] > -0.2


@pytest.mark.parametrize("lanes", [16, 4])
def test_body_contacts_config_and_flat_ground_knee_rows(lanes):
    cfg = A.default_config(4, body_contacts=1)
    assert cfg.body_contacts == 1 and abs(cfg.knee_radius - 0.02) < 1e-12
    # the knee rows on the flat-ground instantiation (no heightfield): same limp-robot scenario as above
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 2
    act = np.zeros((n, 12)); act[1, 1::3] = 2.0
    cfg = A.default_config(n, solver_iters=4, motor_mode=1, body_contacts=1, joint_limits=0)
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    orc.reset(); emu.reset()
    for k in range(12):
        orc.step(act); emu.step(act)
        so, se = orc.get_state(), emu.get_state()
        assert np.abs(so[:, 13:25] - se[:, 13:25]).max() < 2e-3 and np.abs(so[:, :3] - se[:, :3]).max() < 1e-4, k
    assert orc.get_state()[0, 2] > -0.2


def _folded_drop_state(st):
    """legs folded up beside the trunk (thigh 2.5, calf -2.6: knees above the back, feet level with the belly), dropped from
    16 cm: the trunk's bottom corners reach the ground first.  Robot 1 lands across a 5 cm step edge."""
    st = st.copy()
    st[:, 2] = 0.16
    st[:, 13:25] = np.tile([0.0, 2.5, -2.6], 4)
    st[1, 13:25] = np.tile([0.3, 2.3, -2.5], 4); st[1, 0] = 0.15
    return st


def _step_edge_heightfield():
    hf = {"heights": np.zeros((64, 64), dtype=np.float32), "cell": 0.1, "origin": (-3.2, -3.2)}
    hf["heights"][:, 34:] = 0.05    # a 5 cm step ahead of x = 0.2
    return hf


@pytest.mark.parametrize("terrain", [0, 1])
def test_emulation_simultaneous_body_rows_match_oracle(terrain):
    """body_contacts = 3: knee, shin-midpoint and trunk-corner spheres of every leg collide AT ONCE (three frictionless rows per
    leg after its foot rows: 24 rows per robot in the oracle, six rows per lane with 6 x 6 Delassus blocks in the 4-lane kernel
    source, which is the mapping that serves this setting).  (a) the folded-legs belly landing: kernel source == oracle, the
    belly rests on the corner spheres; over 40 steps the trajectories stay closer than with the deepest-sphere row of
    body_contacts = 2, whose choice of sphere flips between ticks; (b) the limp standing robot is caught at the same height;
    (c) the two settings are different models (legs that rest on knee AND shin spheres end up elsewhere)."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 2
    hf = _step_edge_heightfield() if terrain else None
    rest = A.TRUNK_HALF[2] + 0.02
    finals = {}
    for bc in (2, 3):
        cfg = A.default_config(n, solver_iters=4, motor_mode=1, body_contacts=bc, terrain=terrain, heightfield=hf, joint_limits=0)
        orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=4)
        for s in (orc, emu):
            if terrain:
                s.set_heightfield(hf["heights"])
            s.reset()
        st = _folded_drop_state(orc.get_state())
        orc.set_state(st); emu.set_state(st)
        act = np.zeros((n, 12))
        for k in range(40):
            orc.step(act); emu.step(act)
            so, se = orc.get_state(), emu.get_state()
            if bc == 3:
                bound = (1e-4, 1e-5) if k < 12 else (3e-2, 3e-4)
                assert np.abs(so[:, 13:25] - se[:, 13:25]).max() < bound[0] and np.abs(so[:, :3] - se[:, :3]).max() < bound[1], k
        finals[bc] = so
        assert abs(so[0, 2] - rest) < 5e-4
    assert np.abs(finals[2][:, 13:25] - finals[3][:, 13:25]).max() > 0.05
    # (b) limp standing robot
    cfg = A.default_config(n, solver_iters=4, motor_mode=1, body_contacts=3, terrain=terrain, heightfield=hf, joint_limits=0)
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=4)
    for s in (orc, emu):
        if terrain:
            s.set_heightfield(hf["heights"])
        s.reset()
    act = np.zeros((n, 12)); act[1, 1::3] = 2.0
    low = np.full(n, 1.0)
    for k in range(30):
        orc.step(act); emu.step(act)
        so, se = orc.get_state(), emu.get_state()
        low = np.minimum(low, np.minimum(so[:, 2], se[:, 2]))
        if k < 12:
            assert np.abs(so[:, 13:25] - se[:, 13:25]).max() < 2e-4 and np.abs(so[:, :3] - se[:, :3]).max() < 1e-5, k
    assert low.min() > rest - 2e-3 and np.abs(so[:, :3] - se[:, :3]).max() < 5e-3
    # the residual stopping rule with 24 rows: a frozen robot does not move while its wave neighbours sweep on
    cfg = A.default_config(n, motor_mode=1, body_contacts=3, terrain=terrain, heightfield=hf, joint_limits=0)
    runs = []
    for extra in (0, 3):
        EmuSim.set_extra_sweeps(extra)
        try:
            emu = EmuSim(cfg, lanes=4)
            if terrain:
                emu.set_heightfield(hf["heights"])
            emu.reset()
            emu.set_state(_folded_drop_state(emu.get_state()))
            for k in range(15):
                emu.step(np.zeros((n, 12)))
            runs.append(emu.get_state())
        finally:
            EmuSim.set_extra_sweeps(0)
    assert np.array_equal(runs[0], runs[1])


@pytest.mark.parametrize("lanes", [16, 4])
@pytest.mark.parametrize("terrain", [0, 1])
def test_emulation_trunk_and_shin_contacts_match_oracle(terrain, lanes):
    """body_contacts = 2: the leg's 4th row takes the deepest of knee / shin midpoint / trunk corner.  (a) a limp robot with
    folded legs lands on its belly: the trunk corners carry it at half height + radius, kernel source == oracle; (b) the limp
    standing robot of the knee test is caught at the same height instead of sinking through the floor."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 2
    hf = _step_edge_heightfield() if terrain else None
    cfg = A.default_config(n, solver_iters=4, motor_mode=1, body_contacts=2, terrain=terrain, heightfield=hf, joint_limits=0)
    assert cfg.body_contacts == 2 and tuple(cfg.trunk_half) == A.TRUNK_HALF
    rest = A.TRUNK_HALF[2] + cfg.knee_radius
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    for s in (orc, emu):
        if terrain:
            s.set_heightfield(hf["heights"])
        s.reset()
    st = _folded_drop_state(orc.get_state())
    orc.set_state(st); emu.set_state(st)
    act = np.zeros((n, 12))
    for k in range(12):
        orc.step(act); emu.step(act)
        so, se = orc.get_state(), emu.get_state()
        # (the corner spheres grip now -- body_friction -- and the folded legs settle against them: from step 8 on the fp32
        # oracle itself is 2e-4 rad off the fp64 one)
        assert np.abs(so[:, 13:25] - se[:, 13:25]).max() < (1e-4 if k < 8 else 5e-4) and np.abs(so[:, :3] - se[:, :3]).max() < 1e-5, k
    assert abs(so[0, 2] - rest) < 5e-4                      # flat part: belly rests on its corner spheres
    if terrain:
        assert rest + 0.01 < so[1, 2] < rest + 0.05 + 1e-3  # across the step edge: the front corners sit on the step
    # (b) limp standing robot, 40 control steps: with the knee rows alone the trunk passes through the floor
    lows = {}
    for bc in (1, 2):
        cfg = A.default_config(n, solver_iters=4, motor_mode=1, body_contacts=bc, terrain=terrain, heightfield=hf, joint_limits=0)
        orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
        for s in (orc, emu):
            if terrain:
                s.set_heightfield(hf["heights"])
            s.reset()
        low = np.full(n, 1.0)
        for k in range(40):
            orc.step(act); emu.step(act)
            so, se = orc.get_state(), emu.get_state()
            low = np.minimum(low, np.minimum(so[:, 2], se[:, 2]))
            if k < 8:
                assert np.abs(so[:, 13:25] - se[:, 13:25]).max() < 1e-4 and np.abs(so[:, :3] - se[:, :3]).max() < 1e-5, (bc, k)
        lows[bc] = low
        if bc == 2:   # limp legs flop chaotically (fp32 vs fp64), the trunk does not
            assert np.abs(so[:, :3] - se[:, :3]).max() < 5e-3
    assert lows[1][0] < 0.0 and lows[2].min() > rest - 2e-3


@pytest.mark.parametrize("lanes", [4, 16])
def test_emulation_joint_limits_match_oracle(lanes):
    """EtgConfig.joint_limits (bounds of a1.py:186-195): constant torques drive the hip and knee joints into their stops
    (TORQUE mode).  The kernel source of both mappings against the oracle's model; the stops hold; and without the option
    the same torques run the joints far past the bounds."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 4
    act = np.zeros((n, 12))
    act[0, 0::3] = 6.0                       # abduction torque on every hip: upper hip bound 0.803
    act[1, 0::3] = -6.0                      # lower hip bound
    act[2, 2::3] = 8.0                       # knees extend: upper calf bound -0.916
    act[3, 2::3] = -8.0; act[3, 1::3] = 3.0  # knees fold: lower calf bound -2.697
    cfg = A.default_config(n, solver_iters=4, motor_mode=1, joint_limits=1, body_contacts=0)   # toe spheres only: the stops are the subject
    orc, emu = OracleSim(cfg), EmuSim(cfg, lanes=lanes)
    orc.reset(); emu.reset()
    worst = 0.0
    for k in range(12):
        orc.step(act); emu.step(act)
        so, se = orc.get_state(), emu.get_state()
        worst = max(worst, np.abs(so[:, 13:25] - se[:, 13:25]).max())
        assert np.abs(so[:, :3] - se[:, :3]).max() < 2e-3, k
        if k == 5:      # the splayed robot is about to sink into the floor (no body contacts here): look at the stops now
            q8 = so[:, 13:25].reshape(n, 4, 3)
    assert worst < 5e-3, worst
    q = orc.get_state()[:, 13:25].reshape(n, 4, 3)
    lo, hi = np.array(A.JOINT_LOWER), np.array(A.JOINT_UPPER)
    assert (q <= hi + 0.03).all() and (q >= lo - 0.03).all()          # the stops hold (Baumgarte leaves a small overshoot)
    assert np.abs(q8[0, :, 0] - hi[0]).max() < 0.03 and np.abs(q8[1, :, 0] - lo[0]).max() < 0.03   # and the hips sit on them
    free = OracleSim(A.default_config(n, solver_iters=4, motor_mode=1, joint_limits=0, body_contacts=0))
    free.reset()
    for k in range(12):
        free.step(act)
    qf = free.get_state()[:, 13:25].reshape(n, 4, 3)
    assert qf[0, :, 0].min() > hi[0] + 0.3                              # no stops: far past the bound


def test_param2dynamic_rows_torch_equals_numpy():
    import torch
    p = np.random.default_rng(3).uniform(-1.3, 1.3, size=(64, 48))
    a = A.param2dynamic_rows(p)
    b = A.param2dynamic_rows_torch(torch.as_tensor(p)).numpy()
    assert np.abs(a - b).max() < 1e-5


@pytest.mark.parametrize("lanes", [4, 16])
@pytest.mark.parametrize("option", ["strength", "strength_torque_mode", "clip_delayed", "restitution"])
def test_emulation_robot_layer_gaps_match_oracle(lanes, option):
    """Round-4 robot-layer options, kernel source (emulation, both mappings) against the oracle:
    * strength: per-motor strength ratios (laikago_motor.py:67-76,138,167; Minitaur.SetMotorStrengthRatios) with a torque limit
      -- the ratio acts before the clip; strength_torque_mode: ratio x commanded torque, no clip (laikago_motor.py:137-139);
    * clip_delayed: A1._ClipMotorCommands (a1.py:439-457) clips around GetMotorAngles(), the control-latency-delayed reading;
    * restitution: EtgConfig.foot_restitution on a robot dropped from 8 cm above its stance."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 3
    rng = np.random.default_rng(21)
    kw, strength = dict(settle_ticks=120, solver_iters=4), None
    if option == "strength":
        kw.update(torque_limit=12.0)
        strength = rng.uniform(0.4, 1.0, size=(n, 12))
    elif option == "strength_torque_mode":
        kw.update(motor_mode=1, torque_limit=2.0, body_contacts=0)   # (random torques fold the robot up: without body rows the collapse stays comparable)
        strength = rng.uniform(0.4, 1.0, size=(n, 12))
    elif option == "clip_delayed":
        kw.update(clip_motor_commands=0.05)
    else:
        kw.update(foot_restitution=0.6)
    cfg = A.default_config(n, **kw)
    # `plain`: the same run without the option (strength ratios 1 / no clip / no restitution)
    ref_cfg = A.default_config(n, **{k: v for k, v in kw.items() if k not in ("foot_restitution", "clip_motor_commands")})
    orc, emu, plain = OracleSim(cfg), EmuSim(cfg, lanes=lanes), OracleSim(ref_cfg)
    for s in (orc, emu, plain):
        if strength is not None and s is not plain:
            s.set_motor_strength(strength)
        s.reset()
    assert np.abs(emu.get_state() - orc.get_state())[:, 13:25].max() < 5e-5
    if option == "restitution":
        st = orc.get_state().copy()
        st[:, 2] += 0.08
        for s in (orc, emu, plain):
            s.set_state(st)
    worst = 0.0
    for k in range(8):
        if option == "strength_torque_mode":
            act = rng.uniform(-6.0, 6.0, size=(n, 12))
        else:
            act = rng.uniform(-0.4, 0.4, size=(n, 12)) if option == "clip_delayed" else rng.uniform(-0.15, 0.15, size=(n, 12))
        orc.step(act); emu.step(act); plain.step(act)
        worst = max(worst, np.abs(emu.get_state() - orc.get_state())[:, 13:25].max())
    assert worst < 3e-4, worst
    assert np.abs(plain.get_state() - orc.get_state())[:, :25].max() > 1e-3      # the option does change the motion


@pytest.mark.parametrize("lanes", [4, 16])
def test_a_robot_that_blew_up_does_not_take_the_heightfield_lookup_with_it(lanes):
    """A pd_latency of 6 ms makes the delayed damping term of the PD law unstable (kd * latency / link inertia > ~1): the settle
    blows up to non-finite numbers -- physics, not a defect, and the library only rejects latencies its ring cannot hold.  The
    heightfield lookup then gets a NaN coordinate: both the oracle and the kernel source must clamp it like max(NaN, 0) = 0 does
    (found by tools/fuzz_parity.py: the oracle's `if (fx < 0)` let the NaN through to the index), and the step reports the
    robot as terminated."""
    from oracle.oracle import OracleSim
    from tests.emu.emu import EmuSim
    n = 4
    hf = dict(heights=np.random.default_rng(3).uniform(0, 0.04, size=(64, 64)).astype(np.float32), cell=0.05, origin=(-1.6, -1.6))
    cfg = A.default_config(n, pd_latency=0.006, terrain=1, heightfield=hf)
    for sim in (OracleSim(cfg, dtype=np.float64), OracleSim(cfg, dtype=np.float32), EmuSim(cfg, lanes=lanes)):
        sim.set_heightfield(hf["heights"])
        sim.reset()
        assert not np.isfinite(sim.get_state()).all()
        _, _, done, _ = sim.step(np.zeros((n, 12), dtype=np.float32))
        assert np.asarray(done).astype(bool).all()


def test_emulated_body_paths_leave_a_robot_bit_identical():
    """The 16-lane tick takes its body-row paths when ANY robot of the wave has a body sphere inside the margin / a loaded body
    normal: a robot without such rows must come out bit-identical whichever path its wave takes, and a converged robot must sit
    through its neighbours' extra sweeps unchanged.  The emulation runs one robot per "wave", so the wave-uniform tests are forced
    (EmuSim.set_force_body) and extra sweeps appended (set_extra_sweeps): same states and rewards, bit for bit, on the skating
    gait of configs[1] (knee spheres inside the margin on half of the ticks, loaded on a tenth)."""
    from tests.emu.emu import EmuSim
    n = 6
    W, B = _params(n, seed=3)
    runs = {}
    try:
        for tag, fb, ex in (("plain", 0, 0), ("forced", 1, 0), ("forced+3", 1, 3)):
            EmuSim.set_force_body(fb); EmuSim.set_extra_sweeps(ex)
            emu = EmuSim(A.default_config(n), lanes=16)
            emu.set_params(etg_w=W, etg_b=B)
            emu.reset()
            rews = []
            for k in range(40):
                rews.append(emu.step(np.zeros((n, 12)))[1].copy())
            runs[tag] = (emu.get_state(), np.stack(rews))
    finally:
        EmuSim.set_force_body(0); EmuSim.set_extra_sweeps(0)
    for tag in ("forced", "forced+3"):
        assert np.array_equal(runs["plain"][0], runs[tag][0]) and np.array_equal(runs["plain"][1], runs[tag][1]), tag
