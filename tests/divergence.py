"""First-divergence analysis of one robot (test infrastructure; used by tests/test_gpu_regressions.py and tools/first_divergence.py).

Question: a robot's GPU trajectory is 1e-3 .. 1e-2 rad off the fp64 oracle's after k control steps -- a defect, or a contact
that switched on one tick earlier?  Answer, per robot:

 1. trajectory: the first control step k* at which the GPU-vs-fp64 gap jumps (by more than 10x, beyond the floor);
 2. ONE-STEP analysis at k*: the GPU's state BEFORE the step (fp32, through the C-ABI) is installed in a probe env and in an
    ensemble of oracles (fp64, fp32, fp64 with the action moved by +-1 fp32 ulp -- and, for the ring-dependent options, nothing
    else differs: etg_set_state re-seeds the latency ring on both sides); all take the step; the GPU's result is compared
    with every member's.  "On a branch" = within the floor of some member;
 3. tick trace: the oracle members record, per physics tick of that step, which rows are active (inside the margin / joint at a
    stop), which carry load, the distances phi, the impulses and the sweep count (Env::trace).  The report names the first tick
    at which the member nearest to the GPU and the fp64 oracle differ in their ACTIVE or LOADED row sets, the row, and how close
    its distance was to the activation threshold.

Everything here needs the GPU except `compare_traces`."""
import numpy as np

ROW_NAMES = (["foot %s %s" % (leg, r) for leg in ("FR", "FL", "RR", "RL") for r in ("n", "t1", "t2")] +
             ["body %s %s" % (leg, r) for leg in ("FR", "FL", "RR", "RL") for r in ("n", "t1", "t2")] +
             ["joint stop %s %s" % (leg, j) for leg in ("FR", "FL", "RR", "RL") for j in ("hip", "thigh", "calf")])


def _bits(x):
    return [r for r in range(36) if (int(x) >> r) & 1]


def compare_traces(ta, tb, margin):
    """two tick traces [ticks, 64] of one control step -> dict(tick, kind, rows, detail) of the first tick at which the ACTIVE
    or LOADED row sets differ, or None"""
    for t in range(min(len(ta), len(tb))):
        a_act, b_act = int(ta[t, 0]), int(tb[t, 0])
        a_ld, b_ld = int(ta[t, 1]), int(tb[t, 1])
        if a_act != b_act:
            rows = _bits(a_act ^ b_act)
            det = []
            for r in rows:
                if r < 12:
                    det.append("%s: phi %.3e / %.3e (margin %.3e)" % (ROW_NAMES[r], ta[t, 4 + r // 3], tb[t, 4 + r // 3], margin))
                elif r < 24:
                    l = (r - 12) // 3
                    det.append("%s: phi %.3e / %.3e (margin %.3e), sphere picked %d / %d" % (
                        ROW_NAMES[r], ta[t, 8 + l], tb[t, 8 + l], margin, (int(ta[t, 2]) >> (2 * l)) & 3, (int(tb[t, 2]) >> (2 * l)) & 3))
                else:
                    det.append("%s (joint at its bound in one evaluation only)" % ROW_NAMES[r])
            return dict(tick=t, kind="row set inside the margin differs", rows=rows, detail=det,
                        sweeps=(int(ta[t, 3]), int(tb[t, 3])))
        if a_ld != b_ld:
            rows = _bits(a_ld ^ b_ld)
            det = ["%s: impulse %.3e / %.3e" % (ROW_NAMES[r], ta[t, 12 + r], tb[t, 12 + r]) for r in rows]
            return dict(tick=t, kind="set of rows that carry load differs", rows=rows, detail=det, sweeps=(int(ta[t, 3]), int(tb[t, 3])))
        if int(ta[t, 2]) != int(tb[t, 2]):
            legs = [l for l in range(4) if ((int(ta[t, 2]) ^ int(tb[t, 2])) >> (2 * l)) & 3]
            det = ["leg %d: deepest sphere %d / %d, distance %.3e / %.3e" % (l, (int(ta[t, 2]) >> (2 * l)) & 3, (int(tb[t, 2]) >> (2 * l)) & 3,
                                                                              ta[t, 8 + l], tb[t, 8 + l]) for l in legs]
            if any((a_act >> (12 + 3 * l)) & 1 for l in legs):
                return dict(tick=t, kind="deepest body sphere differs", rows=[12 + 3 * l for l in legs], detail=det, sweeps=(int(ta[t, 3]), int(tb[t, 3])))
        if int(ta[t, 3]) != int(tb[t, 3]):
            dl = np.abs(ta[t, 12:48] - tb[t, 12:48])
            r = int(dl.argmax())
            return dict(tick=t, kind="sweep count differs (the residual test stopped one evaluation earlier)", rows=[r],
                        detail=["largest impulse gap after the tick: %s %.3e / %.3e" % (ROW_NAMES[r], ta[t, 12 + r], tb[t, 12 + r])],
                        sweeps=(int(ta[t, 3]), int(tb[t, 3])))
    return None


def one_step_report(make_env_fn, make_ens_fn, install, state_before, action, robot, steps_taken, floor_q=2e-5, say=print):
    """The one-step analysis of `robot` from `state_before` [N,37] (fp32) with `action` [N,adim].  make_env_fn() -> a fresh env
    (same configuration as the trajectory's), make_ens_fn() -> OracleEnsemble, install(env, ens): parameters + reset on both.
    steps_taken: control steps the trajectory had taken before this one (the ETG phase): probe and members take that many steps
    from their own reset first, then get the state.  -> dict(nearest, dist, gaps, divergence)"""
    import torch
    probe = make_env_fn()
    ens = make_ens_fn()
    install(probe, ens)
    n = probe.num_envs
    adim = action.shape[1]
    zero = np.zeros((n, adim), dtype=np.float32)
    for _ in range(steps_taken):      # advance the step counters (ETG phase) to the trajectory's
        probe.step(torch.as_tensor(zero), want_info=False)
        ens.step(zero, want_info=False)
    st = torch.as_tensor(state_before, dtype=torch.float32)
    probe.set_state(st)
    ens.set_state(np.asarray(state_before, dtype=np.float64))
    members = [ens.nominal] + ens.members
    names = ["fp64 oracle", "fp32 oracle"] + ["fp64 oracle, action +-1 ulp (%d)" % i for i in range(len(ens.nudged))]
    for o in members:
        o.trace(robot, 64)
    _, _, _, info = probe.step(torch.as_tensor(action, dtype=torch.float32))
    ens.step(action)
    traces = [o.trace_rows() for o in members]
    sg = probe.get_state().cpu().numpy().astype(np.float64)[robot]
    states = [np.asarray(o.get_state(), dtype=np.float64)[robot] for o in members]
    dq = [float(np.abs(sg[13:25] - s[13:25]).max()) for s in states]
    gaps = [float(np.abs(s[13:25] - states[0][13:25]).max()) for s in states]
    near = int(np.argmin(dq))
    sweeps_gpu = int(info["solver_sweeps"].cpu().numpy().reshape(-1)[robot]) if "solver_sweeps" in info else -1
    say("    one control step from the GPU's own state, robot %d: GPU joints vs each member (member's own gap to the fp64 oracle):" % robot)
    for nm, d, g, tr in zip(names, dq, gaps, traces):
        say("      %-38s %.2e (%.2e)  sweeps in the step %d" % (nm, d, g, int(tr[:, 3].sum())))
    say("      GPU: sweeps its wave executed in the step %d" % sweeps_gpu)
    out = dict(nearest=names[near], dist=dq[near], dist_nominal=dq[0], gaps=gaps, divergence=None, on_branch=dq[near] <= floor_q)
    margin = float(ens.cfg.contact_margin)
    if near != 0 and dq[0] > floor_q:
        div = compare_traces(traces[0], traces[near], margin)
        out["divergence"] = div
        if div:
            say("      the nearest member (%s) parts from the fp64 oracle at tick %d of 13: %s -- %s; sweeps of that tick %d / %d"
                % (names[near], div["tick"], div["kind"], "; ".join(div["detail"]), div["sweeps"][0], div["sweeps"][1]))
        else:
            say("      the nearest member (%s) has the same active and loaded row sets as the fp64 oracle in every tick: a smooth gap" % names[near])
    elif dq[0] <= floor_q:
        say("      the GPU is on the fp64 oracle's own branch (%.2e <= %.1e)" % (dq[0], floor_q))
    probe.close()
    return out


def first_jump(gaps, floor):
    """gaps [steps] of one robot -> index of the first step whose gap exceeds the floor and 10 x the step before, or None"""
    prev = 0.0
    for k, g in enumerate(gaps):
        if g > floor and g > 10.0 * max(prev, 1e-7):
            return k
        prev = g
    return None
