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


def emu_tick_rows(tr, i, t):
    """the emulation's / GPU debugging build's tick trace ([N,16,16,10]: physics_tick16's rowf, phi, lam, lam2, jactf, lamq, sphere
    code, sweeps so far, q, qd per lane) of robot i, tick t -> dict(act, loaded, pick, sweeps_cum, lam[36], phi_f, phi_b)"""
    x = tr[i, t]
    act, ld = 0, 0
    for leg in range(4):
        for sub in range(3):
            if x[4 * leg + sub, 0] > 0.5: act |= 1 << (3 * leg + sub)
        if x[4 * leg, 0] > 0.5 and x[4 * leg, 2] > 0: ld |= 1 << (3 * leg)
        if x[4 * leg + 3, 0] > 0.5:
            act |= 7 << (12 + 3 * leg)
            if x[4 * leg + 3, 2] > 0: ld |= 1 << (12 + 3 * leg)
        for sub in range(3):
            if x[4 * leg + sub, 4] > 0.5: act |= 1 << (24 + 3 * leg + sub)
    pick = 0
    for leg in range(4):        # w_shin + 2 w_trunk of the leg's body contact: 0 knee, 1 shin midpoint, 2 trunk corner (soft weights: rounded)
        pick |= min(max(int(round(float(x[4 * leg, 6]))), 0), 2) << (2 * leg)
    lam = np.zeros(36)
    for leg in range(4):
        for sub in range(3):
            lam[3 * leg + sub] = x[4 * leg + sub, 2]
            lam[24 + 3 * leg + sub] = x[4 * leg + sub, 5]
        lam[12 + 3 * leg] = x[4 * leg + 3, 2]
        lam[12 + 3 * leg + 1] = x[4 * leg + 1, 3]; lam[12 + 3 * leg + 2] = x[4 * leg + 2, 3]
    return dict(act=act, loaded=ld, pick=pick, sweeps_cum=int(x[0, 7]), lam=lam,
                phi_f=[float(x[4 * leg, 1]) for leg in range(4)], phi_b=[float(x[4 * leg + 3, 1]) for leg in range(4)])


def first_decision_gap(emu_tr, i, orc_tr, margin=0.02, sweeps=True):
    """the first physics tick of a control step at which the kernel source (emulation trace of robot i) and an oracle (trace_rows of
    the robot) DECIDE differently -> (tick, kind, text) or None.  kinds: "sphere" (another sphere of a leg's body contact is the
    deepest), "active" (a row inside the margin / a joint at its stop in one only), "loaded" (a normal row carries load in one
    only), "sweeps" (the residual test stopped one evaluation earlier: the emulated wave holds one robot, so the counts compare)"""
    prev = 0
    for t in range(min(13, len(orc_tr))):
        g = emu_tick_rows(emu_tr, i, t)
        o_act, o_ld, o_pick, o_sw = int(orc_tr[t, 0]), int(orc_tr[t, 1]) & 0b001001001001001001001001, int(orc_tr[t, 2]), int(orc_tr[t, 3])
        sw = g["sweeps_cum"] - prev
        prev = g["sweeps_cum"]
        for leg in range(4):
            if (o_act >> (12 + 3 * leg)) & 1 and (g["act"] >> (12 + 3 * leg)) & 1 and ((g["pick"] >> (2 * leg)) & 3) != ((o_pick >> (2 * leg)) & 3):
                names = ("knee", "shin midpoint", "trunk corner")
                return t, "sphere", "tick %d: the body contact of leg %s sits on the %s in the kernel source and on the %s in the oracle (contact distance %.3e / %.3e)" % (
                    t, ("FR", "FL", "RR", "RL")[leg], names[(g["pick"] >> (2 * leg)) & 3], names[(o_pick >> (2 * leg)) & 3], g["phi_b"][leg], orc_tr[t, 8 + leg])
        if g["act"] != o_act:
            rows = _bits(g["act"] ^ o_act)
            return t, "active", "tick %d: rows active in one evaluation only: %s (distances %s / %s, margin %.3g)" % (
                t, [ROW_NAMES[r] for r in rows], ["%.6e" % x for x in g["phi_f"] + g["phi_b"]], ["%.6e" % x for x in orc_tr[t, 4:12]], margin)
        if g["loaded"] != o_ld:
            rows = _bits(g["loaded"] ^ o_ld)
            return t, "loaded", "tick %d: normal rows loaded in one evaluation only: %s" % (t, [ROW_NAMES[r] for r in rows])
        if sweeps and sw != o_sw:
            return t, "sweeps", "tick %d: %d sweeps in the kernel source, %d in the oracle" % (t, sw, o_sw)
    return None


def oracle_pick_changes(orc_tr):
    """an oracle's tick trace of one control step -> [(tick, leg)] at which the sphere an ACTIVE body contact sits on changes
    from the tick before (hard deepest-of-three choice: the robot crosses a tie of two spheres within this step)"""
    out = []
    for t in range(1, len(orc_tr)):
        for leg in range(4):
            both = (int(orc_tr[t, 0]) >> (12 + 3 * leg)) & (int(orc_tr[t - 1, 0]) >> (12 + 3 * leg)) & 1
            if both and ((int(orc_tr[t, 2]) ^ int(orc_tr[t - 1, 2])) >> (2 * leg)) & 3:
                out.append((t, leg))
    return out


def lockstep_offenders(run, probe, ens, emu, steps, action_fn, floor_q=2e-5, floor_p=1e-5, say=print, what=""):
    """The one-step consistency run of tests/test_gpu_parity5.py with the classification of every (robot, step) pair that is off
    every branch of the oracle ensemble.  run / probe: two GPU envs in the same state (probe repeats every step of run from run's
    own state + contact impulses); ens: OracleEnsemble, emu: EmuSim of the same configuration with set_trace() on (16-lane
    mapping), all reset and ready.  For an offending pair the emulation -- the kernel SOURCE on the host: plain C++, fp32, libm, no
    FMA contraction -- is asked: if it is within the floor of the GPU, the GPU computes what its source says, and the tick at which
    that source first decides differently from the fp64 oracle names the cause.
    -> (tally dict, [dict(step, robot, gap, emu_gap, kind, text)])"""
    import torch
    n = run.num_envs
    etr = emu.set_trace(True)
    # on a tracing build of the library (tools/build_variant.sh trace -DETG_TRACE_TICKS; ETG_LIB=...) the GPU's own tick trace of
    # the probe's step is read as well: the first differing decision is then the GPU's, not inferred through the emulation
    gtr = None
    lib = getattr(probe, "_lib", None)
    if lib is not None and hasattr(lib, "etg_debug_set_trace") and getattr(probe, "lanes_per_robot", 0) == 16:
        import ctypes as C
        gtr_t = torch.zeros(n, 16, 16, 10, device=probe.device)
        lib.etg_debug_set_trace.argtypes = [C.c_void_p, C.c_void_p]
        assert lib.etg_debug_set_trace(probe._h, C.c_void_p(gtr_t.data_ptr())) == 0
        gtr = gtr_t
    tally = dict(pairs=0, nominal=0, other=0, none=0)
    out = []
    for k in range(steps):
        act = np.asarray(action_fn(k), dtype=np.float32)
        st, lam = run.get_state(), run.get_contact_impulses()
        stn, lamn = st.cpu().numpy(), lam.cpu().numpy()
        probe.set_state(st); probe.set_contact_impulses(lam)
        ens.set_state(stn.astype(np.float64)); ens.set_lambda(lamn.astype(np.float64))
        emu.set_state(stn); emu.set_contact_impulses(lamn)
        a = torch.as_tensor(act)
        probe.step(a, want_info=False)
        otr = ens.nominal.trace_all(16)                  # the fp64 oracle's tick trace of this step, every robot
        ens.step(act, want_info=False)
        ocnt = ens.nominal.trace_counts()
        emu.step(act)
        run.step(a, want_info=False)
        sg = probe.get_state().cpu().numpy().astype(np.float64)
        so, mem = ens.get_state(), ens.member_states()
        se = emu.get_state().astype(np.float64)
        from tests.parity_util import one_step_verdict
        dq = np.stack([np.abs(sg - m)[:, 13:25].max(1) for m in [so] + mem])
        nq, aq, _, _ = one_step_verdict(sg[:, 13:25], so[:, 13:25], [m[:, 13:25] for m in mem], floor_q)
        npz, ap, _, _ = one_step_verdict(sg[:, :7], so[:, :7], [m[:, :7] for m in mem], floor_p)
        on_nom, on_any = nq & npz, aq & ap
        tally["pairs"] += n
        tally["nominal"] += int(on_nom.sum())
        tally["other"] += int((on_any & ~on_nom).sum())
        bad = np.nonzero(~on_any)[0]
        tally["none"] += len(bad)
        for i in bad:
            emu_gap = float(np.abs(se[i] - sg[i])[13:25].max())
            kind, text = "?", "the emulation does not reproduce the GPU's result (%.2e): not classified" % emu_gap
            if emu_gap <= 10 * floor_q:
                d = first_decision_gap(etr, int(i), otr[i, :ocnt[i]], float(ens.cfg.contact_margin))
                kind, text = (d[1], d[2]) if d else ("smooth", "no tick with a differing decision: a smooth gap")
            if gtr is not None:                          # the GPU's own trace (its sweep counts are the WAVE's: not compared)
                d = first_decision_gap(gtr.cpu().numpy(), int(i), otr[i, :ocnt[i]], float(ens.cfg.contact_margin), sweeps=False)
                kind, text = (d[1], "GPU trace: " + d[2]) if d else (kind, text + "; GPU trace: no tick with a differing row set")
                per = [emu_tick_rows(gtr.cpu().numpy(), int(i), t)["sweeps_cum"] for t in range(13)]
                text += "; wave sweeps per tick on the GPU %s, oracle (this robot) %s" % (list(np.diff([0] + per)), [int(x) for x in otr[i, :ocnt[i], 3]])
            ties = oracle_pick_changes(otr[i, :ocnt[i]])
            if ties:
                text += "; in the fp64 oracle's own step the sphere under the body contact changes at (tick, leg) %s" % ties[:4]
            out.append(dict(step=k, robot=int(i), gap=float(dq[0, i]), nearest=float(dq[:, i].min()), emu_gap=emu_gap, kind=kind, text=text,
                            tie_in_step=bool(ties)))
            say("[parity]    %s step %d robot %d: GPU %.2e off the fp64 oracle (nearest member %.2e); the kernel source on the host is %.2e from the GPU; %s"
                % (what, k, i, dq[0, i], dq[:, i].min(), emu_gap, text))
    say("[parity] %-50s (robot, step) pairs %d: on the fp64 oracle's branch %d, on another member's %d, on none %d %s"
        % (what, tally["pairs"], tally["nominal"], tally["other"], tally["none"],
           ("(first differing decision: %s)" % dict((k_, sum(1 for o_ in out if o_["kind"] == k_)) for k_ in sorted(set(o_["kind"] for o_ in out)))) if out else ""))
    return tally, out


def first_jump(gaps, floor):
    """gaps [steps] of one robot -> index of the first step whose gap exceeds the floor and 10 x the step before, or None"""
    prev = 0.0
    for k, g in enumerate(gaps):
        if g > floor and g > 10.0 * max(prev, 1e-7):
            return k
        prev = g
    return None
