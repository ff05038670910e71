"""Where does a GPU control step part from the oracle's?  (one MI355X; a debugging build of the library)

  tools/build_variant.sh lib_trace -DETG_TRACE_TICKS
  ETG_EMU_FLAGS=-DETG_TRACE_TICKS python -c "from tests.emu.emu import lib; lib()"
  ETG_LIB=gpurun_variants/lib_trace.so ETG_EMU_FLAGS=-DETG_TRACE_TICKS python tools/first_divergence.py [--case flat16] [--n 96] [--steps 30]

Runs the one-step-consistency scenario of tests/test_gpu_parity5.py: at every control step the GPU's own state (etg_get_state +
etg_get_contact_impulses) is installed in a probe env, in the oracle ensemble and in the HOST EMULATION of the kernel source
(tests/emu: the same C++ in fp32 with libm and without FMA contraction), and all of them take the step with a per-tick trace:
which rows were active, which sphere of a leg was the body contact, how many sweeps ran, the impulses, the joint angles.  For
every (robot, step) whose GPU result is further than the floor from every oracle member it prints the first tick at which
the GPU's DISCRETE decisions (active rows, sphere pick, joint stops, sweep count) differ from the fp64 oracle's / the
emulation's, with the quantities the decision hung on; if no decision differs in any tick, the tick at which the impulses first
part and by how much.  Cases: flat16, flat4 (4-lane mapping: GPU trace absent, oracle-only report), dyn16 (random dynamics).
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from paddlerobotics_amd import a1_model as A
from paddlerobotics_amd import _lib
from tests.test_gpu_parity import _etg_params, _make
from tests.parity_util import OracleEnsemble
from tests.divergence import ROW_NAMES
from tests.emu.emu import EmuSim

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="flat16")
ap.add_argument("--n", type=int, default=96)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--floor", type=float, default=2e-5)
ap.add_argument("--max-reports", type=int, default=12)
ap.add_argument("--blend", type=float, default=None, help="EtgConfig.body_blend (default: the library's; 0 = round 5's hard choice of the deepest sphere)")
args = ap.parse_args()
n, lanes = args.n, 4 if args.case.endswith("4") else 16
seed = {"flat16": 61 + 16, "flat4": 61 + 4, "dyn16": 77}[args.case]
amp = 0.2 if args.case == "dyn16" else 0.15
dyn = None
if args.case == "dyn16":
    rng0 = np.random.default_rng(9)
    dyn = np.stack([A.dynamic_dict_to_row(A.param2dynamic_dict(rng0.uniform(-0.3, 0.3, 48))) for _ in range(n)]).astype(np.float32).astype(np.float64)

W, B = _etg_params(n, seed=seed)
bkw = {} if args.blend is None else dict(body_blend=args.blend)
run, probe = _make(n, lanes_per_robot=lanes, **bkw), _make(n, lanes_per_robot=lanes, **bkw)
ens = OracleEnsemble(n, E=3, E64=1, seed=seed, **bkw)
emu = EmuSim(A.default_config(n, **bkw), lanes=lanes)
if dyn is not None:
    ens.set_params(dyn=dyn); emu.set_params(dyn=dyn)
    for e in (run, probe):
        e.set_dynamic_param(torch.as_tensor(dyn, dtype=torch.float32, device="cuda:0"))
for e in (run, probe):
    e.reset(ETG_w=W, ETG_b=B)
ens.set_params(etg_w=W, etg_b=B); ens.reset()
emu.set_params(etg_w=W, etg_b=B); emu.reset()
lib = _lib.load()
have_trace = hasattr(lib, "etg_debug_set_trace") and lanes == 16
gtrace = torch.zeros(n, 16, 16, 10, device="cuda:0")
if have_trace:
    lib.etg_debug_set_trace.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.etg_debug_set_trace(probe._h, C.c_void_p(gtrace.data_ptr())) == 0
etrace = emu.set_trace(True) if lanes == 16 else None
print("case %s: %d robots x %d steps, lanes %d, body_blend %g, GPU tick trace %s" % (args.case, n, args.steps, lanes, run.cfg.body_blend, "on" if have_trace else "OFF (build lib_trace and set ETG_LIB)"))


def gpu_rows(tr, i, t):
    """[16 lanes, 10] of robot i, tick t -> dict in the oracle trace's terms"""
    x = tr[i, t]
    act, ld = 0, 0
    for leg in range(4):
        for sub in range(3):
            if x[4 * leg + sub, 0] > 0.5: act |= 1 << (3 * leg + sub)
            if x[4 * leg + sub, 0] > 0.5 and sub == 0 and x[4 * leg, 2] > 0: ld |= 1 << (3 * leg)
        if x[4 * leg + 3, 0] > 0.5:
            act |= 7 << (12 + 3 * leg)
            if x[4 * leg + 3, 2] > 0: ld |= 1 << (12 + 3 * leg)
        for sub in range(3):
            if x[4 * leg + sub, 4] > 0.5: act |= 1 << (24 + 3 * leg + sub)
    pick = 0
    for leg in range(4):
        code = int(round(float(x[4 * leg, 6])))          # w_shin + 2 w_trunk: 0 knee, 1 shin midpoint, 2 trunk corner (soft weights: the heaviest)
        pick |= min(max(code, 0), 2) << (2 * leg)
    lam = np.zeros(36)
    for leg in range(4):
        for sub in range(3):
            lam[3 * leg + sub] = x[4 * leg + sub, 2]
            lam[24 + 3 * leg + sub] = x[4 * leg + sub, 5]
        lam[12 + 3 * leg] = x[4 * leg + 3, 2]
        lam[12 + 3 * leg + 1] = x[4 * leg + 1, 3]; lam[12 + 3 * leg + 2] = x[4 * leg + 2, 3]
    phi_f = [float(x[4 * leg, 1]) for leg in range(4)]
    phi_b = [float(x[4 * leg + 3, 1]) for leg in range(4)]
    q = np.array([x[4 * leg + sub, 8] for leg in range(4) for sub in range(3)])
    return dict(act=act, loaded=ld, pick=pick, sweeps_cum=int(x[0, 7]), lam=lam, phi_f=phi_f, phi_b=phi_b, q=q)


def orc_rows(tr, t):
    x = tr[t]
    act = int(x[0])
    # the oracle marks all three rows of an active body contact; joint rows at 24..35
    return dict(act=act, loaded=int(x[1]) & 0b001001001001001001001001, pick=int(x[2]), sweeps=int(x[3]), lam=x[12:48].copy(),
                phi_f=x[4:8].tolist(), phi_b=x[8:12].tolist(), q_after=x[48:60].copy())


def names(mask):
    return [ROW_NAMES[r] for r in range(36) if (mask >> r) & 1]


rng = np.random.default_rng(seed + 100)
reports = 0
tally = dict(pairs=0, nominal=0, other=0, none=0, emu_with_gpu=0, emu_with_orc=0)
for k in range(args.steps):
    act = rng.uniform(-amp, amp, size=(n, 12)).astype(np.float32)
    st, lam = run.get_state(), run.get_contact_impulses()
    probe.set_state(st); probe.set_contact_impulses(lam)
    stn, lamn = st.cpu().numpy(), lam.cpu().numpy()
    ens.set_state(stn.astype(np.float64)); ens.set_lambda(lamn.astype(np.float64))
    emu.set_state(stn); emu.set_contact_impulses(lamn)
    for o in (ens.nominal, ens.o32):
        o._tr = []
    # oracle traces: one buffer per robot on the fp64 and the fp32 oracle
    otr = {}
    a = torch.as_tensor(act)
    gtrace.zero_()
    _, _, _, info = probe.step(a)
    # (the oracle's trace API records one robot per call: step the ensemble once per traced robot would be slow -- trace lazily below)
    ens.step(act)
    emu.step(act)
    run.step(a, want_info=False)
    sg = probe.get_state().cpu().numpy().astype(np.float64)
    so = ens.get_state()
    mem = ens.member_states()
    se = emu.get_state().astype(np.float64)
    d_all = np.stack([np.abs(sg - m)[:, 13:25].max(1) for m in [so] + mem])
    d_emu_g = np.abs(se - sg)[:, 13:25].max(1)
    d_emu_o = np.abs(se - so)[:, 13:25].max(1)
    on_nom = d_all[0] <= args.floor
    on_any = d_all.min(0) <= args.floor
    tally["pairs"] += n; tally["nominal"] += int(on_nom.sum()); tally["other"] += int((on_any & ~on_nom).sum()); tally["none"] += int((~on_any).sum())
    bad = np.nonzero(~on_nom)[0]
    for i in bad:
        tally["emu_with_gpu"] += int(d_emu_g[i] <= args.floor)
        tally["emu_with_orc"] += int(d_emu_o[i] <= args.floor)
        if reports >= args.max_reports:
            continue
        reports += 1
        print("\n=== step %d robot %d: GPU vs fp64 oracle %.2e | vs members %s | emulation vs GPU %.2e, emulation vs fp64 oracle %.2e | wave sweeps (GPU, step) %d"
              % (k, i, d_all[0, i], ["%.1e" % x for x in d_all[1:, i]], d_emu_g[i], d_emu_o[i], int(info["solver_sweeps"].cpu().numpy().reshape(-1)[i])))
        # re-run the step on single-robot copies of the fp64 / fp32 oracle with the trace on: the ensemble has moved on, so re-create
        # it from the stored state (the oracle's step counter = k after k steps: use fresh oracles advanced by k zero-action steps)
        from oracle.oracle import OracleSim
        tro = {}
        for nm, dt in (("fp64", np.float64), ("fp32", np.float32)):
            o = OracleSim(A.default_config(n, **bkw), dtype=dt)
            o.threads = os.cpu_count() or 1
            if dyn is not None:
                o.set_params(dyn=dyn)
            o.set_params(etg_w=W, etg_b=B); o.reset()
            for _ in range(k):
                o.step(np.zeros((n, 12)), want_info=False)
            o.set_state(stn.astype(np.float64)); o.set_lambda(lamn.astype(np.float64))
            o.trace(int(i), 16)
            o.step(act.astype(np.float64), want_info=False)
            tro[nm] = o.trace_rows()
            chk = np.abs(np.asarray(o.get_state(), dtype=np.float64)[i] - (so if nm == "fp64" else mem[0])[i])[13:25].max()
            if chk > 1e-9:
                print("    (note: the re-created %s oracle step differs from the ensemble's by %.1e)" % (nm, chk))
        gt = gtrace.cpu().numpy() if have_trace else None
        et = etrace if etrace is not None else None
        first = None
        prev_cum_g = prev_cum_e = 0
        for t in range(13):
            o64, o32 = orc_rows(tro["fp64"], t), orc_rows(tro["fp32"], t)
            line = "    tick %2d: fp64 sweeps %2d active %s pick %s | fp32 sweeps %2d" % (t, o64["sweeps"], bin(o64["act"] & 0xFFFFFF), format(o64["pick"], "08b"), o32["sweeps"])
            diffs = []
            for nm, tr in (("GPU", gt), ("emu", et)):
                if tr is None:
                    continue
                g = gpu_rows(tr, i, t)
                prev = prev_cum_g if nm == "GPU" else prev_cum_e
                sw = g["sweeps_cum"] - prev
                if nm == "GPU": prev_cum_g = g["sweeps_cum"]
                else: prev_cum_e = g["sweeps_cum"]
                line += " | %s sweeps(wave) %2d" % (nm, sw)
                if g["act"] != o64["act"]:
                    diffs.append("%s ACTIVE rows differ: %s" % (nm, names(g["act"] ^ o64["act"])))
                    for r in [r for r in range(24) if ((g["act"] ^ o64["act"]) >> r) & 1]:
                        if r < 12:
                            diffs.append("   %s phi %s %.9e fp64 %.9e (margin %.3e)" % (ROW_NAMES[r], nm, g["phi_f"][r // 3], o64["phi_f"][r // 3], 0.02))
                        elif (r - 12) % 3 == 0:
                            diffs.append("   %s phi %s %.9e fp64 %.9e" % (ROW_NAMES[r], nm, g["phi_b"][(r - 12) // 3], o64["phi_b"][(r - 12) // 3]))
                # the sphere pick only matters for legs whose body contact is active
                for leg in range(4):
                    if (o64["act"] >> (12 + 3 * leg)) & 1 and ((g["pick"] >> (2 * leg)) & 3) != ((o64["pick"] >> (2 * leg)) & 3):
                        diffs.append("%s body sphere of leg %d: %d, fp64 %d" % (nm, leg, (g["pick"] >> (2 * leg)) & 3, (o64["pick"] >> (2 * leg)) & 3))
                if g["loaded"] != (o64["loaded"]):
                    diffs.append("%s LOADED normal rows differ: %s" % (nm, names(g["loaded"] ^ o64["loaded"])))
                dl = np.abs(g["lam"][:24] - o64["lam"][:24])
                r = int(dl.argmax())
                line += " max |d impulse| %.1e (%s: %.4e vs %.4e)" % (dl[r], ROW_NAMES[r], g["lam"][r], o64["lam"][r])
            if o32["act"] != o64["act"] or o32["sweeps"] != o64["sweeps"]:
                diffs.append("fp32 oracle differs from fp64: active %s sweeps %d/%d" % (names(o32["act"] ^ o64["act"]), o32["sweeps"], o64["sweeps"]))
            print(line)
            for d in diffs:
                print("        " + d)
            if diffs and first is None:
                first = t
        print("    -> first tick with a differing decision: %s" % ("none: a smooth divergence" if first is None else first))
print("\n%s: (robot, step) pairs %d: on the fp64 oracle's branch %d, on another member's %d, on none %d | of the pairs off the fp64 branch the "
      "emulation sided with the GPU %d times, with the fp64 oracle %d times" % (args.case, tally["pairs"], tally["nominal"], tally["other"],
                                                                                  tally["none"], tally["emu_with_gpu"], tally["emu_with_orc"]))
