"""Which robots make the slow ticks under the reference's randomisation (train.py:117: param2dynamic_dict(U(-0.3, 0.3)), actions
+-0.3 rad)?  The oracle, per robot and control step: sweeps, body rows loaded, joints at a stop -- CPU only.
usage: python tools/tail_probe.py [body_contacts=2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from paddlerobotics_amd import a1_model as A
from oracle.oracle import OracleSim
import bench
bc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, steps = 512, 60
rng = np.random.default_rng(3)
dyn = A.param2dynamic_rows(rng.uniform(-0.3, 0.3, size=(n, 48)))
w, b = bench.etg_population(n, 0, "cpu")
sim = OracleSim(A.default_config(n, body_contacts=bc), threads=16)
sim.set_params(dyn=dyn, etg_w=w.double().numpy(), etg_b=b.double().numpy())
sim.reset(); sim.body_stats()
lo, hi = np.array(A.JOINT_LOWER * 4), np.array(A.JOINT_UPPER * 4)
rec = []
alive = np.ones(n, bool)
for k in range(steps):
    act = rng.uniform(-0.3, 0.3, size=(n, 12))
    _, _, done, info = sim.step(act)
    sw = info[:, A.INFO_SWEEPS] / 13.0
    bs = sim.body_stats()
    q = sim.get_state()[:, 13:25]
    stop = ((q >= hi - 1e-9) | (q <= lo + 1e-9)).sum(1)
    rec.append(np.stack([sw, bs[:, 1] / 13.0, stop, alive.astype(float)], 1))
    alive &= ~done.astype(bool)
R = np.stack(rec)          # [steps, n, 4]
sw, loaded, stop, al = R[..., 0], R[..., 1], R[..., 2], R[..., 3]
mu = dyn[:, 1]
print("body_contacts %d | %d robots x %d steps, random dynamics U(-0.3, 0.3), actions +-0.3 rad" % (bc, n, steps))
print("sweeps per tick: mean %.2f  p90 %.1f  p99 %.1f  max %.1f ; robot-steps with >= 20 sweeps per tick: %.3f, at the cap (>= 45): %.3f" % (
    sw.mean(), np.percentile(sw, 90), np.percentile(sw, 99), sw.max(), (sw >= 20).mean(), (sw >= 45).mean()))
slow = sw >= 20
for name, m in (("all robot-steps", np.ones_like(slow)), ("slow (>= 20 sweeps/tick)", slow)):
    print("  %-26s share with a loaded body row %.2f | with a joint at a stop %.2f | terminated earlier %.2f | mean foot mu %.2f" % (
        name, (loaded[m] > 0).mean(), (stop[m] > 0).mean(), (al[m] < 0.5).mean(), np.broadcast_to(mu, sw.shape)[m].mean()))
for lo_, hi_ in ((0, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 3.3)):
    m = (mu >= lo_) & (mu < hi_)
    if m.any():
        print("  foot mu in [%.1f, %.1f): %3d robots, sweeps per tick mean %.2f p99 %.1f, slow share %.3f" % (lo_, hi_, m.sum(), sw[:, m].mean(), np.percentile(sw[:, m], 99), slow[:, m].mean()))
# neither a body row nor a stop: what is left?
rest = slow & (loaded == 0) & (stop == 0)
print("  slow robot-steps with NEITHER a loaded body row NOR a joint at a stop: %.3f of the slow ones (mean mu %.2f)" % (rest.sum() / max(slow.sum(), 1), np.broadcast_to(mu, sw.shape)[rest].mean() if rest.any() else float('nan')))
