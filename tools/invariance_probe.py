"""Which robots change (bitwise) when they share their wavefront with other neighbours?  (debug aid for
test_wave_neighbours_do_not_influence_a_robot)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from paddlerobotics_amd import a1_model as A
from paddlerobotics_amd.env import make_env
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.test_gpu_parity3 import _etg_params
n = 256
W, B = _etg_params(n, seed=23)
rng = np.random.default_rng(5)
dyn = np.stack([A.dynamic_dict_to_row(A.param2dynamic_dict(rng.uniform(-0.3, 0.3, 48))) for _ in range(n)])
perm = rng.permutation(n)
act = rng.uniform(-0.15, 0.15, size=(12, n, 12)).astype(np.float32)
kw = dict(body_contacts=int(sys.argv[1])) if len(sys.argv) > 1 else {}
res = []
for order in (np.arange(n), perm):
    env = make_env("Quadrupedal", num_envs=n, device="cuda:0", lanes_per_robot=16, **kw)
    env.reset(ETG_w=W[order], ETG_b=B[order], dynamic_param=torch.as_tensor(dyn[order], dtype=torch.float32))
    sts = [env.get_state().cpu().numpy().copy()]
    for k in range(12):
        env.step(torch.as_tensor(act[k][order]))
        sts.append(env.get_state().cpu().numpy().copy())
    res.append((order, np.stack(sts)))
    env.close()
(o0, a), (o1, b) = res
inv = np.empty(n, dtype=int); inv[o1] = np.arange(n)
b = b[:, inv]
for k in range(13):
    d = np.abs(a[k] - b[k]).max(1)
    bad = np.nonzero(d > 0)[0]
    print("after step %2d: %3d robots differ, max %.2e %s" % (k, len(bad), d.max(), bad[:10].tolist()))
