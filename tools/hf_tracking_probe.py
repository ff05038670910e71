#!/usr/bin/env python
"""Heightfield bit-tracking: on the configs[4] terrain (bilinear, C0: normals jump at cell edges) a foot
that comes to rest on an edge during the 500-tick settle lands on one side or the other depending on the last bit, so any two
fp32 evaluations of the same model either TRACK the fp64 oracle (gap ~1e-6) or part from it (1e-4 .. 1e-3 m).  This probe
measures the tracking rate of several fp32 evaluations against the fp64 oracle on N robots spread over the terrain:
  o32   the oracle's own fp32 build (generic CRBA + dense Cholesky, IEEE ops, no contraction)
  emu4 / emu16   the KERNEL SOURCE of both mappings compiled for the host (IEEE ops, no contraction, libm sin / cos): same
        arithmetic ORDER as the GPU, none of its hardware approximations
  gpu   the library on the device, and build variants of it (ETG_LIB=...: IEEE heightfield / contact-frame ops, no FMA
        contraction, IEEE rcp / rsq / sqrt + polynomial sin / cos everywhere)
Usage: python tools/hf_tracking_probe.py [--n 256] [--no-gpu] [--libs a.so b.so ...]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd import a1_model as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--no-emu", action="store_true")
    ap.add_argument("--libs", nargs="*", default=[])
    a = ap.parse_args()
    from oracle.oracle import OracleSim
    n = a.n
    rng = np.random.default_rng(0)
    hf = dict(heights=rng.uniform(0, 0.05, size=(256, 256)).astype(np.float32), cell=0.05, origin=(-6.4, -6.4))
    xy = np.random.default_rng(4).uniform(-2.0, 2.0, size=(n, 2))
    cfg = lambda: A.default_config(n, terrain=1, heightfield=hf, solver_iters=4)
    ncpu = os.cpu_count() or 1

    def settle_pose(sim):
        sim.set_heightfield(hf["heights"])
        sim.set_reset_offsets(xy)
        sim.reset()
        return np.asarray(sim.get_state())[:, :7].astype(np.float64)

    o64 = OracleSim(cfg(), threads=ncpu)
    ref = settle_pose(o64)
    rows = []

    def report(name, pose):
        gap = np.abs(pose - ref).max(1)
        tr = float((gap < 5e-6).mean())
        se = np.sqrt(max(tr * (1 - tr), 1e-9) / n)
        rows.append((name, tr, se, float(np.median(gap)), float(gap.max())))
        print("%-44s tracking %5.1f %% +- %.1f   median gap %.2e   max %.2e" % (name, 100 * tr, 100 * se, np.median(gap), gap.max()), flush=True)

    report("oracle fp32 (IEEE, generic algorithm)", settle_pose(OracleSim(cfg(), dtype=np.float32, threads=ncpu)))
    if not a.no_emu:
        from tests.emu.emu import EmuSim
        for lanes in (16, 4):
            report("kernel source on the host, %d lanes (IEEE)" % lanes, settle_pose(EmuSim(cfg(), lanes=lanes)))
    if not a.no_gpu:
        import subprocess
        for lib in [None] + a.libs:
            env = dict(os.environ)
            if lib:
                env["ETG_LIB"] = lib
            code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from paddlerobotics_amd.env import make_env;"
                    "rng = np.random.default_rng(0); hf = dict(heights=rng.uniform(0, 0.05, size=(256, 256)).astype(np.float32), cell=0.05, origin=(-6.4, -6.4));"
                    "xy = np.random.default_rng(4).uniform(-2.0, 2.0, size=(%d, 2));"
                    "out = {};\n"
                    "for lanes in (16, 4):\n"
                    "    e = make_env('Quadrupedal', num_envs=%d, device='cuda:0', task='heightfield', heightfield=hf, solver_iters=4, lanes_per_robot=lanes)\n"
                    "    e.set_reset_offsets(torch.as_tensor(xy, dtype=torch.float32)); e.reset(); out[lanes] = e.get_state().cpu().numpy()[:, :7]; e.close()\n"
                    "np.savez('/tmp/_hf_probe.npz', l16=out[16], l4=out[4])") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n, n)
            subprocess.check_call([sys.executable, "-c", code], env=env)
            z = np.load("/tmp/_hf_probe.npz")
            tag = os.path.basename(lib) if lib else "default build"
            report("GPU %s, 16 lanes" % tag, z["l16"])
            report("GPU %s, 4 lanes" % tag, z["l4"])


if __name__ == "__main__":
    main()
