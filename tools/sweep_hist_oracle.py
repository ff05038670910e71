import numpy as np, time, sys
from paddlerobotics_amd import a1_model as A
from oracle.oracle import OracleSim
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
w0, b0, pts = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
n=64
k=np.arange(64)
for ws in (0.85,0.1):
  for thr in (1e-7,1e-6):
    cfg=A.default_config(n, warmstart=ws, solver_residual=thr)
    orc=OracleSim(cfg,threads=16)
    orc.set_params(etg_w=w0,etg_b=b0)
    t=time.time(); orc.reset(); h0=orc.sweep_hist()
    ret,ln=orc.run_steps(100)
    h=orc.sweep_hist()
    print("ws",ws,"thr",thr,"settle mean sweeps %.2f"%((h0*k).sum()/h0.sum()),"walk mean %.2f"%((h*k).sum()/h.sum()),"max",k[h>0].max(),"len",ln.mean(),"time %.1f"%(time.time()-t))
    print((h/h.sum()).round(3)[:52])
