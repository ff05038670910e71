"""time the fused MFMA policy kernel (BASELINE config 3 shape): 4096 x (49 -> 256 -> 256 -> 12)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.policy import MfmaPolicy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pol = MfmaPolicy(49, 12); pol.load_state_dict(MfmaPolicy.init_like_reference(49, 12, seed=0))
obs = torch.randn(n, 49, device="cuda:0"); out = torch.empty(n, 12, device="cuda:0")
for prec in (0, 1):
    for _ in range(20): pol.predict(obs, 0.3, prec, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): pol.predict(obs, 0.3, prec, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print("precision %d: %.2f us per call, %.1f TFLOP/s" % (prec, us, 2 * n * (49 * 256 + 256 * 256 + 256 * 12) / us / 1e6))
