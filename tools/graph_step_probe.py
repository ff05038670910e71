"""Does replaying env.step() launches from a HIP graph shorten the gap between dependent launches?  4096 robots, one MI355X:
env.step(None) back to back on a stream, against the same steps captured once (torch.cuda.CUDAGraph = hipGraph) and replayed;
then the closed loop policy.predict() + env.step() the same two ways."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from paddlerobotics_amd.env import make_env
from paddlerobotics_amd.policy import MfmaPolicy
from paddlerobotics_amd import a1_model as A
N = 4096
w, b = bench.etg_population(N, 0, "cuda:0")
env = make_env("Quadrupedal", num_envs=N, device="cuda:0")
pol = MfmaPolicy(A.OBS_DIM, 12, device="cuda:0"); pol.load_state_dict(MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0))
act = torch.zeros(N, 12, device="cuda:0")

def plain(closed, n=400):
    env.reset(ETG_w=w, ETG_b=b)
    for _ in range(50):
        env.step(None, want_info=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if closed:
            pol.predict(env.obs, 0.3, 0, out=act); env.step(act, want_info=False)
        else:
            env.step(None, want_info=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

def graphed(closed, chunk=50, reps=8):
    env.reset(ETG_w=w, ETG_b=b)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            if closed: pol.predict(env.obs, 0.3, 0, out=act); env.step(act, want_info=False)
            else: env.step(None, want_info=False)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(chunk):
            if closed: pol.predict(env.obs, 0.3, 0, out=act); env.step(act, want_info=False)
            else: env.step(None, want_info=False)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * chunk) * 1e6

for closed in (False, True):
    name = "policy.predict() + env.step()" if closed else "env.step(None)"
    try:
        print("%-32s plain launches %.2f us per control step" % (name, plain(closed)), flush=True)
        print("%-32s hipGraph replay %.2f us per control step" % (name, graphed(closed)), flush=True)
    except Exception as e:
        print(name, "failed:", repr(e)[:300], flush=True)
