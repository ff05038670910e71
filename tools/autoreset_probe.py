"""Where the auto-reset step spends its time (run by hand on the GPU box, optionally under rocprofv3 --kernel-trace --stats):
host time per env.step(auto_reset) call vs device time per step, next to the plain env.step()."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench as B
    from paddlerobotics_amd.env import make_env
    N = 4096
    dev = torch.device("cuda:0")
    w, b = B.etg_population(N, 0, dev)
    for auto in (False, True):
        env = make_env("Quadrupedal", num_envs=N, device="cuda:0", auto_reset=auto)
        env.reset(ETG_w=w, ETG_b=b)
        for _ in range(50):
            env.step(None, want_info=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            env.step(None, want_info=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("auto_reset=%s: host %.1f us per call, wall %.1f us per step, done fraction per step %.4f" % (
            auto, (t1 - t0) / 400 * 1e6, (t2 - t0) / 400 * 1e6, float(env.done.float().mean())))
        env.close()


if __name__ == "__main__":
    main()
