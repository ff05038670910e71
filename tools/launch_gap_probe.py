"""Launch-gap probe (run by hand on the GPU box: python tools/launch_gap_probe.py): per-step cost of env.step() with and
without HIP event pairs, the host cost of one call, and the fused rollout."""
def main():
    import sys, time, torch
    sys.path.insert(0, "/root/repo")
    import bench as B
    from paddlerobotics_amd.env import make_env
    N = 4096
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0")
    w, b = B.etg_population(N, 0, torch.device("cuda:0"))
    env.reset(ETG_w=w, ETG_b=b)
    for _ in range(20): env.step(None, want_info=False)
    def run(K, events):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            if events: ev[k][0].record()
            env.step(None, want_info=False)
            if events: ev[k][1].record()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        return dt / K * 1e6
    for rep in range(2):
        print("events per step: %.1f us/step   no events: %.1f us/step" % (run(400, True), run(400, False)))
    # host-side cost of one env.step call (no sync)
    t0 = time.perf_counter()
    for k in range(400): env.step(None, want_info=False)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("host time per env.step call: %.1f us" % ((t1 - t0) / 400 * 1e6))
    # one call for 400 steps through the C-ABI rollout (host loop inside C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout_openloop(400); torch.cuda.synchronize()
    print("etg_rollout_openloop(400): %.1f us/step" % ((time.perf_counter() - t0) / 400 * 1e6))


if __name__ == "__main__":
    main()
