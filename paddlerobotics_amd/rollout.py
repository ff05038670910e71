"""Batched counterpart of the reference's episode loops and of one ES generation.

  run_episode / run_EStrain_episode   pretrain.py:129-154, train.py:213-249
  one ES generation                   pretrain.py:220-243, train.py:398-418
  scatter / gather of candidates      model/Dynamic_parallel_model.py:157-160 (xparl) -> here one
                                      all_gather of the episode returns over RCCL (or gloo in tests)

Candidate i of the population is robot i of the global batch; with world_size ranks, rank r owns
candidates [r*N, (r+1)*N).  Every rank draws the same population (same seed) and runs the same
tell(), so the next population needs no broadcast.
"""
import numpy as np
import torch

from .etg_fit import opt_with_points_batched


def run_episodes(env, max_step, ETG_w=None, ETG_b=None, policy=None, action_bound=0.3, precision=0,
                 dynamic_param=None):
    """reset(ETG_w, ETG_b) then `max_step`+1 steps (the reference loops while steps <= max_step and passes
    donef=(steps > max_step)); returns per-robot (episode_return, episode_length) with alive masking."""
    env.reset(ETG_w=ETG_w, ETG_b=ETG_b, dynamic_param=dynamic_param)
    if policy is None and not getattr(env, "_rand_force", False):
        # open loop: the fused rollout (etg_rollout_openloop, up to 400 control steps per launch).  The forced `done`
        # of the last step only ends the episode; return and length are the same as with the stepping loop.
        return env.rollout_openloop(max_step + 1)
    if policy is not None and hasattr(env, "rollout_policy") and not getattr(env, "_rand_force", False):
        # closed loop with a fixed actor: policy tile + control step fused per workgroup (etg_rollout_policy); falls back
        # to predict() + step() inside when the fused kernel does not apply
        return env.rollout_policy(policy, max_step + 1, action_bound, precision)
    act = None
    for steps in range(1, max_step + 2):
        if policy is not None:
            act = policy.predict(env.obs, action_bound, precision, out=act)
        env.step(act, donef=(steps > max_step), want_info=False)
    return env.episode_stats()


def shard_bounds(total, rank, world):
    per = total // world
    if per * world != total:
        raise ValueError("population %d is not divisible by world size %d" % (total, world))
    return rank * per, (rank + 1) * per


def gather_returns(local_returns, dist=None):
    """The one exchange of the path: all ranks end up with the full [world*N] return vector."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_returns
    if dist.get_backend() == "gloo" and local_returns.is_cuda:   # gloo has no device all_gather: stage through the host
        out = torch.empty(local_returns.numel() * dist.get_world_size(), dtype=local_returns.dtype)
        dist.all_gather_into_tensor(out, local_returns.detach().cpu().contiguous())
        return out.to(local_returns.device)
    out = torch.empty(local_returns.numel() * dist.get_world_size(), dtype=local_returns.dtype,
                      device=local_returns.device)
    dist.all_gather_into_tensor(out, local_returns.contiguous())
    return out


def es_generation(solver, evaluate, dist=None, rank=0, world=1):
    """ask -> evaluate this rank's slice -> gather -> tell (replicated). `evaluate(solutions_slice)`
    returns the slice's fitness tensor. Returns the full fitness vector."""
    solutions = solver.ask()
    lo, hi = shard_bounds(solutions.shape[0], rank, world)
    local = evaluate(solutions[lo:hi])
    fitness = gather_returns(torch.as_tensor(local), dist)
    solver.tell(fitness)
    return fitness


def make_etg_evaluator(env, etg_layer, ETG_T, prior_points, w0, b0, max_step=400, policy=None, action_bound=0.3, rpm=None):
    """Fitness of ETG control-point offsets (12 numbers per candidate), as in pretrain.py:226-233.  With an actor and a
    replay memory `rpm` the candidates' episodes are kept for the SAC learner, as run_EStrain_episode does with --es_rpm
    (train.py:240-241, 404-409): recorded by the fused closed-loop kernel where it applies, by the stepping loop otherwise."""
    prior = torch.as_tensor(np.asarray(prior_points), dtype=torch.float64, device=env.device)

    def evaluate(solutions):
        pts = prior[None] + solutions.to(env.device).reshape(-1, 6, 2)
        w, b = opt_with_points_batched(etg_layer, ETG_T, pts, b0, w0, device=env.device)
        if rpm is not None and policy is not None:
            from . import replay
            from .env import FusedKernelUnavailable
            try:
                ret, _ = replay.collect_recorded(env, rpm, max_step, policy, action_bound, ETG_w=w.float(), ETG_b=b.float())
            except FusedKernelUnavailable:   # ONLY "the fused kernel does not cover this configuration": other errors surface
                ret, _, _ = replay.collect_transitions(env, rpm, max_step, policy=policy, action_bound=action_bound,
                                                       ETG_w=w.float(), ETG_b=b.float())
            return ret
        ret, _ = run_episodes(env, max_step, ETG_w=w.float(), ETG_b=b.float(), policy=policy,
                              action_bound=action_bound)
        return ret
    return evaluate


def dynamics_id_loss(drpy, motor, mean_dict, key):
    """loss_func of model/Dynamic_parallel_model.py:30-42, batched: drpy [N,T,3], motor [N,T,12] -> loss [N].
    Per candidate: max over the columns of the time-mean of the squared, std-normalised deviation from the recorded
    means, averaged over the motor-angle and rpy-rate groups."""
    def group(x, mean, std):
        mean = torch.as_tensor(np.asarray(mean), dtype=x.dtype, device=x.device)
        std = torch.as_tensor(np.asarray(std), dtype=x.dtype, device=x.device)
        return (((x - mean) ** 2) / (std ** 2)).mean(dim=1).max(dim=1).values
    loss_motor = group(motor, mean_dict[key + "_motor_mean"], mean_dict[key + "_motor_std"])
    loss_drpy = group(drpy, mean_dict[key + "_drpy_mean"], mean_dict[key + "_drpy_std"])
    return (loss_drpy + loss_motor) / 2.0


def make_dynamics_id_evaluator(env, gait, mean_dict, e_steps=100, keys=("exp", "ori"), fused=True):
    """Fitness of dynamic-parameter candidates (48 numbers in [-1,1] per candidate = robot), the batched
    RemoteESAgent.batch_sample_episodes of model/Dynamic_parallel_model.py:53-77: for every gait `key`, reset with
    param2dynamic_dict(candidate), replay `e_steps` recorded joint targets (action = gait[key][i] - pose_ori, an
    env made with ETG=0), record info["joint_angle"] and info["obs-IMU"][3:], reward = 30 - loss_func; the
    candidate's fitness is the mean over the keys.  `env` must have been made with ETG=0.  fused (default): the replay runs
    through env.rollout_actions (etg_rollout_actions: the tape is known, so 100 steps are 2 launches); False = env.step()
    per control step with the info columns sliced on the host side, the reference's own loop shape."""
    from . import a1_model as A
    from .env import FusedKernelUnavailable
    if getattr(env, "ETG", 1):
        raise ValueError("the dynamics-identification replay needs an env made with ETG=0 (Dynamic_parallel_model.py:49)")
    pose = torch.as_tensor(A.INIT_MOTOR_ANGLES, dtype=torch.float32, device=env.device)
    acts = {k: (torch.as_tensor(np.asarray(gait[k])[:e_steps], dtype=torch.float32, device=env.device) - pose) for k in keys}

    def evaluate(solutions):
        rows = A.param2dynamic_rows_torch(solutions.to(env.device).float())
        n = env.num_envs
        fit = torch.zeros(n, dtype=torch.float32, device=env.device)
        motor = drpy = None
        for key in keys:
            env.reset(dynamic_param=rows)
            rec = None
            if fused:
                # the commands are known in advance: one launch per 50 steps, the two info columns recorded in the kernel;
                # configurations the fused kernel does not cover (HYBRID commands, auto_reset envs) take the stepping loop
                # (the reference's replay runs all e_steps whatever `done` says -- Dynamic_parallel_model.py:58-64 has no break --
                # so the tape is rolled with finished robots simulated on)
                try:
                    env.set_rollout_mode(simulate_finished=True)
                    _, _, rec = env.rollout_actions(acts[key], record=("joint_angle", "obs-IMU"))
                    motor, drpy = rec["joint_angle"].transpose(0, 1), rec["obs-IMU"][:, :, 3:].transpose(0, 1)
                except FusedKernelUnavailable:
                    rec = None
                finally:
                    env.set_rollout_mode(simulate_finished=False)
            if rec is None:
                if motor is None or motor.shape != (n, e_steps, 12) or not motor.is_contiguous():
                    motor = torch.empty(n, e_steps, 12, device=env.device)
                    drpy = torch.empty(n, e_steps, 3, device=env.device)
                for i in range(e_steps):
                    _, _, _, info = env.step(acts[key][i].expand(n, 12), donef=False)
                    motor[:, i] = info["joint_angle"]
                    drpy[:, i] = info["obs-IMU"][:, 3:]
            fit += 30.0 - dynamics_id_loss(drpy, motor, mean_dict, key)
        return fit / len(keys)
    return evaluate
