"""SimpleGA (the solver ETGRL uses: train.py:288-295, pretrain.py) with the population kept as a
torch tensor on any device, so ask -> Opt_with_points -> rollout -> tell never leaves the GPU.

Mirrors alg/es.py:214-326 (estool's SimpleGA): same constructor arguments, ask()/tell()/result()/
reset()/get_best_param()/current_param(), same elite selection, mating, sigma decay and L2 weight
decay (alg/es.py:29-31).  Random draws come from a torch.Generator; `ask(draws=...)` accepts the
three draw arrays explicitly, which is how the golden test replays the reference's numpy stream.
"""
import torch


class SimpleGA:
    def __init__(self, num_params, sigma_init=0.1, sigma_decay=0.999, sigma_limit=0.01, popsize=256,
                 elite_ratio=0.1, forget_best=False, weight_decay=0.01, param=None, device="cpu", seed=0,
                 dtype=torch.float64):
        self.num_params, self.popsize = int(num_params), int(popsize)
        self.sigma_init, self.sigma_decay, self.sigma_limit = sigma_init, sigma_decay, sigma_limit
        self.elite_ratio = elite_ratio
        self.elite_popsize = int(self.popsize * self.elite_ratio)
        self.sigma = self.sigma_init
        self.device, self.dtype = torch.device(device), dtype
        self.elite_params = torch.zeros(self.elite_popsize, self.num_params, dtype=dtype, device=self.device)
        self.elite_rewards = torch.zeros(self.elite_popsize, dtype=dtype, device=self.device)
        self.best_param = torch.zeros(self.num_params, dtype=dtype, device=self.device) if param is None else \
            torch.as_tensor(param, dtype=dtype, device=self.device).clone()
        self.curr_best_param = self.best_param
        # the best rewards stay 0-d tensors on the solver's device (tell() does not synchronise: the generation loop of
        # rollout.es_generation queues ask -> fit -> rollout -> tell without a host read); result() turns them into floats
        self._best_reward = torch.zeros((), dtype=dtype, device=self.device)
        self._curr_best_reward = torch.zeros((), dtype=dtype, device=self.device)
        self.first_iteration = True
        self.forget_best = forget_best
        self.weight_decay = weight_decay
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def reset(self, param):
        self.best_param = torch.as_tensor(param, dtype=self.dtype, device=self.device).clone()
        self.curr_best_param = self.best_param.clone()
        self.first_iteration = True

    def rms_stdev(self):
        return self.sigma

    def ask(self, draws=None):
        """Returns solutions [popsize, num_params].  draws = (normal[pop,n], parents[pop,2] ints,
        mate_uniform[pop,n]) overrides the generator (parents/mate are unused on the first iteration)."""
        P, n = self.popsize, self.num_params
        if draws is None:
            normal = torch.randn(P, n, generator=self.gen, device=self.device, dtype=self.dtype)
            parents = torch.randint(0, max(self.elite_popsize, 1), (P, 2), generator=self.gen, device=self.device)
            mate_u = torch.rand(P, n, generator=self.gen, device=self.device, dtype=self.dtype)
        else:
            normal, parents, mate_u = [torch.as_tensor(d, device=self.device) for d in draws]
            normal, mate_u = normal.to(self.dtype), mate_u.to(self.dtype)
        self.epsilon = normal * self.sigma
        if self.first_iteration:
            solutions = self.best_param[None, :] + self.epsilon
        else:
            a = self.elite_params[parents[:, 0].long()]
            b = self.elite_params[parents[:, 1].long()]
            child = torch.where(mate_u > 0.5, b, a)          # mate(): c[idx] = b[idx] where rand > 0.5
            solutions = child + self.epsilon
        self.solutions = solutions
        return solutions

    def tell(self, reward_table_result):
        reward_table = torch.as_tensor(reward_table_result, dtype=self.dtype, device=self.device).clone()
        assert reward_table.numel() == self.popsize, "Inconsistent reward_table size reported."
        if self.weight_decay > 0:
            reward_table = reward_table - self.weight_decay * (self.solutions * self.solutions).mean(dim=1)
        if self.forget_best or self.first_iteration:
            reward, solution = reward_table, self.solutions
        else:
            reward = torch.cat([reward_table, self.elite_rewards])
            solution = torch.cat([self.solutions, self.elite_params])
        # alg/es.py:300 `np.argsort(reward)[::-1][0:elite_popsize]`: the REVERSE of an ascending sort, so among equal rewards the
        # later index comes first (a descending stable sort would put the earlier one first)
        idx = torch.argsort(reward, stable=True).flip(0)[: self.elite_popsize]
        self.elite_rewards = reward[idx]
        self.elite_params = solution[idx]
        self._curr_best_reward = self.elite_rewards[0].clone()
        self.curr_best_param = self.elite_params[0].clone()
        if self.first_iteration:
            self.first_iteration = False
            self._best_reward = self._curr_best_reward
            self.best_param = self.curr_best_param
        else:                                               # alg/es.py:308-311 `if curr_best_reward > best_reward`, as a select on the device
            better = self._curr_best_reward > self._best_reward
            self._best_reward = torch.where(better, self._curr_best_reward, self._best_reward)
            self.best_param = torch.where(better, self.curr_best_param, self.best_param)
        if self.sigma > self.sigma_limit:
            self.sigma *= self.sigma_decay

    @property
    def best_reward(self):
        return float(self._best_reward)

    @property
    def curr_best_reward(self):
        return float(self._curr_best_reward)

    def current_param(self):
        return self.elite_params[0]

    def set_mu(self, mu):
        pass

    def get_best_param(self):
        return self.best_param

    def result(self):
        return (self.best_param, self.best_reward, self.curr_best_reward, self.sigma, self.curr_best_param)
