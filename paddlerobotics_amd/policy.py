"""Residual policy forward on the MFMA kernel (csrc/policy_mlp.hip).

Mirrors model/mujoco_model.py:44-60 (Actor) + alg/sac.py:60-76 (predict = tanh(mean); sample = tanh of the
reparameterised Gaussian, with its log-probability) and
loads the reference's checkpoints (mujoco_agent.py:61-65: torch state_dict with keys
actor_model.l1.weight, ... , actor_model.mean_linear.bias).
"""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class MfmaPolicy:
    def __init__(self, obs_dim, action_dim=12, hidden=256, device="cuda:0"):
        self.obs_dim, self.action_dim, self.hidden = int(obs_dim), int(action_dim), int(hidden)
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._lib.etg_policy_create(self.obs_dim, self.hidden, self.action_dim, idx, C.byref(self._h)))
        self._w = None

    @staticmethod
    def init_like_reference(obs_dim, action_dim=12, hidden=256, seed=0):
        """torch default nn.Linear init under torch.manual_seed(seed) (BASELINE config 3)."""
        torch.manual_seed(seed)
        l1 = torch.nn.Linear(obs_dim, hidden)
        l2 = torch.nn.Linear(hidden, hidden)
        mean = torch.nn.Linear(hidden, action_dim)
        std = torch.nn.Linear(hidden, action_dim)
        return {"actor_model.l1.weight": l1.weight.detach(), "actor_model.l1.bias": l1.bias.detach(),
                "actor_model.l2.weight": l2.weight.detach(), "actor_model.l2.bias": l2.bias.detach(),
                "actor_model.mean_linear.weight": mean.weight.detach(),
                "actor_model.mean_linear.bias": mean.bias.detach(),
                "actor_model.std_linear.weight": std.weight.detach(),
                "actor_model.std_linear.bias": std.bias.detach()}

    def load_state_dict(self, sd):
        keys = ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "mean_linear.weight", "mean_linear.bias")
        ws = []
        for k in keys:
            t = sd["actor_model." + k] if ("actor_model." + k) in sd else sd[k]
            ws.append(torch.as_tensor(t, dtype=torch.float32).to(self.device).contiguous())
        if tuple(ws[0].shape) != (self.hidden, self.obs_dim) or tuple(ws[4].shape) != (self.action_dim, self.hidden):
            raise ValueError("state_dict shapes do not match the policy dimensions")
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib.etg_policy_load(self._h, *[_ptr(w) for w in ws], stream))
        self._w = ws
        self._std = None
        for pre in ("actor_model.", ""):                    # the log-std head, if the checkpoint has it
            if pre + "std_linear.weight" in sd:
                st = [torch.as_tensor(sd[pre + "std_linear." + k], dtype=torch.float32).to(self.device).contiguous()
                      for k in ("weight", "bias")]
                _lib.check(self._lib.etg_policy_load_std(self._h, _ptr(st[0]), _ptr(st[1]), stream))
                self._std = st
                break

    def restore(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu"))

    def predict(self, obs, act_scale=1.0, precision=0, out=None):
        """obs [N, obs_dim] float32 on the device -> tanh(mean) * act_scale, [N, action_dim]."""
        if self._w is None:
            raise RuntimeError("load_state_dict() first")
        obs = obs.contiguous()
        if obs.dtype != torch.float32 or obs.dim() != 2 or obs.shape[1] != self.obs_dim:
            raise ValueError("obs must be float32 [N,%d]" % self.obs_dim)
        if out is None:
            out = torch.empty(obs.shape[0], self.action_dim, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib.etg_policy_forward(self._h, _ptr(obs), int(obs.shape[0]), C.c_float(act_scale),
                                                int(precision), _ptr(out), stream))
        return out

    def sample(self, obs, act_scale=1.0, precision=0, noise=None, generator=None, return_logp=True):
        """SAC.sample (alg/sac.py:65-76): tanh(mean + exp(clamp(log_std, -20, 2)) * N(0,1)) * act_scale and the
        log-probability of the squashed action; `noise` [N, action_dim] may be supplied for reproducibility."""
        if self._w is None or self._std is None:
            raise RuntimeError("load_state_dict() with a std_linear head first")
        obs = obs.contiguous()
        if obs.dtype != torch.float32 or obs.dim() != 2 or obs.shape[1] != self.obs_dim:
            raise ValueError("obs must be float32 [N,%d]" % self.obs_dim)
        n = int(obs.shape[0])
        if noise is None:
            noise = torch.randn(n, self.action_dim, device=self.device, generator=generator)
        noise = noise.to(torch.float32).contiguous()
        act = torch.empty(n, self.action_dim, device=self.device)
        logp = torch.empty(n, device=self.device) if return_logp else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib.etg_policy_sample(self._h, _ptr(obs), n, _ptr(noise), C.c_float(act_scale), int(precision),
                                               _ptr(act), None if logp is None else _ptr(logp), stream))
        return (act, logp.unsqueeze(1)) if return_logp else act

    def close(self):
        if self._h:
            self._lib.etg_policy_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
