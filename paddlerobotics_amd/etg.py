"""Host-side ETG parameterisation: control points -> RBF weights (w, b).

Mirrors the reference interface so ETGRL's drivers keep working:
  ETG_layer(T, dt, H, sigma_sq, phase, amp, T2_radio)   rlschool (absent); ctor args train.py:296-297
  LS_sol(A, b, precision, alpha, lamb, w0)               train.py:59-79
  Opt_with_points(ETG, ETG_T, points, b0, w0, ...)       train.py:81-110
The RBF form (centres on the phase circle at t_h = h*T/(H-0.9)) is pinned by the
reference's gait_action_list_ETG_exp.npy fixture (tests/test_golden_etg.py).
A batched device version of Opt_with_points lives in paddlerobotics_amd/etg_fit.py.
"""
import numpy as np


class ETG_layer:
    """Gaussian RBF features of a 2-D phase oscillator."""

    def __init__(self, T, dt, H, sigma_sq, phase, amp, T2_radio):
        self.T, self.dt, self.H = float(T), float(dt), int(H)
        self.sigma_sq, self.amp = float(sigma_sq), float(amp)
        self.phase = np.asarray(phase, dtype=np.float64).reshape(-1)
        self.omega = 2.0 * np.pi / self.T
        self.T2_ratio = float(T2_radio)
        self.t = 0.0
        centres_t = np.arange(self.H) * self.T / (self.H - 0.9)
        self.u = np.stack([self.forward(t) for t in centres_t]).reshape(self.H, -1)

    def forward(self, t):
        return self.amp * np.sin(self.phase + t * self.omega)

    def _rbf(self, t):
        d = self.forward(t)[None, :] - self.u
        return np.exp(-np.sum(d * d, axis=1) / self.sigma_sq)

    def update(self, t=None):
        time = self.t if t is None else t
        self.t += self.dt
        return self._rbf(time)

    def update2(self, t=None, info=None):
        time = self.t if t is None else t
        self.t += self.dt
        return self._rbf(time), self._rbf(time + self.T2_ratio * self.T)

    def reset(self):
        self.t = 0.0


def LS_sol(A, b, precision=1e-4, alpha=0.05, lamb=1, w0=None):
    """Gradient descent on |Ax-b|^2 (+ lamb|x-w0|^2), <=1000 steps (train.py:59-79)."""
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    x = np.zeros((A.shape[1], 1)) if w0 is None else np.array(w0, dtype=np.float64).reshape(-1, 1)
    anchor = None if w0 is None else x.copy()
    AtA, Atb = A.T @ A, A.T @ b

    def sq_err(v):
        r = A @ v - b
        return float((r.T @ r).item())

    it = 0
    while sq_err(x) > precision and it < 1000:
        g = AtA @ x - Atb
        if anchor is not None:
            g = g + lamb * (x - anchor)
        x = x - alpha * g
        it += 1
    return x


def default_points(Steplength=0.05, Footheight=0.08, Penetration=0.01):
    """The 6 prior control points of train.py:87-88."""
    return np.array([[0, -Penetration], [-Steplength, -Penetration * 0.5],
                     [-Steplength * 1.5, 0.6 * Footheight], [0, Footheight],
                     [Steplength * 1.5, 0.6 * Footheight], [Steplength, -Penetration * 0.5]])


def control_times(ETG_T):
    return [0.5 * ETG_T + 0.1, 0, 0.05, 0.1, 0.15, 0.2]       # train.py:82


def Opt_with_points(ETG, ETG_T=0.4, points=None, b0=None, w0=None, precision=1e-4, lamb=0.5,
                    plot=False, **kwargs):
    """Fit ETG weights through 6 (x,z) control points; returns (w[3,20], b[3], points)."""
    if points is None:
        points = default_points(kwargs.get("Steplength", 0.05), kwargs.get("Footheight", 0.08),
                                kwargs.get("Penetration", 0.01))
    points = np.asarray(points, dtype=np.float64)
    feats = np.array([ETG.update(t) for t in control_times(ETG_T)]).reshape(-1, ETG.H)
    b = np.mean(points, axis=0) if b0 is None else np.array([b0[0], b0[-1]])
    centred = points - b
    sols = []
    for dim, row in ((0, 0), (1, -1)):
        if w0 is None:
            sols.append(LS_sol(feats, centred[:, dim], precision=precision, alpha=0.05))
        else:
            sols.append(LS_sol(feats, centred[:, dim], precision=precision, alpha=0.05, lamb=lamb,
                               w0=np.asarray(w0)[row, :]))
    w_ = np.zeros((3, ETG.H))
    w_[0], w_[2] = sols[0].reshape(-1), sols[1].reshape(-1)
    b_ = np.array([b[0], 0.0, b[1]])
    return w_, b_, points


def etg_joint_action(layer, w, b, t):
    """Joint-space ETG action (12, minus pose_ori) at control time t -- the host-side restatement of what the
    step kernel does per robot (csrc/etg_core.h: etg_action): RBF features at t for legs FR/RL and at
    t + T2*T for FL/RR (trot), foot offset w r + b in the base frame, closed-form leg IK
    (deployment/robots/a1.py:97-110).  Used for exporting `gait_action_list_*.npy` and by tools/."""
    from . import a1_model as A
    r1, r2 = layer._rbf(t), layer._rbf(t + layer.T2_ratio * layer.T)
    w, b = np.asarray(w, dtype=np.float64), np.asarray(b, dtype=np.float64)
    act = np.zeros(12)
    for leg in range(4):
        d = w @ (r1 if leg in (0, 3) else r2) + b
        x, y, z = A.BASE_FOOT[3 * leg:3 * leg + 3] + d - A.HIP_OFFSETS[leg]
        l_hip = A.L_HIP * (-1.0) ** (leg + 1)
        ck = (x * x + y * y + z * z - l_hip ** 2 - A.L_LOW ** 2 - A.L_UP ** 2) / (2 * A.L_LOW * A.L_UP)
        knee = -np.arccos(np.clip(ck, -1, 1))
        l = np.sqrt(max(A.L_UP ** 2 + A.L_LOW ** 2 + 2 * A.L_UP * A.L_LOW * np.cos(knee), 1e-12))
        hip = np.arcsin(np.clip(-x / l, -1, 1)) - knee / 2
        c1 = l_hip * y - l * np.cos(hip + knee / 2) * z
        s1 = l * np.cos(hip + knee / 2) * y + l_hip * z
        act[3 * leg:3 * leg + 3] = np.array([np.arctan2(s1, c1), hip, knee]) - A.INIT_MOTOR_ANGLES[3 * leg:3 * leg + 3]
    return act
