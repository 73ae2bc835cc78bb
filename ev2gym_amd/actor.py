"""Policy-in-the-loop rollouts: the BASELINE configs[4] shape (an SB3-DDPG-sized actor, obs -> 400 -> 300 -> P, tanh,
producing the actions on the device between env steps).

The reference trains SB3 agents against one CPU env (`/root/reference/train_stable_baselines.py:62-130`); here the actor
and thousands of envs live on the same GPU.  The engine hands observations over as float32 (ev2g_step_extras.obs_f32) and
takes float32 actions (ev2g_step_extras.actions_f32), so no conversion pass sits between the policy and the step kernel, and
`FusedMLPActor` evaluates the whole network in ONE kernel (`ev2g_mlp_forward`: bf16 MFMA, fp32 accumulation) enqueued together
with the env step by one C call per rollout segment (`ev2g_rollout`).  `TorchMLPActor` is the same network through
torch.nn (fp32, a dozen launches per forward): the comparison point, and the numerics reference of the fused kernel.
The policy network itself is outside the hot path of SURVEY.md par.8 -- random weights, no learner.
"""
from __future__ import annotations

import numpy as np


def init_mlp_weights(D, P, seed=0, h1=400, h2=300):
    """torch.nn.Linear-style uniform(-1/sqrt(in), 1/sqrt(in)) weights as numpy float32 (W[out,in], b[out])."""
    rng = np.random.default_rng(seed)
    out = []
    for n_in, n_out in ((D, h1), (h1, h2), (h2, P)):
        k = 1.0 / np.sqrt(n_in)
        out += [rng.uniform(-k, k, (n_out, n_in)).astype(np.float32), rng.uniform(-k, k, n_out).astype(np.float32)]
    return out


def mlp_forward_numpy(x, weights, lo, bf16=False):
    """Reference forward.  bf16=True mimics the fused kernel's operand rounding (bf16 inputs / weights / hidden activations,
    fp32 accumulation)."""
    def r(a):
        if not bf16:
            return a.astype(np.float32)
        u = np.ascontiguousarray(a, np.float32).view(np.uint32)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.view(np.float32)
    W1, b1, W2, b2, W3, b3 = weights
    h = np.maximum(r(x) @ r(W1).T + b1, 0)
    h = np.maximum(r(h) @ r(W2).T + b2, 0)
    y = np.tanh(r(h) @ r(W3).T + b3)
    return (y * 0.5 + 0.5) if lo == 0.0 else y


class FusedMLPActor:
    """obs[E,D] float32 -> actions[E,P] float32 in [lo, 1], one kernel per forward; `run(loop, k)` enqueues k rollout steps."""

    def __init__(self, eng, E, P, D, lo, dev=None, seed=0, weights=None, precision="bf16"):
        self.eng, self.E, self.P, self.D, self.lo = eng, E, P, D, lo
        self.weights = weights if weights is not None else init_mlp_weights(D, P, seed)
        self.precision = precision
        self.mlp = eng.mlp_create(*self.weights, out_lo=lo, precision=precision)
        if dev is not None:
            import torch
            self.obs32 = torch.zeros((E, D), dtype=torch.float32, device=dev)
            self.act32 = torch.zeros((E, P), dtype=torch.float32, device=dev)
        else:
            self.obs32, self.act32 = eng.empty((E, D), np.float32), eng.empty((E, P), np.float32)
        eng.set_extras(obs_f32=self.obs32, obs_f32_stride=0, actions_f32=self.act32)
        self.describe = (f"fused MLP {D}->400->300->{P} tanh: one kernel per forward ({'bf16 operands' if precision == 'bf16' else ('float32 weights as two bf16 terms' if precision in ('fp32', 'f32') else 'float32 weights as three bf16 terms')} on the bf16 matrix cores, fp32 accumulate), float32 "
                         "obs/action hand-over, one C call per segment (ev2g_rollout: ONE launch with the policy inside the step kernel's launch where the shape is eligible, else actor + step launches)")

    def run(self, loop, k):
        loop.eng.rollout(self.mlp, k, loop.rew, 0, loop.done, 0, loop.mask, 0, auto_reset=0)

    def forward_train_us(self, n=200):
        """Average duration of one forward inside a back-to-back train of n on the engine's stream (microseconds)."""
        import time
        self.eng.mlp_forward(self.mlp, self.obs32, self.act32, self.E)
        self.eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            self.eng.mlp_forward(self.mlp, self.obs32, self.act32, self.E)
        self.eng.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    def close(self):
        self.eng.mlp_destroy(self.mlp)


class TorchMLPActor:
    """The same network through torch.nn (fp32): one forward = a dozen launches.  Comparison point."""

    def __init__(self, eng, E, P, D, lo, dev, seed=0, weights=None):
        import torch
        self.torch = torch
        self.eng, self.E, self.P, self.D, self.lo = eng, E, P, D, lo
        self.weights = weights if weights is not None else init_mlp_weights(D, P, seed)
        self.net = torch.nn.Sequential(torch.nn.Linear(D, 400), torch.nn.ReLU(), torch.nn.Linear(400, 300), torch.nn.ReLU(),
                                       torch.nn.Linear(300, P), torch.nn.Tanh()).to(dev)
        with torch.no_grad():
            for lin, (W, b) in zip((self.net[0], self.net[2], self.net[4]), zip(self.weights[0::2], self.weights[1::2])):
                lin.weight.copy_(torch.from_numpy(W)); lin.bias.copy_(torch.from_numpy(b))
        self.obs32 = torch.zeros((E, D), dtype=torch.float32, device=dev)
        self.act32 = torch.zeros((E, P), dtype=torch.float32, device=dev)
        eng.set_extras(obs_f32=self.obs32, obs_f32_stride=0, actions_f32=self.act32)
        self.describe = f"torch MLP {D}->400->300->{P} tanh, fp32, float32 obs/action hand-over"

    def forward(self):
        torch = self.torch
        with torch.no_grad():
            a = self.net(self.obs32)
            if self.lo == 0.0:
                a = a * 0.5 + 0.5
            self.act32.copy_(a)

    def run(self, loop, k):
        for _ in range(k):
            self.forward()
            loop.eng.step_n(1, None, 0, None, 0, loop.rew, 0, loop.done, 0, loop.mask, 0, auto_reset=False, persistent=False)

    def close(self):
        pass


def make_actor(eng, E, P, D, lo, dev, seed=0, kind="fused"):
    if kind == "fused_fp32":
        return FusedMLPActor(eng, E, P, D, lo, dev, seed=seed, precision="fp32")
    cls = FusedMLPActor if kind == "fused" else TorchMLPActor
    return cls(eng, E, P, D, lo, dev, seed=seed)
