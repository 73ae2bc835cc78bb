"""Policy-in-the-loop rollouts: the BASELINE configs[4] shape (an SB3-DDPG-sized actor, obs -> 400 -> 300 -> P, tanh,
producing the actions on the device between env steps).

The reference trains SB3 agents against one CPU env (`/root/reference/train_stable_baselines.py:62-130`); here the actor
and thousands of envs live on the same GPU, and the engine hands observations over as float32 (ev2g_step_extras.obs_f32)
and takes float32 actions (ev2g_step_extras.actions_f32), so no conversion pass sits between the policy and the step
kernel.  The policy network itself is outside the hot path of SURVEY.md §8 -- it is PyTorch (plumbing), random weights.
"""
from __future__ import annotations


class TorchMLPActor:
    """obs[E,D] float32 -> actions[E,P] float32 in [lo, 1].  One forward per env step, written straight into the buffer the
    step kernel reads."""

    def __init__(self, eng, E, P, D, lo, dev, seed=0, dtype="fp32"):
        import torch
        self.torch = torch
        self.eng, self.E, self.P, self.D, self.lo = eng, E, P, D, lo
        torch.manual_seed(seed)
        self.net = torch.nn.Sequential(torch.nn.Linear(D, 400), torch.nn.ReLU(), torch.nn.Linear(400, 300), torch.nn.ReLU(),
                                       torch.nn.Linear(300, P), torch.nn.Tanh()).to(dev)
        self.obs32 = torch.zeros((E, D), dtype=torch.float32, device=dev)
        self.act32 = torch.zeros((E, P), dtype=torch.float32, device=dev)
        eng.set_extras(obs_f32=self.obs32, obs_f32_stride=0, actions_f32=self.act32)
        self.describe = f"torch MLP {D}->400->300->{P} tanh, {dtype}, float32 obs/action hand-over"

    def forward(self):
        torch = self.torch
        with torch.no_grad():
            a = self.net(self.obs32)
            if self.lo == 0.0:
                a = a * 0.5 + 0.5
            self.act32.copy_(a)

    def step(self, loop):
        """One policy forward + one env step (float32 actions: the `actions` argument of ev2g_step_n stays NULL)."""
        self.forward()
        loop.eng.step_n(1, None, 0, None, 0, loop.rew, 0, loop.done, 0, loop.mask, 0, auto_reset=False, persistent=False)


def make_actor(eng, E, P, D, lo, dev, seed=0):
    return TorchMLPActor(eng, E, P, D, lo, dev, seed=seed)
