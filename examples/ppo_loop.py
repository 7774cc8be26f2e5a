#!/usr/bin/env python3
"""The loop of the reference's trainer (rl/train.py:58-130: act -> envs.step -> bookkeeping -> insert ->
compute_returns -> PPO epochs) with every environment-side piece on the device:

    envs     SFVecNormalize(SFVecEnv(...))       <- VecNormalize(SubprocVecEnv([...]))     rl/train.py:30-36
    rollouts DeviceRollout(envs, T)              <- RolloutStorage + the per-step bookkeeping rl/train.py:41,79-98

The actor-critic below is a stand-in (two tanh layers, like rl/model.py's MLP branch) so that the script is
self-contained; the point is the data path: nothing crosses PCIe inside an iteration.

    python examples/ppo_loop.py --envs 4096 --iters 20
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import DeviceRollout, SFVecEnv, SFVecNormalize  # noqa: E402


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, n_actions, hidden=64):
        super().__init__()
        self.body = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.pi, self.v = nn.Linear(hidden, n_actions), nn.Linear(hidden, 1)

    def forward(self, obs):
        h = self.body(obs)
        return torch.distributions.Categorical(logits=self.pi(h)), self.v(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20, help="num_fwd_steps")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gametype", default="autoturn")
    ap.add_argument("--ppo-epochs", type=int, default=4)
    ap.add_argument("--mini-batches", type=int, default=4)
    a = ap.parse_args()
    torch.manual_seed(0)
    envs = SFVecNormalize(SFVecEnv(a.envs, gametype=a.gametype, spawn_stride=1))
    ro = DeviceRollout(envs, a.steps)
    net = ActorCritic(envs.venv.obs_dim, envs.venv.n_actions).to(envs.device)
    opt = torch.optim.Adam(net.parameters(), lr=7e-4)
    ro.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.iters):
        for t in range(a.steps):
            with torch.no_grad():
                dist, value = net(ro.observations[t])
                action = dist.sample()
            ro.step(t, action, value_pred=value, action_log_prob=dist.log_prob(action).unsqueeze(1))
        with torch.no_grad():
            next_value = net(ro.observations[-1])[1]
        ro.compute_returns(next_value, True, 0.99, 0.95)
        adv = ro.returns[:-1] - ro.value_preds[:-1]
        adv = (adv - adv.mean()) / (adv.std() + 1e-5)
        for _ in range(a.ppo_epochs):
            for obs, _, act, ret, _, old_logp, adv_t in ro.feed_forward_generator(adv, a.mini_batches):
                dist, value = net(obs)
                ratio = torch.exp(dist.log_prob(act.squeeze(1)).unsqueeze(1) - old_logp)
                loss = (-torch.min(ratio * adv_t, torch.clamp(ratio, 0.9, 1.1) * adv_t).mean()
                        + 0.5 * (value - ret).pow(2).mean() - 0.01 * dist.entropy().mean())
                opt.zero_grad()
                loss.backward()
                opt.step()
        ro.after_update()
        if (it + 1) % 5 == 0 or it == a.iters - 1:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("iter %3d  env-steps %9d  %.3g env-steps/s (whole loop)  mean final reward %.3f  kills %d" % (
                it + 1, (it + 1) * a.steps * a.envs, (it + 1) * a.steps * a.envs / dt, float(ro.final_rewards.mean()),
                ro.num_destruction))
    envs.close()


if __name__ == "__main__":
    main()
