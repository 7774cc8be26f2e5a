"""TEST INFRASTRUCTURE ONLY -- numpy float32 restatement of the trainer arithmetic that the device
helpers sf_compute_returns / sf_record_step replace.  PINNED: tests/golden/trainer/*.npz were produced by
the reference's own rl/storage.py (imported) and the loop body of rl/train.py:82-88
(tests/golden/trainer/make_returns_golden.py)."""
import numpy as np

f32 = np.float32


def compute_returns(rewards, value_preds, masks, next_value, use_gae, gamma, tau):
    """RolloutStorage.compute_returns, rl/storage.py:50-63.  Arrays [T][N] / [T+1][N] float32.
    Returns (returns [T+1][N], value_preds as the reference leaves them)."""
    T = rewards.shape[0]
    vp = value_preds.astype(f32).copy()
    ret = np.zeros_like(vp)
    g, gt = f32(gamma), f32(gamma * tau)  # Python scalars meet float32 tensors
    if use_gae:
        vp[-1] = next_value                                                     # :52
        gae = f32(0)
        for t in reversed(range(T)):
            delta = (rewards[t] + (g * vp[t + 1]) * masks[t + 1]) - vp[t]       # :55
            gae = delta + (gt * masks[t + 1]) * gae                             # :56
            ret[t] = gae + vp[t]                                                # :57
    else:
        ret[-1] = next_value                                                    # :59
        for t in reversed(range(T)):
            ret[t] = ((ret[t + 1] * g) * masks[t + 1]) + rewards[t]             # :61-62
    return ret, vp


def record_step(reward, done, episode_rewards, final_rewards):
    """rl/train.py:82-88 for one step; returns (reward float32, masks, episode_rewards, final_rewards)."""
    r = reward.astype(f32)
    masks = np.where(done, f32(0), f32(1)).astype(f32)
    ep = episode_rewards + r
    fin = final_rewards * masks
    fin = fin + (f32(1) - masks) * ep
    ep = ep * masks
    return r, masks, ep, fin
