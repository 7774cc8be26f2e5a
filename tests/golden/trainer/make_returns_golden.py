"""Generates tests/golden/returns_*.npz by running the REFERENCE's own RolloutStorage.compute_returns
(/root/reference/rl/storage.py:50-63, pure torch, imported by path) and the trainer's episode bookkeeping
(/root/reference/rl/train.py:82-88, restated inline because it is a loop body, not a function) on seeded
inputs.  Build container only; the .npz files are the fixtures that travel.

    python tests/golden/trainer/make_returns_golden.py
"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_storage", "/root/reference/rl/storage.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


class _Discrete:  # RolloutStorage only looks at the class name (rl/storage.py:17)
    pass


_Discrete.__name__ = "Discrete"


def case(name, T, N, use_gae, gamma, tau, seed, p_done):
    g = torch.Generator().manual_seed(seed)
    st = ref.RolloutStorage(T, N, (3,), _Discrete(), 1)
    st.rewards.copy_(torch.randint(-1, 4, (T, N, 1), generator=g).float())
    st.value_preds.copy_(torch.randn(T + 1, N, 1, generator=g) * 3)
    st.masks.copy_((torch.rand(T + 1, N, 1, generator=g) >= p_done).float())
    next_value = torch.randn(N, 1, generator=g)
    vp_in = st.value_preds.clone()
    st.compute_returns(next_value, use_gae, gamma, tau)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rewards=st.rewards.numpy()[..., 0], value_preds_in=vp_in.numpy()[..., 0],
                        value_preds_out=st.value_preds.numpy()[..., 0], masks=st.masks.numpy()[..., 0],
                        next_value=next_value.numpy()[:, 0], returns=st.returns.numpy()[..., 0],
                        use_gae=use_gae, gamma=gamma, tau=tau)


def bookkeeping(name, T, N, seed):
    """rl/train.py:82-88 for T steps: reward float, masks, episode_rewards, final_rewards."""
    rng = np.random.RandomState(seed)
    rewards = rng.randint(-1, 4, (T, N)).astype(np.int32)
    done = rng.rand(T, N) < 0.07
    episode_rewards, final_rewards = torch.zeros([N, 1]), torch.zeros([N, 1])
    ep, fin, msk = [], [], []
    for t in range(T):
        reward = torch.from_numpy(np.expand_dims(np.stack(list(rewards[t])), 1)).float()
        episode_rewards += reward
        masks = torch.FloatTensor([[0.0] if i else [1.0] for i in done[t]])
        final_rewards *= masks
        final_rewards += (1 - masks) * episode_rewards
        episode_rewards *= masks
        ep.append(episode_rewards.numpy()[:, 0].copy())
        fin.append(final_rewards.numpy()[:, 0].copy())
        msk.append(masks.numpy()[:, 0].copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rewards=rewards, done=done.astype(np.uint8),
                        episode_rewards=np.stack(ep), final_rewards=np.stack(fin), masks=np.stack(msk))


if __name__ == "__main__":
    case("returns_gae", 20, 257, True, 0.99, 0.95, 1, 0.05)
    case("returns_gae_long", 128, 64, True, 0.995, 0.9, 2, 0.02)
    case("returns_plain", 20, 257, False, 0.99, 0.95, 3, 0.05)
    case("returns_one_step", 1, 5, True, 0.9, 1.0, 4, 0.5)
    bookkeeping("trainer_bookkeeping", 60, 130, 5)
    print("ok")
