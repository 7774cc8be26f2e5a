"""The numpy restatement of the trainer arithmetic against fixtures recorded from the reference's own
rl/storage.py and rl/train.py:82-88 (tests/golden/trainer/).  No GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

TR = os.path.join(GOLDEN, "trainer")


@pytest.mark.parametrize("name", ["returns_gae", "returns_gae_long", "returns_plain", "returns_one_step"])
def test_compute_returns_restatement_is_bit_exact(name):
    from oracle import trainer_np as T
    z = np.load(os.path.join(TR, name + ".npz"))
    ret, vp = T.compute_returns(z["rewards"], z["value_preds_in"], z["masks"], z["next_value"], bool(z["use_gae"]),
                                float(z["gamma"]), float(z["tau"]))
    assert np.array_equal(ret, z["returns"])
    assert np.array_equal(vp, z["value_preds_out"])


def test_bookkeeping_restatement_is_bit_exact():
    from oracle import trainer_np as T
    z = np.load(os.path.join(TR, "trainer_bookkeeping.npz"))
    n = z["rewards"].shape[1]
    ep, fin = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for t in range(z["rewards"].shape[0]):
        r, m, ep, fin = T.record_step(z["rewards"][t], z["done"][t].astype(bool), ep, fin)
        assert np.array_equal(m, z["masks"][t]) and np.array_equal(ep, z["episode_rewards"][t])
        assert np.array_equal(fin, z["final_rewards"][t])
