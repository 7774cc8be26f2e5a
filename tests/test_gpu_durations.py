"""Batch-level duration telemetry (SRC/game.hh:98-101: thrust_durations, shot_durations, shot_intervals_invul / _vul) kept on
the device for every env (spacefortress_amd/durations.py, sfmi.h: sf_get_field_dev) against what the reference's own CPython
extension returned for the same key calls, tick by tick (tests/golden/getters)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
NAMES = ("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul")


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    from spacefortress_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsfmi.so not built: the GPU tests never fall back"
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


@pytest.mark.parametrize("name", ["youturn_hunter", "youturn_rapid_fire", "autoturn_allkeys", "testyouturn_random"])
def test_a_batchs_duration_vectors_equal_the_references(sfa, name):
    z = np.load(os.path.join(GOLDEN, "getters", name + ".npz"))
    gz = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    N = 5  # the same recorded game in every lane (same seed, spawn_stride 0 = the same spawn): five identical logs
    env = sfa.SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"], spawn_stride=0,
                       spawn_skip=meta["spawn_skip"])
    log = env.enable_durations(capacity=512)
    acts = torch.from_numpy(gz["actions"].astype(np.uint8)).to(env.device)
    for t in range(len(acts)):
        env.step_tensors(acts[t].repeat(N).contiguous())
        if t % 37 == 0 or t == len(acts) - 1:
            for k in NAMES:
                assert log.get(k)[1].tolist() == [int(z[k + "_len"][t])] * N, (t, k)
    assert int(log.dropped) == 0
    for k in NAMES:
        want = tuple(int(v) for v in z[k])
        for i in (0, N - 1):
            assert log.of(i, k) == want, (k, i)
    # (a field on the device equals the host's copy of it; the missile view is the host's)
    assert np.array_equal(env.get_field_tensor("fire_timer").cpu().numpy(), env.get_field("fire_timer"))
    assert np.array_equal(env.get_field_tensor("stats").cpu().numpy(), env.get_field("stats"))
    assert np.array_equal(env.get_field_tensor("ship_x").cpu().numpy(), env.get_field("ship_x"))
    with pytest.raises(Exception):
        env.get_field_tensor("missile_x")
    env.reset()
    assert all(int(log.get(k)[1].sum()) == 0 for k in NAMES)  # new Games: empty vectors (SRC/game.cpp:64-67)
    env.close()


def test_vectors_restart_with_every_new_game_and_overflow_is_counted(sfa):
    env = sfa.SFVecEnv(64, gametype="youturn", spawn_stride=1)
    log = env.enable_durations(capacity=4)
    env.set_field("time", np.full(64, 34 * 5285, np.int32))  # ten ticks from the end of the game
    fire, noop = torch.ones(64, dtype=torch.uint8, device=env.device), torch.zeros(64, dtype=torch.uint8, device=env.device)
    done_seen = False
    for t in range(12):
        _, _, done, _ = env.step_tensors(fire if t % 2 == 0 else noop)
        done_seen |= bool(done.any())
    assert done_seen
    assert int(log.dropped) > 0                      # five presses before the game ended, four slots
    assert log.get("shot_durations")[1].max() <= 1   # ... and the new Game's vectors started empty
    env.close()
