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


@pytest.mark.parametrize("wrapper", ["framestack", "rollout", "rollout_stack", "vecnormalize", "rollout_vecnormalize"])
def test_the_trainers_wrappers_keep_the_log_too(sfa, wrapper):
    """the wrappers that call the C ABI themselves (FrameStack.step, DeviceRollout.step, SFVecNormalize's fused
    step) report to the same two hooks as step_tensors (SFVecEnv._before_step / _stepped): the log they leave equals the
    reference's vectors, as for plain steps.  Fused and sampled launches are refused while a log is enabled; set_field and
    load_state_dict start it over."""
    name = "youturn_rapid_fire"
    z = np.load(os.path.join(GOLDEN, "getters", name + ".npz"))
    gz = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    N = 64
    image = wrapper in ("framestack", "rollout_stack")
    env = sfa.SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"], spawn_stride=0,
                       spawn_skip=meta["spawn_skip"], obs_type="image" if image else "features")
    log = env.enable_durations(capacity=512)
    acts = torch.from_numpy(gz["actions"].astype(np.uint8)).to(env.device)
    T = len(acts)
    if wrapper == "framestack":
        w = sfa.FrameStack(env, 4)
        step = lambda t: w.step(acts[t].repeat(N).contiguous())
    elif wrapper == "vecnormalize":
        w = sfa.SFVecNormalize(env)
        step = lambda t: w.step_tensors(acts[t].repeat(N).contiguous())
    else:
        inner = sfa.SFVecNormalize(env) if wrapper == "rollout_vecnormalize" else env
        w = sfa.DeviceRollout(inner, num_steps=T, num_stack=4 if wrapper == "rollout_stack" else 1)
        step = lambda t: w.step(t, acts[t].repeat(N).contiguous())
    # (no reset: the recorded run starts from the Game sf_create made; the wrappers' buffers start as they are)
    for t in range(T):
        step(t)
    assert int(log.dropped) == 0
    assert sum(len(z[k]) for k in NAMES) > 20  # (the run does push: dozens of shots)
    for k in NAMES:
        want = tuple(int(v) for v in z[k])
        for i in (0, N - 1):
            assert log.of(i, k) == want, (wrapper, k, i)
    for call in (lambda: env.rollout(acts[:4, None].repeat(1, N).contiguous()), lambda: env.step_sampled(), lambda: env.rollout_sampled(3)):
        with pytest.raises(RuntimeError):
            call()
    env.set_field("vlner", np.zeros(N, np.int32))
    assert all(int(log.get(k)[1].sum()) == 0 for k in NAMES)
    step(0) if wrapper in ("framestack", "vecnormalize") else env.step_tensors(acts[0].repeat(N).contiguous())
    sd = env.state_dict()
    env.load_state_dict(sd)
    assert all(int(log.get(k)[1].sum()) == 0 for k in NAMES)
    env.close()
