"""Replay files (spacefortress_amd/replay.py): the on-disk record of a batch's game -- parameters, uint8 [T, N] actions,
what they produced -- and the device-side player that verifies it.  The reference keeps nothing but per-tick dump()
strings (SRC/game.cpp:519-576) and cannot read them back; the engine is deterministic given (preset, seed, spawn offset,
actions), which is all a file has to hold."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names


def test_file_round_trip_without_a_gpu(tmp_path):
    from spacefortress_amd.replay import FORMAT, Replay, state_digest

    rng = np.random.default_rng(0)
    meta = {"gametype": "youturn", "action_set": 1, "seed": 1, "spawn_skip": 3, "spawn_stride": 1, "n_envs": 7, "auto_reset": True,
            "tick_ms": 34, "build_id": "x"}
    acts = rng.integers(0, 5, (50, 7)).astype(np.uint8)
    r = Replay(meta, acts, returns=np.arange(7), kills=np.zeros(7), dones=np.ones(7), digest="ab" * 32)
    p = r.save(str(tmp_path / "game.sfreplay"))
    assert os.path.exists(p) and not os.path.exists(p + ".npz")
    q = Replay.load(p)
    assert q.meta == dict(meta, steps=50, format=FORMAT) and np.array_equal(q.actions, acts)
    assert np.array_equal(q.returns, np.arange(7)) and q.digest == "ab" * 32 and q.dones.dtype == np.int64
    with pytest.raises(ValueError):
        Replay(meta, acts[:, :3])
    # a file of another format version is refused, not misread
    z = dict(np.load(p))
    z["meta"] = np.array(json.dumps(dict(meta, steps=50, format=FORMAT + 1)))
    with open(tmp_path / "future.sfreplay", "wb") as f:
        np.savez_compressed(f, **z)
    with pytest.raises(ValueError):
        Replay.load(str(tmp_path / "future.sfreplay"))
    sd = {"a": np.arange(4, dtype=np.int32), "b": np.ones((2, 3))}
    assert state_digest(sd) == state_digest({"b": np.ones((2, 3)), "a": np.arange(4, dtype=np.int32)})
    assert state_digest(sd) != state_digest({"a": np.arange(4, dtype=np.int64), "b": np.ones((2, 3))})


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names())
def test_goldens_as_replay_files(tmp_path, name):
    """Every golden vector (recorded from the REAL reference engine, tests/golden/make_golden.py) written as a replay file,
    read back and played on the device: reward / done / info of every step are the reference's, the per-env sums verify."""
    from spacefortress_amd.replay import Replay, build_id

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    rmeta = {"gametype": meta["gametype"], "action_set": meta["action_set"], "seed": meta["seed"], "spawn_skip": meta["spawn_skip"],
             "spawn_stride": 0, "n_envs": 2, "auto_reset": True, "tick_ms": 34, "build_id": build_id()}
    acts = np.repeat(z["actions"][:, None], 2, 1).astype(np.uint8)
    want = {"returns": np.repeat(z["reward"].astype(np.int64).sum(), 2), "kills": np.repeat(z["info"].astype(np.int64).sum(), 2),
            "dones": np.repeat(z["done"].astype(np.int64).sum(), 2)}
    path = Replay(rmeta, acts, **want).save(str(tmp_path / (name + ".sfreplay")))
    rp = Replay.load(path)
    out = rp.run(keep_steps=True, chunk=97)
    for lane in range(2):
        assert np.array_equal(out["reward"][:, lane], z["reward"])
        assert np.array_equal(out["done"][:, lane], z["done"].astype(bool))
        assert np.array_equal(out["info"][:, lane], z["info"].astype(bool))
    # the final state is the reference's last snapshot (unless the last step ended the episode: the lane was reset)
    if not z["done"][-1] and (len(acts) % meta["snap_every"]) == 0:
        from sfcompare import compare_state
        assert not compare_state(out["env"].state_dict(), np.repeat(z["snaps"][-1][None], 2))
    out["env"].close()


@pytest.mark.gpu
def test_record_save_load_replay(tmp_path):
    """A recording over every way of stepping -- given actions one step at a time, a fused rollout, actions drawn on the
    device (one step, and K fused) -- saved, loaded and played back through sf_rollout alone: the same returns, kills,
    episode ends and final state, bit for bit; a file with one action changed is caught."""
    import torch

    from sfscript import open_loop_actions
    from spacefortress_amd import SFVecEnv
    from spacefortress_amd.replay import Replay, ReplayMismatch

    N = 512
    rng = np.random.default_rng(3)
    env = SFVecEnv(N, gametype="autoturn", spawn_stride=1, spawn_skip=5, seed=7)
    env.start_recording()
    acts = torch.from_numpy(open_loop_actions("hunter", (700, N), env.n_actions, rng, phase=rng.integers(0, 96, N))).to(env.device)
    for t in range(300):
        env.step_tensors(acts[t])
    env.rollout(acts[300:700])
    env.seed_actions(42)
    for t in range(40):
        env.step_sampled()
    env.rollout_sampled(60, want_obs=False, want_actions=False)
    env.step(np.zeros(N, np.int64))  # the host API records too
    rp = env.save_replay(str(tmp_path / "run.sfreplay"))
    assert rp.actions.shape == (801, N) and rp.kills.sum() > 50 and rp.meta["seed"] == 7 and rp.meta["spawn_skip"] == 5
    with pytest.raises(RuntimeError):
        env.start_recording()  # not a new batch any more
    with pytest.raises(RuntimeError):
        env.reset()            # ... and a reset ends a recording loudly
    env.close()
    back = SFVecEnv.load_replay(str(tmp_path / "run.sfreplay"))
    out = back.run(chunk=128)
    assert np.array_equal(out["returns"], rp.returns) and out["digest"] == rp.digest
    out["env"].close()
    bad = Replay(back.meta, back.actions.copy(), back.returns, back.kills, back.dones, back.digest)
    bad.actions[500, 17] = (bad.actions[500, 17] + 1) % 3
    with pytest.raises(ReplayMismatch):
        bad.run(chunk=256)


@pytest.mark.gpu
def test_the_trainers_wrappers_record_too(tmp_path):
    """FrameStack.step, DeviceRollout.step and SFVecNormalize's fused step call the C ABI themselves; they report every launch to
    the batch's one hook (SFVecEnv._stepped), so a recording made on the image / PPO paths holds the actions played and
    verifies when replayed -- and a batch stepped only through a wrapper is no longer a new one."""
    import torch

    from spacefortress_amd import DeviceRollout, FrameStack, SFVecEnv, SFVecNormalize

    N = 256
    g = torch.Generator().manual_seed(5)
    # a frame stack on an image batch
    env = SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=1, seed=3)
    env.start_recording()
    fs = FrameStack(env, 4)
    acts = torch.randint(0, env.n_actions, (120, N), generator=g, dtype=torch.uint8).to(env.device)
    for t in range(120):
        fs.step(acts[t])
    rp = env.save_replay(str(tmp_path / "stack.sfreplay"))
    assert rp.actions.shape == (120, N) and np.array_equal(rp.actions, acts.cpu().numpy())
    with pytest.raises(RuntimeError):
        fs.reset()  # a reset ends a recording loudly, through the wrapper as well
    env.close()
    out = SFVecEnv.load_replay(str(tmp_path / "stack.sfreplay")).run(chunk=64)
    assert np.array_equal(out["returns"], rp.returns) and out["digest"] == rp.digest
    out["env"].close()
    # the PPO path: a rollout buffer over a normalised features batch
    env = SFVecEnv(N, gametype="autoturn", obs_type="features", spawn_stride=1, seed=9)
    env.start_recording()
    ro = DeviceRollout(SFVecNormalize(env), num_steps=16)
    acts = torch.randint(0, env.n_actions, (48, N), generator=g, dtype=torch.int64).to(env.device)
    for t in range(48):
        ro.step(t % 16, acts[t])
    rp = env.save_replay(str(tmp_path / "ppo.sfreplay"))
    assert rp.actions.shape == (48, N) and np.array_equal(rp.actions, acts.cpu().numpy().astype(np.uint8))
    env.close()
    out = SFVecEnv.load_replay(str(tmp_path / "ppo.sfreplay")).run(chunk=16)
    assert np.array_equal(out["returns"], rp.returns) and out["digest"] == rp.digest
    out["env"].close()
    # stepped through a wrapper only: not a new batch any more
    env = SFVecEnv(N, gametype="youturn", obs_type="image")
    FrameStack(env, 2).step(torch.zeros(N, dtype=torch.uint8, device=env.device))
    with pytest.raises(RuntimeError):
        env.start_recording()
    env.close()
