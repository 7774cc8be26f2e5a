"""The device-side trainer helpers (sf_compute_returns, sf_record_step, DeviceRollout) against fixtures
recorded from the reference's own rl/storage.py / rl/train.py:82-88 -- BIT-EXACT (float32, same operation
order, no FMA contraction) -- and an end-to-end rollout against a host-side replay of the same steps."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
TR = os.path.join(GOLDEN, "trainer")


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("name", ["returns_gae", "returns_gae_long", "returns_plain", "returns_one_step"])
def test_compute_returns_bit_exact(sfa, name):
    from spacefortress_amd import _lib
    z = np.load(os.path.join(TR, name + ".npz"))
    dev = torch.device("cuda")
    T, n = z["rewards"].shape
    rewards, vp, masks = (torch.from_numpy(z[k]).to(dev) for k in ("rewards", "value_preds_in", "masks"))
    nv = torch.from_numpy(z["next_value"]).to(dev)
    ret = torch.zeros(T + 1, n, device=dev)
    _lib.check(_lib.lib().sf_compute_returns(T, n, _p(rewards), _p(vp), _p(masks), _p(nv), _p(ret), int(z["use_gae"]),
                                             float(z["gamma"]), float(z["tau"]), None))
    torch.cuda.synchronize()
    assert np.array_equal(ret.cpu().numpy(), z["returns"])
    assert np.array_equal(vp.cpu().numpy(), z["value_preds_out"])


def test_record_step_bit_exact(sfa):
    from spacefortress_amd import _lib
    z = np.load(os.path.join(TR, "trainer_bookkeeping.npz"))
    dev = torch.device("cuda")
    T, n = z["rewards"].shape
    ep, fin = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    r_out, m_out = torch.empty(n, device=dev), torch.empty(n, device=dev)
    for t in range(T):
        r, d = torch.from_numpy(z["rewards"][t]).to(dev), torch.from_numpy(z["done"][t]).to(dev)
        _lib.check(_lib.lib().sf_record_step(n, _p(r), _p(d), _p(r_out), _p(m_out), _p(ep), _p(fin), None, 0, None, None))
        assert np.array_equal(m_out.cpu().numpy(), z["masks"][t])
        assert np.array_equal(r_out.cpu().numpy(), z["rewards"][t].astype(np.float32))
        assert np.array_equal(ep.cpu().numpy(), z["episode_rewards"][t])
        assert np.array_equal(fin.cpu().numpy(), z["final_rewards"][t])
    assert _lib.lib().sf_record_step(0, None, None, None, None, None, None, None, 0, None, None) < 0


def test_device_rollout_end_to_end(sfa):
    """A 12-step rollout collected entirely on the device equals the same steps taken one by one with the
    trainer's host-side bookkeeping; returns equal the oracle's restatement of rl/storage.py."""
    from oracle import trainer_np as TN
    N, T = 300, 12
    env = sfa.SFVecEnv(N, gametype="autoturn", spawn_stride=1)
    twin = sfa.SFVecEnv(N, gametype="autoturn", spawn_stride=1)
    for e in (env, twin):  # late in the episode so that `done` happens inside the rollout
        e.set_field("time", np.full(N, 34 * 5289, np.int32))
    ro = sfa.DeviceRollout(env, T)
    g = torch.Generator(device=env.device).manual_seed(0)
    ro.observations[0].copy_(twin.step_tensors(torch.zeros(N, dtype=torch.uint8, device=env.device))[0])
    env.step_tensors(torch.zeros(N, dtype=torch.uint8, device=env.device))
    ep, fin = np.zeros(N, np.float32), np.zeros(N, np.float32)
    kills = 0
    for t in range(T):
        a = torch.randint(0, 3, (N, 1), device=env.device, generator=g)
        v = torch.randn(N, 1, device=env.device, generator=g)
        ro.step(t, a, value_pred=v, action_log_prob=-v)
        o, r, d, i = twin.step_tensors(a.view(-1))
        _, m, ep, fin = TN.record_step(r.cpu().numpy(), d.cpu().numpy().astype(bool), ep, fin)
        kills += int(i.sum())
        assert torch.equal(ro.observations[t + 1], o)
        assert np.array_equal(ro.rewards[t, :, 0].cpu().numpy(), r.cpu().numpy().astype(np.float32))
        assert np.array_equal(ro.masks[t + 1, :, 0].cpu().numpy(), m)
        assert torch.equal(ro.actions[t], a) and torch.equal(ro.value_preds[t], v)
    assert (ro.masks[1:] == 0).any()
    assert np.array_equal(ro.episode_rewards[:, 0].cpu().numpy(), ep) and np.array_equal(ro.final_rewards[:, 0].cpu().numpy(), fin)
    assert int(ro.num_destruction) == kills
    nv = torch.randn(N, 1, device=env.device, generator=g)
    vp_in = ro.value_preds[..., 0].cpu().numpy().copy()
    ro.compute_returns(nv, True, 0.99, 0.95)
    want, _ = TN.compute_returns(ro.rewards[..., 0].cpu().numpy(), vp_in, ro.masks[..., 0].cpu().numpy(),
                                 nv[:, 0].cpu().numpy(), True, 0.99, 0.95)
    assert np.array_equal(ro.returns[..., 0].cpu().numpy(), want)
    last = ro.observations[-1].clone()
    ro.after_update()
    assert torch.equal(ro.observations[0], last) and torch.equal(ro.masks[0], ro.masks[-1])
    env.close()
    twin.close()


def test_step_record_equals_step_then_record(sfa):
    """sf_step_record (bookkeeping in the step kernel's epilogue) against sf_step followed by sf_record_step,
    both bit-exact images of rl/train.py:82-88; also with an image observation and uint8 / int64 actions."""
    from spacefortress_amd import _lib
    L = _lib.lib()
    N, T = 700, 40
    rng = np.random.default_rng(6)
    for obs_type, adt in (("features", torch.uint8), ("image", torch.int64)):
        a_env = sfa.SFVecEnv(N, gametype="youturn", obs_type=obs_type, spawn_stride=2)
        b_env = sfa.SFVecEnv(N, gametype="youturn", obs_type=obs_type, spawn_stride=2)
        for e in (a_env, b_env):
            e.set_field("time", np.full(N, 34 * 5270, np.int32))
        dev = a_env.device
        z = lambda dt=torch.float32: torch.zeros(N, dtype=dt, device=dev)
        ep1, fin1, ep2, fin2 = z(), z(), z(), z()
        r1, m1, r2, m2 = z(), z(), z(), z()
        act1, act2 = z(torch.int64), z(torch.int64)
        for t in range(T):
            a = torch.from_numpy(rng.integers(0, 5, N)).to(dev).to(adt)
            o1, rew, done, info = a_env.step_tensors(a)
            _lib.check(L.sf_record_step(N, _p(rew), _p(done), _p(r1), _p(m1), _p(ep1), _p(fin1), _p(a), a.element_size(), _p(act1), None))
            o2, rw2, dn2, in2 = b_env._alloc()
            _lib.check(L.sf_step_record(b_env._h, _p(a), a.element_size(), _p(o2), _p(rw2), _p(dn2), _p(in2), _p(r2), _p(m2),
                                        _p(ep2), _p(fin2), _p(act2), None))
            for x, y in ((o1, o2), (rew, rw2), (done, dn2), (info, in2), (r1, r2), (m1, m2), (ep1, ep2), (fin1, fin2), (act1, act2)):
                assert torch.equal(x, y), (obs_type, t)
        assert float(m1.min()) == 0.0 or (ep1 != 0).any()
        a_env.close()
        b_env.close()


def test_device_rollout_over_vecnormalize(sfa):
    """The trainer's configuration for 1-D observations (rl/train.py:35-36): the storage receives NORMALISED
    observations and rewards, and the episode bookkeeping runs on the normalised rewards.  Against a twin
    SFVecNormalize stepped one call at a time + the numpy restatement of the bookkeeping."""
    from oracle import trainer_np as TN
    N, T = 500, 10
    mk = lambda: sfa.SFVecNormalize(sfa.SFVecEnv(N, gametype="youturn", spawn_stride=1))
    env, twin = mk(), mk()
    ro = sfa.DeviceRollout(env, T)
    o0 = ro.reset()
    assert torch.allclose(o0, twin.reset(), atol=1e-6)
    for e in (env, twin):  # late in the episode, so that `done` happens inside the rollout
        e.venv.set_field("time", np.full(N, 34 * 5288, np.int32))
    g = torch.Generator(device=o0.device).manual_seed(1)
    ep, fin = np.zeros(N, np.float32), np.zeros(N, np.float32)
    for t in range(T):
        a = torch.randint(0, 5, (N,), device=o0.device, generator=g)
        obs, rew, mask = ro.step(t, a)
        o2, r2, d2, i2 = twin.step_tensors(a)
        assert torch.allclose(obs, o2, atol=1e-6) and torch.allclose(rew[:, 0], r2, atol=1e-6)
        r_np = rew[:, 0].cpu().numpy()
        m = np.where(d2.cpu().numpy().astype(bool), np.float32(0), np.float32(1))
        ep = ep + r_np
        fin = fin * m + (np.float32(1) - m) * ep
        ep = ep * m
        assert np.array_equal(mask[:, 0].cpu().numpy(), m)
    assert np.array_equal(ro.episode_rewards[:, 0].cpu().numpy(), ep) and np.array_equal(ro.final_rewards[:, 0].cpu().numpy(), fin)
    assert (ro.masks[1:] == 0).any()
    st1, st2 = env.state_dict()["stats"], twin.state_dict()["stats"]
    assert np.allclose(st1, st2, rtol=1e-12)
    env.close()
    twin.close()


def test_device_rollout_with_frame_stack(sfa):
    """Image observations with num_stack = 4 (BASELINE configs[4]; rl/train.py:38-39,51-56,92-98): every step's
    stored observation is the previous one shifted by a frame, zeroed for finished envs, with the new frame
    last -- against FrameStack (already checked against the trainer's own update) on a twin batch."""
    N, T, S = 96, 14, 4
    env = sfa.SFVecEnv(N, gametype="autoturn", obs_type="image", spawn_stride=1)
    twin = sfa.SFVecEnv(N, gametype="autoturn", obs_type="image", spawn_stride=1)
    ro = sfa.DeviceRollout(env, T, num_stack=S)
    fs = sfa.FrameStack(twin, S)
    assert ro.observations.shape == (T + 1, N, S, 84, 84) and ro.observations.dtype == torch.uint8
    assert torch.equal(ro.reset(), fs.reset())
    for e in (env, twin):
        e.set_field("time", np.full(N, 34 * 5287, np.int32))
    g = torch.Generator(device=env.device).manual_seed(2)
    saw_done = False
    for t in range(T):
        a = torch.randint(0, 3, (N,), device=env.device, generator=g, dtype=torch.uint8)
        obs, rew, mask = ro.step(t, a)
        r2, d2, i2 = fs.step(a)
        assert torch.equal(obs, fs.stacked()), t
        assert torch.equal(rew[:, 0], r2.float()) and torch.equal(mask[:, 0] == 0, d2)
        saw_done = saw_done or bool(d2.any())
    assert saw_done
    ro.after_update()
    assert torch.equal(ro.observations[0], fs.stacked())
    with pytest.raises(ValueError):
        sfa.DeviceRollout(sfa.SFVecEnv(4), 2, num_stack=4)
    env.close()
    twin.close()


def test_ppo_generators_cover_the_rollout(sfa):
    """feed_forward_generator / recurrent_generator (rl/storage.py:66-122): shapes of the reference's tuples,
    every transition sampled exactly once, rows consistent across the tensors of a minibatch."""
    N, T = 24, 6
    env = sfa.SFVecEnv(N, gametype="youturn", spawn_stride=1)
    ro = sfa.DeviceRollout(env, T)
    ro.reset()
    for t in range(T):
        a = torch.randint(0, 5, (N,), device=env.device)
        ro.step(t, a, value_pred=torch.full((N, 1), float(t), device=env.device))
    ro.compute_returns(torch.zeros(N, 1, device=env.device), True, 0.99, 0.95)
    ro.returns[:-1] = torch.arange(T * N, device=env.device, dtype=torch.float32).view(T, N, 1)  # a unique tag per transition
    adv = ro.returns[:-1] * 2
    seen = []
    for obs, st, act, ret, msk, logp, ad in ro.feed_forward_generator(adv, 4):
        assert obs.shape == (T * N // 4, 19) and act.shape == (T * N // 4, 1) and ret.shape == msk.shape == logp.shape == ad.shape
        assert torch.equal(ad, ret * 2)
        tags = ret[:, 0].long()
        assert torch.equal(obs, ro.observations[:-1].reshape(T * N, 19)[tags])
        seen.append(tags)
    assert torch.equal(torch.sort(torch.cat(seen)).values, torch.arange(T * N, device=env.device))
    seen = []
    for obs, st, act, ret, msk, logp, ad in ro.recurrent_generator(adv, 3):
        assert obs.shape == (T * (N // 3), 19) and torch.equal(ad, ret * 2)
        tags = ret[:, 0].long().view(N // 3, T)
        assert bool((tags[:, 1:] - tags[:, :-1] == N).all())  # consecutive steps of one env
        seen.append(tags.reshape(-1))
    assert torch.equal(torch.sort(torch.cat(seen)).values, torch.arange(T * N, device=env.device))
    env.close()
