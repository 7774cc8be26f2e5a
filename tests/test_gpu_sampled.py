"""Actions drawn on the device, inside the step launch (sfmi.h: sf_step_sampled / sf_rollout_sampled).

The reference's loop gets fresh actions every step (rl/train.py:76-80); BASELINE.json's rollout is a random-action one
and SURVEY 8(d) counts "action generation on device" into the metric.  What has to hold:
  * the generator is the published Philox4x32-10 (Random123 known answers, checked on the CPU);
  * lane i plays floor(x * n_actions / 2^32) with x = Philox(key = seed, counter = (first_lane + i, tick));
  * the game does not care where an action came from: replaying the recorded actions through sf_step, and through
    the CPU oracle, gives the same rewards / done / info / observations / state;
  * the tick counter lives on the device: a captured HIP graph draws NEW actions on every replay;
  * shards seeded with their first lane draw what the one big batch draws.
"""
import os

import numpy as np
import pytest
import torch

from sfcompare import compare_state, obs_close

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c, k):
    """Vectorised Philox4x32-10: c = 4 uint32 arrays (counter), k = 2 (key); returns the 4 output words."""
    c = [np.asarray(x, np.uint64) for x in np.broadcast_arrays(*c)]
    k = [np.uint64(k[0]), np.uint64(k[1])]
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & m32]
        k = [(k[0] + np.uint64(W0)) & m32, (k[1] + np.uint64(W1)) & m32]
    return [x.astype(np.uint32) for x in c]


def sampled_actions(seed, first_lane, n, tick0, ticks, n_actions):
    lanes = (np.arange(n, dtype=np.uint64) + np.uint64(first_lane)) & np.uint64(0xFFFFFFFF)
    out = np.empty((ticks, n), np.uint8)
    for t in range(ticks):
        x = philox4x32_10((lanes, np.uint64(tick0 + t), np.uint64(0), np.uint64(0)), (seed & 0xFFFFFFFF, seed >> 32))[0]
        out[t] = ((x.astype(np.uint64) * np.uint64(n_actions)) >> np.uint64(32)).astype(np.uint8)
    return out


def test_philox_known_answers():
    """Random123's kat_vectors for philox4x32-10: the numpy restatement the GPU tests compare against is the published one."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        got = philox4x32_10([np.array([x], np.uint64) for x in c], k)
        assert tuple(int(g[0]) for g in got) == want
    a = sampled_actions(7, 0, 100000, 0, 4, 5)
    assert a.min() == 0 and a.max() == 4
    assert np.abs(np.bincount(a.ravel(), minlength=5) / a.size - 0.2).max() < 0.005


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    from spacefortress_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsfmi.so not built: the GPU tests never fall back"
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("gametype,n", [("youturn", 1000), ("autoturn", 4096)])
def test_sampled_steps_are_philox_and_equal_replayed_steps_and_the_oracle(sfa, oracle_mod, gametype, n):
    O = oracle_mod
    T, seed = 700, 0x1234567855AA
    a = sfa.SFVecEnv(n, gametype=gametype, spawn_stride=1, obs_dtype=torch.float64)
    b = sfa.SFVecEnv(n, gametype=gametype, spawn_stride=1, obs_dtype=torch.float64)
    a.seed_actions(seed)
    a.reset()
    b.reset()
    acts = torch.empty((T, n), dtype=torch.uint8, device=a.device)
    outs = []
    for t in range(T):
        o, r, d, i = a.step_sampled(actions_out=acts[t])
        outs.append((o.clone(), r.clone(), d.clone(), i.clone()))
    host = acts.cpu().numpy()
    assert np.array_equal(host, sampled_actions(seed, 0, n, 0, T, a.n_actions))
    orc = O.OracleVecEnv(gametype, n, spawn_stride=1)
    orc.reset()
    for t in range(T):
        o2, r2, d2, i2 = b.step_tensors(acts[t])
        o1, r1, d1, i1 = outs[t]
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(i1, i2), t
        oo, orw, od, oi = orc.step(host[t].astype(np.int32))
        assert np.array_equal(r1.cpu().numpy(), orw) and np.array_equal(d1.cpu().numpy().astype(bool), od), t
        assert np.array_equal(i1.cpu().numpy().astype(bool), oi), t
        assert obs_close(o1.cpu().numpy(), oo, True).all(), t
    assert not compare_state(a.state_dict(), orc.snapshots())
    a.check_actions()
    # seeding again restarts the sequence at tick 0
    a.seed_actions(seed)
    a.step_sampled(actions_out=acts[0])
    assert np.array_equal(acts[0].cpu().numpy(), host[0])
    a.close()
    b.close()


@pytest.mark.gpu
def test_fused_sampled_rollout_equals_sampled_steps(sfa):
    n, K, seed = 4096, 48, 99
    a = sfa.SFVecEnv(n, spawn_stride=1)
    b = sfa.SFVecEnv(n, spawn_stride=1)
    for e in (a, b):
        e.seed_actions(seed, first_lane=123456)
        e.reset()
    acts = torch.empty((2 * K, n), dtype=torch.uint8, device=a.device)
    single = []
    for t in range(2 * K):
        o, r, d, i = a.step_sampled(actions_out=acts[t])
        single.append((o.clone(), r.clone()))
    for half in range(2):  # two launches: the tick counter carries over from one to the next
        obs, rew, done, info, fa = b.rollout_sampled(K)
        assert torch.equal(fa, acts[half * K:(half + 1) * K])
        for t in range(K):
            assert torch.equal(obs[t], single[half * K + t][0]) and torch.equal(rew[t], single[half * K + t][1])
    assert np.array_equal(acts.cpu().numpy(), sampled_actions(seed, 123456, n, 0, 2 * K, 5))
    sa, sb = a.state_dict(), b.state_dict()
    assert all(np.array_equal(sa[k], sb[k]) for k in ("ship_x", "ship_y", "points", "stats", "time", "missile_mask"))
    a.close()
    b.close()


@pytest.mark.gpu
def test_a_captured_graph_draws_new_actions_on_every_replay(sfa):
    n, K, seed = 2048, 6, 5
    env = sfa.SFVecEnv(n, reuse_buffers=True)
    env.seed_actions(seed)
    env.reset()
    acts = torch.zeros((K, n), dtype=torch.uint8, device=env.device)
    rows = [acts[k] for k in range(K)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for k in range(K):
                env.step_sampled(actions_out=rows[k])
    torch.cuda.current_stream(env.device).wait_stream(side)
    torch.cuda.synchronize()
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(acts.cpu().numpy(), sampled_actions(seed, 0, n, rep * K, K, 5)), rep
    env.close()


@pytest.mark.gpu
def test_shards_seeded_with_their_first_lane_draw_what_the_whole_batch_draws(sfa):
    n, T, seed = 1024, 40, 77
    whole = sfa.SFVecEnv(2 * n, spawn_stride=1)
    parts = [sfa.SFVecEnv(n, spawn_stride=1, spawn_skip=r * n) for r in range(2)]
    whole.seed_actions(seed)
    whole.reset()
    for r, p in enumerate(parts):
        p.seed_actions(seed, first_lane=r * n)
        p.reset()
    for t in range(T):
        ow, rw, dw, iw = whole.step_sampled()
        for r, p in enumerate(parts):
            o, rr, d, i = p.step_sampled()
            assert torch.equal(o, ow[r * n:(r + 1) * n]) and torch.equal(rr, rw[r * n:(r + 1) * n]), (t, r)
    whole.close()
    for p in parts:
        p.close()
