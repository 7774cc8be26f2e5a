"""The packed per-episode fields and their limits (sf_layout.h: SF_W_*; sfmi.h: sf_check_state).

The reference keeps its 13 statistics and its timers as plain ints (SRC/game.hh:29-43,60-68) and a bare SSF_Env keeps
ticking past game over until somebody calls reset() (ENV:246).  The device state carries those values in bit fields sized
for ONE episode.  What has to hold: as long as every value fits, a batch without auto-reset equals the reference however
many episodes' worth of ticks it plays; once one does not, the batch SAYS so (sf_check_state -> OverflowError) instead of
handing out wrapped numbers; and sf_set_field refuses values a field cannot hold.
"""
import os

import numpy as np
import pytest
import torch

from sfcompare import compare_state

STAT_BITS = np.array([8, 8, 8, 10, 16, 8, 16, 16, 16, 16, 16, 12, 12])  # sf_layout.h


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    from spacefortress_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsfmi.so not built: the GPU tests never fall back"
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


@pytest.mark.gpu
def test_four_episodes_without_reset_equal_the_oracle_or_raise_the_flag(sfa, oracle_mod):
    O = oracle_mod
    n, ep = 20, 5295
    rng = np.random.default_rng(2)
    env = sfa.SFVecEnv(n, gametype="youturn", auto_reset=False, spawn_stride=1, obs_dtype=torch.float64)
    orcs = [O.OracleEnv("youturn", spawn_skip=i) for i in range(n)]
    env.reset()
    for o in orcs:
        o.reset()
    flagged = False
    # 2, 4 and 5 episodes' worth of ticks, never reset: random play loses a ship through the big hexagon ten times per
    # thousand ticks, so the 8-bit counter of those (103, 204, then 245-261 with this seed) runs over in a few envs of
    # the last phase and in no env before
    for phase, T in enumerate((2 * ep, 2 * ep + 40, ep)):
        acts = rng.integers(0, 5, (T, n)).astype(np.uint8)
        dacts = torch.from_numpy(acts).to(env.device)
        rew = torch.empty((T, n), dtype=torch.int32, device=env.device)
        for t in range(T):
            o, r, d, i = env.step_tensors(dacts[t])
            rew[t] = r
            if t == 0 and phase == 0:
                assert not d.any()
        rew = rew.cpu().numpy()
        for i, o in enumerate(orcs):
            for t in range(T):
                _, orw, od, _ = o.step(int(acts[t, i]))
                assert rew[t, i] == orw, (phase, i, t)
            assert od  # past game over: `done` stays true, the game goes on (ENV:246)
        snaps = np.array([o.snapshot() for o in orcs])
        over = (snaps["stats"] >= (1 << STAT_BITS)[None, :]).any(1) | (snaps["time"] >= (1 << 24))
        assert np.abs(np.stack([snaps[k] for k in ("fire_timer", "thrust_timer", "left_timer", "right_timer")])).max() < 32768
        if over.any():
            with pytest.raises(OverflowError):
                env.check_state()
            with pytest.raises(OverflowError):
                env.check_state()  # sticky: the fields stay wrapped, every look says so (until reset(), below)
            flagged = True
        else:
            env.check_state()  # nothing outgrew its field: no flag
        ok = np.flatnonzero(~over)
        assert ok.size > 0
        bad = compare_state(env.state_dict(), snaps[ok], lanes=ok)
        assert not bad, (phase, bad)  # every env whose values still fit equals the reference, statistics included
        if phase < 2:
            assert not over.any() and snaps["stats"][:, 3].min() > 100 * (phase + 1)  # 60 ship deaths per episode, all counted
        else:
            assert 0 < over.sum() < n
    assert flagged, "the test is meant to reach the 8-bit death counters"
    env.reset()
    env.check_state()  # new games everywhere: nothing is wrapped any more
    env.close()


@pytest.mark.gpu
def test_a_key_timer_that_outgrows_int16_is_flagged(sfa):
    """No key edge for 32 768 ticks (a caller that holds NOOP for six episodes without reset)."""
    env = sfa.SFVecEnv(64, gametype="autoturn", auto_reset=False)
    env.reset()
    env.set_field("fire_timer", np.full(64, -32760, np.int32))
    noop = torch.zeros(64, dtype=torch.uint8, device=env.device)
    for _ in range(7):
        env.step_tensors(noop)
    env.check_state()
    assert (env.get_field("fire_timer") == -32767).all()
    env.step_tensors(noop)
    env.step_tensors(noop)
    with pytest.raises(OverflowError):
        env.check_state()
    # rewriting ONE packed field repairs that field, and the sticky count stays (other fields, other envs may have
    # wrapped -- one unrelated repair must not erase the evidence); a restored checkpoint -- load_state_dict: every packed field
    # of every env rewritten with values that fit -- starts it over (sf_clear_state_errors)
    env.set_field("fire_timer", np.full(64, -5, np.int32))
    with pytest.raises(OverflowError):
        env.check_state()
    env.load_state_dict(env.state_dict())
    env.check_state()
    assert (env.get_field("fire_timer") == -5).all()
    env.close()


@pytest.mark.gpu
def test_auto_resetting_batches_never_flag(sfa):
    env = sfa.SFVecEnv(256, gametype="youturn")
    env.reset()
    env.set_field("time", np.full(256, 34 * 5290, np.int32))
    a = torch.ones(256, dtype=torch.uint8, device=env.device)
    for _ in range(12):
        env.step_tensors(a)
    env.check_state()
    assert env.episode_stats()[0] == 256
    env.close()


@pytest.mark.gpu
def test_set_field_refuses_what_a_field_cannot_hold(sfa):
    env = sfa.SFVecEnv(8, gametype="youturn")
    env.reset()
    sd = env.state_dict()
    for name, bad in (("vlner", 5000), ("vlner", -1), ("prev_vlner", 4096), ("fire_timer", 40000), ("left_timer", -40000),
                      ("time", 1 << 24), ("ep_kills", 256), ("missile_mask", 1 << 20), ("spawn_cursor", 1 << 24)):
        v = sd[name].copy()
        v[3] = bad
        with pytest.raises(ValueError):
            env.set_field(name, v)
        env.set_field(name, sd[name])  # what fits is accepted
    env.set_field("vlner", np.full(8, 4095, np.int32))
    env.set_field("fire_timer", np.full(8, -32768, np.int32))
    st = sd["stats"].copy()
    st[0, 2] = 300  # big-hex deaths: 8 bits
    st[3, 2] = 300
    with pytest.raises(ValueError):
        env.set_field("stats", st)
    st = sd["stats"].copy()
    st[0, 1], st[1, 1], st[2, 1], st[3, 1] = 3, 2, 1, 7  # ship deaths are the sum of the three kinds, not a field of their own
    with pytest.raises(ValueError):
        env.set_field("stats", st)
    st[3, 1] = 6
    env.set_field("stats", st)
    assert np.array_equal(env.get_field("stats"), st)
    env.close()
