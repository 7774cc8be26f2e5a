"""The `Game`-shaped single-lane view (spacefortress.core.Game) against the REAL reference: what every attribute of the
reference's own CPython extension returned tick by tick (tests/golden/getters, recorded from `_spacefortress` built from
SRC/pymodule.cpp: make_getters_golden.py), the recorded runs of tests/golden (state, raw engine reward of every tick) and the
dumpState() strings recorded from oracle/_ref (tests/golden/telemetry/dumps.npz), character for character."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def drive(g, keys, youturn):
    import spacefortress.core as sf
    (g.press_key if keys & 1 else g.release_key)(sf.FIRE_KEY)  # ENV:213-229
    (g.press_key if keys & 2 else g.release_key)(sf.THRUST_KEY)
    if youturn:
        (g.press_key if keys & 4 else g.release_key)(sf.LEFT_KEY)
        (g.press_key if keys & 8 else g.release_key)(sf.RIGHT_KEY)


@pytest.mark.parametrize("name", ["autoturn_destroy", "youturn_deaths", "youturn_rapid_fire"])
def test_game_shim_replays_the_reference(name):
    import spacefortress.core as sf
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    tele = np.load(os.path.join(GOLDEN, "telemetry", "dumps.npz"))
    dumps = tele[name]
    meta = json.loads(str(z["meta"]))
    youturn = meta["gametype"] in ("youturn", "test-youturn")
    g = sf.Game(meta["gametype"], width=90, height=92, viewport=(130, 80, 450, 460), lw=3, grayscale=True)
    assert g.dump() == dumps[0].decode()
    mismatched_dumps = 0
    for t, keys in enumerate(z["keys"]):
        drive(g, int(keys), youturn)
        r = g.step_one_tick(34)
        s = z["snaps"][t]
        assert r == int(z["eng_reward"][t]), t          # Game::stepOneTick's own return value
        assert g.time == int(s["time"]) and g.tick == int(s["tick"])
        assert g.ship_alive == bool(s["ship_alive"]) and g.fortress_alive == bool(s["fort_alive"])
        assert (g.ship_x, g.ship_y, g.ship_vx, g.ship_vy, g.ship_angle) == tuple(float(s[k]) for k in
                                                                               ("ship_x", "ship_y", "ship_vx", "ship_vy", "ship_angle"))
        assert g.fortress_angle == float(s["fort_angle"]) and g.vulnerability == int(s["vlner"])
        assert g.points == float(s["points"]) and g.raw_points == float(s["raw_points"])
        assert g.thrust_flag == bool(s["thrust_flag"]) and g.turn_flag == int(s["turn_flag"])
        assert g.stats[:13] == tuple(int(v) for v in s["stats"])
        assert g.timers == tuple(int(s[k]) for k in ("fire_timer", "thrust_timer", "left_timer", "right_timer"))
        assert len(g.missiles) == int(s["missile_alive"].sum()) and g.shells == g.missiles
        if s["ship_alive"]:
            assert abs(g.aim - float(s["aim"])) < 1e-9 and abs(g.vdir - float(s["vdir"])) < 1e-9
        assert g.is_game_over() == bool(z["done"][t])
        if g.dump() != dumps[t + 1].decode():
            mismatched_dumps += 1
            if mismatched_dumps == 1:
                first = (t, g.dump(), dumps[t + 1].decode())
    assert mismatched_dumps == 0, first
    g.draw()
    assert len(g.pb_pixels) == 92 * 90 * 4 and g.pb_width == 90 and g.pb_height == 92
    assert g.config("bigHex") == 200 and g.bighex == 200 and g.smallhex == 40
    with pytest.raises(ValueError):
        g.config("nonsense")
    for key in ("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul"):
        assert getattr(g, key) == tuple(int(v) for v in tele[name + "__" + key]), key
    with pytest.raises(AttributeError):
        g.max_points
    with pytest.raises(ValueError):
        g.step_one_tick(33)
    g.close()


GETTER_RUNS = ["autoturn_destroy", "youturn_deaths", "youturn_rapid_fire", "youturn_allkeys", "autoturn_allkeys",
               "autoturn_small_hex", "youturn_hunter", "testyouturn_random", "testautoturn_random"]


@pytest.mark.parametrize("name", GETTER_RUNS)
def test_game_shim_equals_the_real_extensions_getters(name):
    """Every attribute of SRC/pymodule.cpp:372-411 but the three undefined ones, after every tick, as the reference's own
    extension module returned it for the same key calls (ENV:213-231): floats to the bit, tuples element by element --
    `shells` is the reference's walk over the MISSILES (:131-134), `stats` the 15-tuple (:78-96), `timers` (:98-105),
    `events`, `collisions`, the four duration vectors, `pb_pixels` after draw() outside the score's text rows."""
    import spacefortress.core as sf
    z = np.load(os.path.join(GOLDEN, "getters", name + ".npz"))
    keys = np.load(os.path.join(GOLDEN, name + ".npz"))["keys"]
    meta = json.loads(str(z["meta"]))
    youturn = meta["gametype"] in ("youturn", "test-youturn")
    kw = dict(meta["kwargs"], viewport=tuple(meta["kwargs"]["viewport"]))
    g = sf.Game(meta["gametype"], seed=meta["seed"], **kw)
    assert set(meta["undefined"]) == {"vulnerability_time", "vulnerability_timer", "max_points"} and len(meta["getters"]) == 37
    vectors = ("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul")
    scalars = [k for k in meta["getters"] if k in z.files and z[k].ndim == 1 and k not in ("events", "collisions") + vectors]
    assert len(scalars) == 23, scalars
    mi, frames = 0, dict(zip(z["frame_ticks"].tolist(), z["frames"]))
    for t in range(len(keys)):
        drive(g, int(keys[t]), youturn)
        assert g.step_one_tick(34) == int(z["eng_reward"][t]), t
        for k in scalars:
            got, want = getattr(g, k), z[k][t]
            assert type(got) is (bool if z[k].dtype == np.uint8 else float if z[k].dtype == np.float64 else int), (k, type(got))
            if k in ("aim", "vdir", "ndist"):
                # Game::calculateExtra's bearings (SRC/game.cpp:304-322): the lanes' own degrees arithmetic (sf_deg_dd.h) against
                # glibc's atan2 * 180 / pi -- 1e-9 where north_star allows 1e-5; every whole-degree decision they feed is exact
                assert abs(got - want) < 1e-9, (t, k, got, want)
            else:
                assert got == want, (t, k, got, want)
        assert g.stats == tuple(int(v) for v in z["stats_i"][t]) + tuple(float(v) for v in z["stats_d"][t]), t
        assert g.timers == tuple(int(v) for v in z["timers"][t]), t
        n = int(z["n_missiles"][t])
        want = tuple(tuple(float(v) for v in row) for row in z["missiles"][mi:mi + n])
        assert g.missiles == want and g.shells == tuple(tuple(float(v) for v in row) for row in z["shells"][mi:mi + n]), t
        assert len(g.shells) == int(z["n_shells"][t])
        mi += n
        assert g.events == tuple(e for e in str(z["events"][t]).split(",") if e), (t, g.events, z["events"][t])
        assert g.collisions == tuple(e for e in str(z["collisions"][t]).split(",") if e), (t, g.collisions, z["collisions"][t])
        for k in ("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul"):
            assert len(getattr(g, k)) == int(z[k + "_len"][t]), (t, k)
        assert g.is_game_over() == bool(z["game_over"][t])
        if t in frames:
            g.draw()
            got = np.frombuffer(g.pb_pixels, np.uint8).reshape(92, 90, 4)
            assert np.array_equal(got, frames[t].reshape(92, 90, 4)), t  # (all rows: the score's text is the reference's since round 6)
    for k in ("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul"):
        assert len(z[k + "_games"]) == 1 and getattr(g, k) == tuple(int(v) for v in z[k]), k
    g.close()


@pytest.mark.parametrize("name", ["youturn_rapid_fire", "autoturn_destroy"])
def test_an_envs_own_game_follows_the_envs_steps(name):
    """`env.g` (ENV:164): rl/envs.py:10-16-style code makes the env with gym.make, wraps it, and scripts reach through to
    `env.g.points`, `.missiles`, `.events`...  The recorded run's ACTIONS go through SSF_Env.step; after every step the
    env's Game shows what the reference's extension showed for the same tick (tests/golden/getters)."""
    import spacefortress.gym as sfg
    z = np.load(os.path.join(GOLDEN, "getters", name + ".npz"))
    gz = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))

    # rl/envs.py:10-16: gym.make(env_id) -> env.seed(seed + rank) -> wrapper; the symbolic observation is an explicit kwarg
    env = sfg.make_env("SpaceFortress-%s-image-v0" % meta["gametype"].replace("-", ""), 3, 1, obs_type="features",
                       action_set=meta["action_set"])()
    assert env.np_random is not None and 0 <= env.np_random.randint(10) < 10  # ENV:159-161
    g = env.g
    assert g is env.g and g.points == 0.0 and g.tick == 0 and g.missiles == ()
    mi = 0
    for t, a in enumerate(gz["actions"]):
        obs, r, done, info = env.step(int(a))
        assert r == int(gz["reward"][t]) and done == bool(gz["done"][t]) and info == bool(gz["info"][t]), t
        assert (g.tick, g.time, g.vulnerability) == (int(z["tick"][t]), int(z["time"][t]), int(z["vulnerability"][t])), t
        assert (g.ship_x, g.ship_y, g.points, g.raw_points) == tuple(float(z[k][t]) for k in ("ship_x", "ship_y", "points", "raw_points")), t
        n = int(z["n_missiles"][t])
        assert g.missiles == tuple(tuple(float(v) for v in row) for row in z["missiles"][mi:mi + n]) and len(g.shells) == n, t
        mi += n
        assert g.stats == tuple(int(v) for v in z["stats_i"][t]) + tuple(float(v) for v in z["stats_d"][t]), t
        assert g.timers == tuple(int(v) for v in z["timers"][t]), t
        assert g.events == tuple(e for e in str(z["events"][t]).split(",") if e), (t, g.events)
        assert abs(g.aim - z["aim"][t]) < 1e-9 and abs(g.vdir - z["vdir"][t]) < 1e-9 and abs(g.ndist - z["ndist"][t]) < 1e-9
        assert len(obs) == (19 if meta["gametype"] == "youturn" else 17) and obs[11] == g.vulnerability  # ENV:134-157
    for k in ("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul"):
        assert getattr(g, k) == tuple(int(v) for v in z[k]), k
    g.draw()
    assert len(g.pb_pixels) == 92 * 90 * 4
    with pytest.raises(RuntimeError):
        g.step_one_tick(34)  # the env steps its Game
    env.reset()  # ENV:164: a new Game
    assert env.g.tick == 0 and env.g.events == () and env.g.shot_durations == ()
    env.close()


def test_game_shim_errors():
    import spacefortress.core as sf
    with pytest.raises(RuntimeError):
        sf.Game("no-such-config")  # SRC/pymodule.cpp:341
    g = sf.Game("autoturn")
    with pytest.raises(ValueError):
        g.press_key(sf.LEFT_KEY)
    with pytest.raises(ValueError):
        g.press_key(9)
    g.close()
