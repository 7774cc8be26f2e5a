"""The getter-level fixtures (tests/golden/getters: what the reference's own CPython extension `_spacefortress`, built from
SRC/pymodule.cpp, returned after every tick -- make_getters_golden.py) against the oracle's restatement of the engine
(oracle/sf_oracle.c) on the same key calls: the CPU half of tests/test_gpu_game_shim.py's comparison."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O

RUNS = ["autoturn_destroy", "youturn_deaths", "youturn_rapid_fire", "youturn_allkeys", "autoturn_allkeys",
        "autoturn_small_hex", "youturn_hunter", "testyouturn_random", "testautoturn_random"]


@pytest.mark.parametrize("name", RUNS)
def test_oracle_engine_equals_the_real_extensions_getters(name):
    z = np.load(os.path.join(GOLDEN, "getters", name + ".npz"))
    keys = np.load(os.path.join(GOLDEN, name + ".npz"))["keys"]
    meta = json.loads(str(z["meta"]))
    assert len(meta["getters"]) == 37 and sorted(meta["undefined"]) == ["max_points", "vulnerability_time", "vulnerability_timer"]
    g = O.OracleEnv(meta["gametype"], action_set=0, seed=meta["seed"], spawn_skip=meta["spawn_skip"])
    mi = 0
    for t in range(len(keys)):
        g.apply_keys(int(keys[t]), g.youturn)
        assert g.step_one_tick(34) == int(z["eng_reward"][t]), t
        s = g.snapshot()
        for k, f in (("tick", "tick"), ("time", "time"), ("vulnerability", "vlner"), ("ship_alive", "ship_alive"),
                     ("fortress_alive", "fort_alive"), ("thrust_flag", "thrust_flag"), ("turn_flag", "turn_flag"),
                     ("ship_x", "ship_x"), ("ship_y", "ship_y"), ("ship_vx", "ship_vx"), ("ship_vy", "ship_vy"),
                     ("ship_angle", "ship_angle"), ("fortress_angle", "fort_angle"), ("points", "points"),
                     ("raw_points", "raw_points")):
            assert s[f] == z[k][t], (t, k, s[f], z[k][t])  # floats to the bit
        if s["ship_alive"]:  # (mExtra is refreshed while the ship lives: SRC/game.cpp:304-322)
            for k in ("vdir", "aim", "ndist"):
                assert s[k] == z[k][t], (t, k)
        assert tuple(int(v) for v in s["stats"]) == tuple(int(v) for v in z["stats_i"][t]), t
        assert (float(s["points"]), float(s["raw_points"])) == tuple(z["stats_d"][t]), t
        assert tuple(int(s[k]) for k in ("fire_timer", "thrust_timer", "left_timer", "right_timer")) == tuple(z["timers"][t]), t
        alive = np.nonzero(s["missile_alive"])[0]
        n = int(z["n_missiles"][t])
        assert len(alive) == n == int(z["n_shells"][t]), t  # `shells` walks the missiles (SRC/pymodule.cpp:131-134)
        want = z["missiles"][mi:mi + n]
        got = np.stack([s["missile_x"][alive], s["missile_y"][alive], s["missile_angle"][alive]], 1) if n else np.zeros((0, 3))
        assert np.array_equal(got, want) and np.array_equal(z["shells"][mi:mi + n], want), t
        mi += n
        assert g.is_game_over() == bool(z["game_over"][t])


def test_recorded_pb_pixels_are_the_frames_the_model_draws():
    """`pb_pixels` after draw() (every 97th tick of a run) against oracle/render_np.py on the oracle's state of that tick, outside
    the score's text rows: B = G = R = the grey value, the fourth byte 255."""
    from oracle import render_np as R
    name = "youturn_hunter"
    z = np.load(os.path.join(GOLDEN, "getters", name + ".npz"))
    keys = np.load(os.path.join(GOLDEN, name + ".npz"))["keys"]
    meta = json.loads(str(z["meta"]))
    g = O.OracleEnv(meta["gametype"], action_set=0, seed=meta["seed"], spawn_skip=meta["spawn_skip"])
    frames = dict(zip(z["frame_ticks"].tolist(), z["frames"]))
    hx = g.hex_points()
    hb, hs = hx[:12], hx[12:]
    for t in range(len(keys)):
        g.apply_keys(int(keys[t]), g.youturn)
        g.step_one_tick(34)
        if t in frames:
            f = frames[t].reshape(92, 90, 4)
            assert (f[:, :, 3] == 255).all() and (f[:, :, 0] == f[:, :, 1]).all() and (f[:, :, 1] == f[:, :, 2]).all()
            got = R.render_raw(g.snapshot(), hb, hs)  # every row: the score text comes from the glyph atlas
            assert np.array_equal(got, f[:, :, 0]), t
