#!/usr/bin/env python3
"""Getter-level fixtures from the reference's REAL Python extension.

Run in the build container only (needs /root/reference and /opt/conda's cairo):

    python tests/golden/getters/make_getters_golden.py

`make -C oracle refpy` compiles SRC/pymodule.cpp + the engine + draw.cpp + wireframe.cpp, where they lie, into
oracle/_ref/_spacefortress<ext>.so (no stand-in header or library).  This script imports that module, replays the key
sequences of the recorded runs of tests/golden/*.npz through `_spacefortress.Game` exactly as SSF_Env.step does
(ENV:213-231), and stores what every attribute of SRC/pymodule.cpp:372-411 returned after every tick -- all 37 but the three
whose getters are undefined behaviour (UNDEFINED below): 34 attributes, `pb_pixels` on every 97th tick (after draw()).
The outputs are data: one getters/<name>.npz per run.

Two settings of the C library make the run the recorded one, none of them touches reference code:
  * every env of the reference is its own process with its own rand() stream (SRC/game.cpp:137-148; no srand anywhere):
    initstate(seed) as oracle/ref_driver.cpp does for the recorded runs, and `spawn_skip` games constructed first;
  * Game::Game leaves Fortress::mVulnerabilityTimer and mExtra uninitialised (SRC/game.cpp:78); `new Game` in a fresh
    process reads zeros there, and MALLOC_PERTURB_=255 (glibc: malloc fills with 0xff ^ 255 = 0) keeps it so in a process that
    has run Python for a while.  The script re-executes itself with it set.
While recording, every tick is also compared with the state the same run has in tests/golden/<name>.npz (recorded through
oracle/ref_driver.cpp's own reading of the members): the two ways to look at the reference agree.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
ROOT = os.path.dirname(os.path.dirname(GOLDEN))
REFDIR = os.path.join(ROOT, "oracle", "_ref")

RUNS = ["autoturn_destroy", "youturn_deaths", "youturn_rapid_fire", "youturn_allkeys", "autoturn_allkeys",
        "autoturn_small_hex", "youturn_hunter", "testyouturn_random", "testautoturn_random"]
SCALARS_I = ["tick", "time", "max_time", "bighex", "smallhex", "vulnerability", "turn_flag", "pb_width", "pb_height"]
SCALARS_B = ["ship_alive", "fortress_alive", "thrust_flag"]
SCALARS_D = ["ship_x", "ship_y", "ship_vx", "ship_vy", "ship_angle", "vdir", "aim", "ndist", "fortress_angle", "points",
             "raw_points"]
VECTORS = ["thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul"]
# never called: vulnerability_time / vulnerability_timer hand an int to Py_BuildValue("d") (SRC/pymodule.cpp:43-44), max_points
# reads the double of an int-typed config entry that was never written (:39; Config::getDouble prints "Config mismatch")
UNDEFINED = ["vulnerability_time", "vulnerability_timer", "max_points"]
FRAME_EVERY = 97  # draw() + pb_pixels on these ticks (and the last one)


def main():
    if os.environ.get("MALLOC_PERTURB_") != "255":
        env = dict(os.environ, MALLOC_PERTURB_="255")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refpy"])
    sys.path.insert(0, REFDIR)
    import _spacefortress as sf
    names = sorted(a for a in dir(sf.Game) if not a.startswith("_"))
    getters = [a for a in names if a not in ("press_key", "release_key", "step_one_tick", "is_game_over", "draw", "config", "dump")]
    assert len(getters) == 37, getters
    libc = ctypes.CDLL(None)
    libc.initstate.restype = ctypes.c_void_p
    libc.initstate.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_size_t]
    keep = []  # (the C library writes into the state array it is leaving when it is given a new one: none is ever freed here)
    for name in RUNS:
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        meta = json.loads(str(z["meta"]))
        gt, youturn = meta["gametype"], meta["gametype"] in ("youturn", "test-youturn")
        rng = ctypes.create_string_buffer(128)
        keep.append(rng)
        libc.initstate(meta["seed"], rng, 128)
        kw = dict(width=90, height=92, viewport=(130, 80, 450, 460), lw=3, grayscale=True)  # the wrapper's own (ENV:164)
        for _ in range(meta["spawn_skip"]):
            sf.Game(gt, **kw)
        g = sf.Game(gt, **kw)
        T = len(z["keys"])
        rec = {k: np.zeros(T, np.int64) for k in SCALARS_I}
        rec.update({k: np.zeros(T, np.uint8) for k in SCALARS_B})
        rec.update({k: np.zeros(T, np.float64) for k in SCALARS_D})
        rec["eng_reward"] = np.zeros(T, np.int32)
        rec["game_over"] = np.zeros(T, np.uint8)
        rec["stats_i"] = np.zeros((T, 13), np.int64)
        rec["stats_d"] = np.zeros((T, 2), np.float64)
        rec["timers"] = np.zeros((T, 4), np.int64)
        for k in VECTORS:
            rec[k + "_len"] = np.zeros(T, np.int64)
        n_m, n_s, mis, she, events, collisions, frames, frame_ticks, vec_final = [], [], [], [], [], [], [], [], {}
        snap_every = meta["snap_every"]
        for t in range(T):
            keys = int(z["keys"][t])
            (g.press_key if keys & 1 else g.release_key)(sf.FIRE_KEY)  # ENV:213-229
            (g.press_key if keys & 2 else g.release_key)(sf.THRUST_KEY)
            if youturn:
                (g.press_key if keys & 4 else g.release_key)(sf.LEFT_KEY)
                (g.press_key if keys & 8 else g.release_key)(sf.RIGHT_KEY)
            rec["eng_reward"][t] = g.step_one_tick(34)
            for k in SCALARS_I + SCALARS_B + SCALARS_D:
                rec[k][t] = getattr(g, k)
            st = g.stats
            assert len(st) == 15
            rec["stats_i"][t], rec["stats_d"][t] = st[:13], st[13:]
            rec["timers"][t] = g.timers
            m, s = g.missiles, g.shells
            n_m.append(len(m)); n_s.append(len(s))
            mis.extend(m); she.extend(s)
            events.append(",".join(g.events))
            collisions.append(",".join(g.collisions))
            for k in VECTORS:
                rec[k + "_len"][t] = len(getattr(g, k))
            if t % FRAME_EVERY == 0 or t == T - 1:
                g.draw()
                frames.append(np.frombuffer(g.pb_pixels, np.uint8).copy())
                frame_ticks.append(t)
            over = g.is_game_over()
            rec["game_over"][t] = over
            # ---- the same tick as oracle/ref_driver.cpp recorded it
            assert rec["eng_reward"][t] == z["eng_reward"][t] and over == bool(z["done"][t]), (name, t)
            assert g.ship_x == z["scal_ship_x"][t] and g.ship_y == z["scal_ship_y"][t] and g.points == z["scal_points"][t], (name, t)
            assert len(m) == z["scal_n_missiles"][t], (name, t)
            if (t + 1) % snap_every == 0:
                sn = z["snaps"][(t + 1) // snap_every - 1]
                assert tuple(rec["timers"][t]) == tuple(int(sn[k]) for k in ("fire_timer", "thrust_timer", "left_timer", "right_timer")), (name, t)
                assert tuple(rec["stats_i"][t]) == tuple(int(v) for v in sn["stats"]), (name, t)
                assert g.fortress_angle == float(sn["fort_angle"]) and g.vulnerability == int(sn["vlner"]), (name, t)
            if over:  # the vec-env worker resets: a new Game (ENV:164), its vectors start empty
                for k in VECTORS:
                    vec_final.setdefault(k, []).append(np.array(getattr(g, k), np.int64))
                g = sf.Game(gt, **kw)
        for k in VECTORS:
            vec_final.setdefault(k, []).append(np.array(getattr(g, k), np.int64))
        out = dict(rec)
        out["meta"] = np.array(json.dumps(dict(meta, getters=getters, undefined=UNDEFINED, frame_every=FRAME_EVERY,
                                               kwargs={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()})))
        out["n_missiles"], out["n_shells"] = np.array(n_m, np.int64), np.array(n_s, np.int64)
        out["missiles"] = np.array(mis, np.float64).reshape(-1, 3)
        out["shells"] = np.array(she, np.float64).reshape(-1, 3)
        out["events"], out["collisions"] = np.array(events), np.array(collisions)
        out["frames"] = np.array(frames, np.uint8).reshape(len(frames), -1)
        out["frame_ticks"] = np.array(frame_ticks, np.int64)
        for k in VECTORS:  # per game of the run (a run that ends a game starts another): the vector as it was at the game's end
            out[k + "_games"] = np.array([len(v) for v in vec_final[k]], np.int64)
            out[k] = np.concatenate(vec_final[k]) if vec_final[k] else np.zeros(0, np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("%-24s %5d ticks, %4d missiles, %4d shells, %3d frames, events on %d ticks" %
              (name, T, len(mis), len(she), len(frames), sum(1 for e in events if e)))


if __name__ == "__main__":
    main()
