#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference engine.

Run in the build container only (needs /root/reference; builds oracle/_ref via
oracle/Makefile).  The outputs are data -- action sequences and the states /
rewards the reference produced for them -- never reference source.

    python tests/golden/make_golden.py

Each scenario file ``<name>.npz`` holds
    meta        json: gametype, action_set, seed, spawn_skip, snap_every, description
    actions     u8[T]    index into the wrapper's action table (ENV:64-89)
    keys        u8[T]    the key bits sent (bit0 FIRE, 1 THRUST, 2 LEFT, 3 RIGHT)
    eng_reward  i32[T]   Game::stepOneTick's return (SRC/game.cpp:484)
    reward      i32[T]   after SSF_Env.step's shaping (ENV:233-244), applied here
                         exactly as written there on the engine's reward/vulnerability
    done, info  u8[T]    ENV:246, ENV:233
    snaps       sfo_snapshot[ceil(T/snap_every)]  full engine state AFTER step t,
                         t = snap_every-1, 2*snap_every-1, ... (before any new Game)
    scal_*      per-step scalars (ship_x, ship_y, ship_alive, vlner, points, raw_points,
                n_missiles, n_shells) for every step
    reset_snaps sfo_snapshot[n_resets]  state right after each new Game (ENV:164)

The scripted scenarios are closed-loop: the script looks at the reference's own
state to choose the next action, and the chosen actions are what is recorded.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

ACTION_KEYS = {
    ("youturn", 1): [0, 1, 2, 4, 8],
    ("autoturn", 1): [0, 1, 2],
}


def action_table(gametype, action_set):
    """Key bits per action index, from numpy itself for the meshgrid sets (ENV:67-89)."""
    youturn = gametype in ("youturn", "test-youturn")
    if action_set == 1:
        rows = [[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]] if youturn else [[0, 0], [1, 0], [0, 1]]
        rows = np.array(rows)
    elif youturn or action_set == -1:
        rows = np.array(np.meshgrid([0, 1], [0, 1], [0, 1], [0, 1])).T.reshape(-1, 4)
    else:
        rows = np.array(np.meshgrid([0, 1], [0, 1])).T.reshape(-1, 2)
    ncol = 4 if youturn else 2  # ENV:213-229 reads columns 0,1 (+2,3 in youturn)
    return [int(sum(int(r[c]) << c for c in range(ncol))) for r in rows]


class Recorder:
    def __init__(self, name, gametype, action_set=1, seed=1, spawn_skip=0, snap_every=1, description=""):
        self.name = name
        self.gametype = gametype
        self.table = action_table(gametype, action_set)
        self.meta = dict(gametype=gametype, action_set=action_set, seed=seed, spawn_skip=spawn_skip,
                         snap_every=snap_every, description=description, tickdur=34)
        self.g = O.RefGame(gametype, seed=seed, spawn_skip=spawn_skip)
        self.shaped = gametype in ("autoturn", "youturn")  # ENV:235
        self.prev_vlner = 0  # ENV:92
        self.snap_every = snap_every
        self.actions, self.keys, self.eng, self.rew, self.done, self.info = [], [], [], [], [], []
        self.snaps, self.reset_snaps = [], [self.g.snapshot()]
        self.scal = {k: [] for k in ("ship_x", "ship_y", "ship_alive", "vlner", "points", "raw_points",
                                     "n_missiles", "n_shells")}
        self.state = self.g.snapshot()

    def step(self, action):
        g = self.g
        keys = self.table[action]
        g.apply_keys(keys, g.youturn)
        r = g.step_one_tick(34)
        s = g.snapshot()
        eng = r
        fort_kill = r > 0
        if self.shaped:
            vl = int(s["vlner"])
            if vl <= 10 and not fort_kill:
                r += vl - self.prev_vlner
            r = max(-1, min(1, r))
            r = r + 2 * int(fort_kill)
            self.prev_vlner = vl
        done = g.is_game_over()
        t = len(self.actions)
        self.actions.append(action)
        self.keys.append(keys)
        self.eng.append(eng)
        self.rew.append(r)
        self.done.append(done)
        self.info.append(fort_kill)
        if (t + 1) % self.snap_every == 0:
            self.snaps.append(s)
        for k in ("ship_x", "ship_y", "ship_alive", "vlner", "points", "raw_points"):
            self.scal[k].append(s[k])
        self.scal["n_missiles"].append(int(s["missile_alive"].sum()))
        self.scal["n_shells"].append(int(s["shell_alive"].sum()))
        if done:  # the vec-env worker resets (a15)
            g.new_game()
            self.reset_snaps.append(g.snapshot())
            s = g.snapshot()
        self.state = s
        return s, r, done, fort_kill

    def save(self):
        out = dict(
            meta=np.array(json.dumps(self.meta)),
            actions=np.array(self.actions, np.uint8),
            keys=np.array(self.keys, np.uint8),
            eng_reward=np.array(self.eng, np.int32),
            reward=np.array(self.rew, np.int32),
            done=np.array(self.done, np.uint8),
            info=np.array(self.info, np.uint8),
            snaps=np.array(self.snaps, O.SNAPSHOT_DTYPE),
            reset_snaps=np.array(self.reset_snaps, O.SNAPSHOT_DTYPE),
        )
        for k, v in self.scal.items():
            dt = {"ship_alive": np.uint8, "vlner": np.int32, "n_missiles": np.uint8, "n_shells": np.uint8,
                  "points": np.float32, "raw_points": np.float32}.get(k, np.float64)
            out["scal_" + k] = np.array(v, dt)
        path = os.path.join(HERE, self.name + ".npz")
        np.savez_compressed(path, **out)
        st = self.state
        print("%-28s T=%5d  sum_r=%5d kills=%d deaths=%s  %6.0f KB" % (
            self.name, len(self.actions), sum(self.rew), sum(self.info),
            list(self.snaps[-1]["stats"][:4]) if self.snaps else "-", os.path.getsize(path) / 1024))
        return path


# ---------------------------------------------------------------- scenarios


def random_rollout(name, gametype, T, rng_seed, action_set=1, snap_every=1, seed=1, spawn_skip=0, desc=""):
    rec = Recorder(name, gametype, action_set, seed=seed, spawn_skip=spawn_skip, snap_every=snap_every,
                   description=desc or "uniform random actions, numpy default_rng(%d)" % rng_seed)
    rng = np.random.default_rng(rng_seed)
    acts = rng.integers(0, len(rec.table), T)
    for a in acts:
        rec.step(int(a))
    return rec.save()


def autoturn_destroy(name, concurrent_fire):
    """autoturn: the ship always faces the fortress.  Fire one missile every 10
    ticks until vulnerability reaches 11 (ten +1 rewards, the 11th gives 0), then
    two missiles two ticks apart: the second hit lands < 250 ms after the first
    and destroys the fortress (reward 3, info True).  With concurrent_fire the
    script keeps pressing FIRE every other tick through the kill, so one press
    lands on the kill tick: -0.05 + 1 truncates to 0 (SRC/game.cpp:484) and the
    wrapper never sees the kill."""
    rec = Recorder(name, "autoturn", description=autoturn_destroy.__doc__)
    NOOP, FIRE, THRUST = 0, 1, 2
    t = 0
    kills_wanted = 2
    while sum(rec.info) + (1 if concurrent_fire else 0) * 0 < kills_wanted and t < 4000:
        s = rec.state
        if not s["ship_alive"] or not s["fort_alive"]:
            rec.step(NOOP)
        elif s["vlner"] < 11:
            rec.step(FIRE if t % 10 == 0 else NOOP)
        else:
            # burst: press every other tick until the fortress dies
            for k in range(40):
                a = FIRE if k % 2 == 0 else NOOP
                if not concurrent_fire and k >= 4:
                    a = NOOP
                s2, r, d, info = rec.step(a)
                t += 1
                if not s2["fort_alive"] or not s2["ship_alive"]:
                    break
            if concurrent_fire and rec.snaps[-1]["stats"][5] >= kills_wanted:
                break
            continue
        t += 1
    for _ in range(40):  # fortress respawn (> 1000 ms) and a few hits on a dead fortress
        rec.step(FIRE if _ % 2 == 0 else NOOP)
    return rec.save()


def youturn_hunter(name, T):
    """youturn: closed-loop aiming.  Turn until the reference's own `aim` is
    within 4 degrees, thrust gently to stay between the hexagons, fire every 9
    ticks while aimed; burst when vulnerability >= 11."""
    rec = Recorder(name, "youturn", description=youturn_hunter.__doc__)
    NOOP, FIRE, THRUST, LEFT, RIGHT = range(5)
    last_fire = -100
    for t in range(T):
        s = rec.state
        if not s["ship_alive"]:
            rec.step(NOOP)
            continue
        aim = float(s["aim"])
        if aim > 180:
            aim -= 360
        period = 2 if s["vlner"] >= 11 else 9
        if abs(aim) <= 4 and t - last_fire >= period and not s["fire_flag"]:
            rec.step(FIRE)
            last_fire = t
        elif aim > 4:  # heading must grow: RIGHT adds 6 degrees (SRC/game.cpp:323-324)
            rec.step(RIGHT)
        elif aim < -4:
            rec.step(LEFT)
        else:
            rec.step(NOOP)
    return rec.save()


def youturn_scripted_deaths(name):
    """youturn: (1) hold THRUST from spawn until the ship leaves the big hexagon;
    (2) after respawn do nothing until something kills the ship; (3) spray FIRE
    every other tick while dead and alive (shots while dead cost nothing but
    count); repeat twice."""
    rec = Recorder(name, "youturn", description=youturn_scripted_deaths.__doc__)
    NOOP, FIRE, THRUST, LEFT, RIGHT = range(5)
    for rep in range(2):
        while rec.state["ship_alive"]:
            rec.step(THRUST)
        for k in range(40):
            rec.step(FIRE if k % 2 == 0 else NOOP)
        n = 0
        while rec.state["ship_alive"] and n < 900:
            rec.step(NOOP)
            n += 1
        for k in range(35):
            rec.step([LEFT, FIRE, RIGHT, NOOP][k % 4])
    return rec.save()


def autoturn_small_hex(name):
    """autoturn: hold THRUST: the ship accelerates toward the fortress and dies
    on the small hexagon; NOOP through two respawns (shell or hexagon deaths)."""
    rec = Recorder(name, "autoturn", description=autoturn_small_hex.__doc__)
    NOOP, FIRE, THRUST = 0, 1, 2
    for rep in range(3):
        n = 0
        while rec.state["ship_alive"] and n < 600:
            rec.step(THRUST)
            n += 1
        for k in range(32):
            rec.step(NOOP)
    for k in range(700):
        rec.step(NOOP)
    return rec.save()


def youturn_rapid_fire(name):
    """youturn: FIRE pressed every other tick for 400 ticks with slow turning:
    the largest live-missile counts reachable in play, misses leaving the area."""
    rec = Recorder(name, "youturn", description=youturn_rapid_fire.__doc__)
    NOOP, FIRE, THRUST, LEFT, RIGHT = range(5)
    for t in range(400):
        rec.step(FIRE if t % 2 == 0 else (LEFT if t % 8 == 1 else NOOP))
    return rec.save()


def tables():
    """Constant tables observed from the reference: missile velocity at every
    integer heading, thrust increments, the spawn sequence, hexagon vertices."""
    out = {}
    g = O.RefGame("youturn")
    out["hex_points"] = g.hex_points()
    # spawn sequence of the seed-1 stream and one other seed: (x, y, angle)
    for seed in (1, 12345):
        g = O.RefGame("youturn", seed=seed)
        sp = []
        for i in range(4096):
            s = g.snapshot()
            sp.append((int(s["ship_x"]), int(s["ship_y"]), int(s["ship_angle"])))
            g.new_game()
        out["spawns_seed%d" % seed] = np.array(sp, np.int16)
    s0 = O.RefGame("youturn").snapshot()
    out["start_vel"] = np.array([s0["ship_vx"], s0["ship_vy"]])
    # missile velocity and thrust increment per integer heading: turn a fresh ship
    # (youturn, +-6 deg per tick) -- headings reached depend on the spawn angle, so
    # walk spawns until all 360 headings were seen.
    mv = np.full((360, 2), np.nan)
    th = np.full((360, 2), np.nan)
    g = O.RefGame("youturn")
    NO, FIRE, THRUST, LEFT = 0, 1, 2, 4
    guard = 0
    while (np.isnan(mv).any() or np.isnan(th).any()) and guard < 4000:
        guard += 1
        s = g.snapshot()
        a = int(s["ship_angle"])
        if np.isnan(mv[a, 0]):
            # fire: the missile is created before the ship moves (SRC/game.cpp:237-238)
            g.apply_keys(FIRE, True)
            g.step_one_tick(34)
            s2 = g.snapshot()
            slot = int(np.flatnonzero(s2["missile_alive"])[-1]) if s2["missile_alive"].any() else None
            if slot is not None and int(s2["missile_angle"][slot]) == a:
                mv[a] = (s2["missile_vx"][slot], s2["missile_vy"][slot])
            g.new_game()
            continue
        if np.isnan(th[a, 0]):
            g.apply_keys(THRUST, True)
            g.step_one_tick(34)
            s2 = g.snapshot()
            th[a] = (s2["ship_vx"] - s["ship_vx"], s2["ship_vy"] - s["ship_vy"])  # not exact: informational
            th[a] = (s2["ship_vx"], s2["ship_vy"])  # exact: start_vel + 0.3*(cos,sin)
            g.new_game()
            continue
        g.new_game()
    assert not np.isnan(mv).any() and not np.isnan(th).any(), "not all headings reached"
    out["missile_vel_by_angle"] = mv
    out["thrust_vel_by_angle"] = th  # ship velocity after one thrust tick from the start velocity
    path = os.path.join(HERE, "tables.npz")
    np.savez_compressed(path, **out)
    print("tables.npz %.0f KB" % (os.path.getsize(path) / 1024))


def main():
    O.build()
    assert O.have_ref(), "oracle/_ref/libsfref.so missing: needs /root/reference"
    tables()
    random_rollout("youturn_random_ep", "youturn", 5295 + 705, 101, snap_every=8,
                   desc="full episode + rollover into a second one; uniform random actions default_rng(101)")
    random_rollout("autoturn_random_ep", "autoturn", 5295 + 705, 102, snap_every=8,
                   desc="full episode + rollover; uniform random actions default_rng(102)")
    random_rollout("youturn_random_short", "youturn", 1500, 103)
    random_rollout("autoturn_random_short", "autoturn", 1500, 104)
    random_rollout("youturn_allkeys", "youturn", 1500, 105, action_set=0)
    random_rollout("autoturn_allkeys", "autoturn", 1500, 106, action_set=0)
    random_rollout("testyouturn_random", "test-youturn", 1200, 107)
    random_rollout("testautoturn_random", "test-autoturn", 1200, 108)
    random_rollout("youturn_seed12345_skip3", "youturn", 1200, 109, seed=12345, spawn_skip=3)
    autoturn_destroy("autoturn_destroy", concurrent_fire=False)
    autoturn_destroy("autoturn_destroy_truncated", concurrent_fire=True)
    youturn_hunter("youturn_hunter", 2500)
    youturn_scripted_deaths("youturn_deaths")
    autoturn_small_hex("autoturn_small_hex")
    youturn_rapid_fire("youturn_rapid_fire")


if __name__ == "__main__":
    main()
