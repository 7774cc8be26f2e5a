"""Parity of the HIP path (libsfmi.so through SFVecEnv / the C ABI) against
  * the golden vectors recorded from the real reference engine, and
  * the CPU oracle on seeded random batches,
plus size-independent properties at BASELINE.json's full batch sizes.

Bar: integers, flags, float32 scores, ship and missile positions BIT-EXACT; shell
positions (their velocity comes from a device sin/cos of a non-integer heading)
within 1e-9; observations within 1e-5 relative to the float32 the wrapper emits
(float64 observations: 1e-9)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, golden_names
from sfcompare import compare_state, snapshots_to_fields

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z, json.loads(str(z["meta"]))


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    from spacefortress_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsfmi.so not built: the GPU tests never fall back"
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


def run_device(env, acts_tn, state_every=None, state_cb=None):
    """Step `env` through acts_tn [T, N] (uint8) on the device; returns numpy obs/reward/done/info."""
    T, N = acts_tn.shape
    dev = env.device
    a = torch.from_numpy(np.ascontiguousarray(acts_tn)).to(dev)
    obs = torch.empty((T, N, env.obs_dim), dtype=env.obs_dtype, device=dev)
    rew = torch.empty((T, N), dtype=torch.int32, device=dev)
    done = torch.empty((T, N), dtype=torch.uint8, device=dev)
    info = torch.empty((T, N), dtype=torch.uint8, device=dev)
    for t in range(T):
        env.step_tensors(a[t], out=(obs[t], rew[t], done[t], info[t]))
        if state_every and (t + 1) % state_every == 0:
            state_cb(t, env.state_dict())
    torch.cuda.synchronize()
    env.check_actions()
    return obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool), info.cpu().numpy().astype(bool)


def obs_close(a, b, f64):
    tol = 1e-9 if f64 else 1e-5
    scale = np.maximum(1.0, np.abs(b))
    return np.abs(a.astype(np.float64) - b) <= tol * scale + (0 if f64 else 4e-5)


# ------------------------------------------------------------------ golden vectors

@pytest.mark.parametrize("name", golden_names())
def test_hip_replays_golden(sfa, name):
    """The recorded reference run, replayed on the GPU: rewards/done/info every step,
    full state at every recorded snapshot."""
    z, meta = load(name)
    N = 3  # identical lanes: also checks lanes do not interfere
    env = sfa.SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"],
                       spawn_skip=meta["spawn_skip"], obs_dtype=torch.float64)
    acts = np.repeat(z["actions"][:, None], N, 1)
    every = meta["snap_every"]
    done_steps = set(np.flatnonzero(z["done"]).tolist())
    problems = []

    def cb(t, sd):
        if t in done_steps:
            return  # the lane was auto-reset; the golden snapshot is the pre-reset state
        snaps = np.repeat(z["snaps"][(t + 1) // every - 1][None], N)
        bad = compare_state(sd, snaps)
        if bad and len(problems) < 5:
            problems.append((t, bad))

    # comparing the state each `every` steps costs a sync; thin it out on the long runs
    stride = every if len(acts) <= 700 else every * max(1, 40 // every)
    obs, rew, done, info = run_device(env, acts, state_every=stride, state_cb=cb)
    assert not problems, problems
    for lane in range(N):
        assert np.array_equal(rew[:, lane], z["reward"]), np.flatnonzero(rew[:, lane] != z["reward"])[:5]
        assert np.array_equal(done[:, lane], z["done"].astype(bool))
        assert np.array_equal(info[:, lane], z["info"].astype(bool))
    # observation columns that the golden scalars pin (features obs): x, y, vlner
    nd = ~z["done"].astype(bool)
    assert np.array_equal(obs[nd, 0, 1], z["scal_ship_x"][nd]) and np.array_equal(obs[nd, 0, 2], z["scal_ship_y"][nd])
    assert np.array_equal(obs[nd, 0, 11], z["scal_vlner"][nd])
    assert np.array_equal(obs[nd, 0, 13], z["scal_n_missiles"][nd])
    env.close()


# ------------------------------------------------------------------ oracle, random batches

@pytest.mark.parametrize("gametype,action_set,obs_type,f64", [
    ("youturn", 1, "features", True),
    ("autoturn", 1, "features", True),
    ("youturn", 0, "features", False),
    ("autoturn", 0, "normalized-features", True),
    ("test-youturn", 1, "monitors", False),
    ("test-autoturn", 1, "normalized-features", False),
    ("youturn", 1, "normalized-features", True),
    ("autoturn", 1, "monitors", True),
])
def test_hip_vs_oracle_random(sfa, oracle_mod, gametype, action_set, obs_type, f64):
    """2048 lanes, different random actions and spawn offsets per lane, 1500 steps in lock-step
    with the CPU oracle: every output of every step, and the full state at checkpoints."""
    lockstep(sfa, oracle_mod, gametype, action_set, obs_type, f64, hunter=False)


@pytest.mark.parametrize("gametype,action_set,obs_type,f64", [
    ("autoturn", 1, "features", False),
    ("autoturn", 0, "normalized-features", True),
    ("youturn", 1, "features", True),
])
def test_hip_vs_oracle_hunter(sfa, oracle_mod, gametype, action_set, obs_type, f64):
    """Random play almost never destroys the fortress; this firing pattern (a shot per 272 ms until kill-ready,
    then a double shot; random phase per lane, 10 % random actions) does so hundreds of times in the batch:
    vulnerability resets, kills with info=True, fortress respawns and the reward shaping around them."""
    kills = lockstep(sfa, oracle_mod, gametype, action_set, obs_type, f64, hunter=True)
    assert kills > (400 if gametype == "autoturn" else 20), kills


def lockstep(sfa, oracle_mod, gametype, action_set, obs_type, f64, hunter):
    O = oracle_mod
    N, T = 2048, 1500
    rng = np.random.default_rng(sum(map(ord, gametype)) + action_set)
    env = sfa.SFVecEnv(N, gametype=gametype, action_set=action_set, obs_type=obs_type, spawn_stride=3,
                       spawn_skip=1, obs_dtype=torch.float64 if f64 else torch.float32)
    orc = O.OracleVecEnv(gametype, N, action_set=action_set, obs_type=obs_type, spawn_stride=3, spawn_skip=1)
    acts = rng.integers(0, env.n_actions, (T, N)).astype(np.uint8)
    if hunter:  # FIRE is action 1 in every action set (ENV:211-229)
        pat = np.array(([1] + [0] * 7) * 11 + [1, 0, 1, 0] + [0] * 4, np.uint8)
        tt = (np.arange(T)[:, None] + rng.integers(0, len(pat), N)[None, :]) % len(pat)
        acts = np.where(rng.random((T, N)) < 0.1, acts, pat[tt]).astype(np.uint8)
    o0 = env.reset().cpu().numpy()
    oo0 = orc.reset()
    assert obs_close(o0, oo0, f64).all()
    checkpoints = {}

    def cb(t, sd):
        checkpoints[t] = sd

    obs, rew, done, info = run_device(env, acts, state_every=250, state_cb=cb)
    for t in range(T):
        oo, orw, od, oi = orc.step(acts[t].astype(np.int32))
        assert np.array_equal(rew[t], orw), (t, np.flatnonzero(rew[t] != orw)[:5])
        assert np.array_equal(done[t], od) and np.array_equal(info[t], oi), t
        ok = obs_close(obs[t], oo, f64)
        assert ok.all(), (t, np.argwhere(~ok)[:5], obs[t][~ok][:5], oo[~ok][:5])
        if t in checkpoints:
            bad = compare_state(checkpoints[t], orc.snapshots())
            assert not bad, (t, bad)
            assert np.array_equal(checkpoints[t]["prev_vlner"], orc.prev_vlner())
    env.close()
    return int(info.sum())


def test_episode_rollover_and_stats(sfa, oracle_mod):
    """Every lane finishes its episode at step 5295 and is auto-reset in the kernel; the
    device-side episode accumulators equal what the trainer would compute (rl/train.py:81-88)."""
    O = oracle_mod
    N, T = 64, 5295 + 40
    rng = np.random.default_rng(7)
    env = sfa.SFVecEnv(N, gametype="youturn", spawn_stride=1)
    orc = O.OracleVecEnv("youturn", N, spawn_stride=1)
    acts = rng.integers(0, 5, (T, N)).astype(np.uint8)
    obs, rew, done, info = run_device(env, acts)
    ret = np.zeros(N, np.int64)
    fin = []
    for t in range(T):
        oo, orw, od, oi = orc.step(acts[t].astype(np.int32))
        assert np.array_equal(rew[t], orw) and np.array_equal(done[t], od) and np.array_equal(info[t], oi), t
        assert obs_close(obs[t], oo, False).all(), t
        ret += orw
        for i in np.flatnonzero(od):
            fin.append(int(ret[i]))
            ret[i] = 0
    assert done[5294].all() and done.sum() == N
    st = env.episode_stats()
    fin = np.array(fin)
    assert st[0] == N and st[1] == fin.sum() and st[2] == (fin ** 2).sum()
    assert st[3] == info[:5295].sum() and st[6] == fin.min() and st[7] == fin.max()
    assert not compare_state(env.state_dict(), orc.snapshots())
    env.close()


# ------------------------------------------------------------------ constructed edge cases

def _load_both(sfa, O, gametype, snaps, prev_vlner=None, **kw):
    n = len(snaps)
    env = sfa.SFVecEnv(n, gametype=gametype, obs_dtype=torch.float64, **kw)
    orc = O.OracleVecEnv(gametype, n, **{k: v for k, v in kw.items()
                                         if k in ("action_set", "obs_type", "spawn_stride", "spawn_skip")})
    orc.load_snapshots(snaps, prev_vlner)
    for k, v in snapshots_to_fields(snaps).items():
        env.set_field(k, v)
    if prev_vlner is not None:
        env.set_field("prev_vlner", np.asarray(prev_vlner, np.int32))
    return env, orc


def test_missile_slots_exhausted(sfa, oracle_mod):
    """All 20 missile slots live (unreachable in play): FIRE counts a shot, creates nothing, costs nothing
    (SRC/game.cpp:176-191,239-245).  Also 19 live -> the free slot in the middle is taken."""
    O = oracle_mod
    base = O.OracleVecEnv("youturn", 4).snapshots()
    for i in range(4):
        n_live = (20, 19, 20, 7)[i]
        for s in range(20):
            live = s < n_live if i != 1 else s != 11
            base["missile_alive"][i, s] = int(live)
            base["missile_x"][i, s] = 300 + 3 * s
            base["missile_y"][i, s] = 200 + 2 * s
            base["missile_angle"][i, s] = (17 * s + 5) % 360
    # the oracle moves missiles by their stored velocity: take it from the reference's own table
    tab = np.load(os.path.join(GOLDEN, "tables.npz"))["missile_vel_by_angle"]
    ang = base["missile_angle"].astype(int)
    base["missile_vx"] = tab[ang, 0]
    base["missile_vy"] = tab[ang, 1]
    env, orc = _load_both(sfa, O, "youturn", base)
    acts = np.array([[1, 1, 0, 1], [0, 0, 1, 0], [1, 1, 1, 1], [0, 0, 0, 0], [1, 0, 1, 1]], np.uint8)
    obs, rew, done, info = run_device(env, acts)
    for t in range(len(acts)):
        oo, orw, od, oi = orc.step(acts[t].astype(np.int32))
        assert np.array_equal(rew[t], orw), t
        assert obs_close(obs[t], oo, True).all(), t
    assert not compare_state(env.state_dict(), orc.snapshots())
    env.close()


def test_many_shells_and_kill_order(sfa, oracle_mod):
    """More live shells than the prefetched groups hold (constructed; slots up to 11), several of
    them overlapping the ship in the same tick: only the lowest colliding slot kills the ship
    (SRC/game.cpp:410-420), the others fly on; shells on the area border leave."""
    O = oracle_mod
    n = 6
    base = O.OracleVecEnv("youturn", n).snapshots()
    rng = np.random.default_rng(3)
    for i in range(n):
        sx, sy = float(base["ship_x"][i]), float(base["ship_y"][i])
        vx, vy = float(base["ship_vx"][i]), float(base["ship_vy"][i])
        live = {0: [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], 1: [2, 5, 9], 2: [0, 4, 7, 13, 19], 3: list(range(20)),
                4: [1], 5: [3, 6, 8]}[i]
        for s in live:
            base["shell_alive"][i, s] = 1
            base["shell_vx"][i, s] = rng.uniform(-6, 6)
            base["shell_vy"][i, s] = rng.uniform(-6, 6)
            base["shell_x"][i, s] = rng.uniform(5, 705)
            base["shell_y"][i, s] = rng.uniform(5, 620)
        # shells that will sit on the ship after both have moved this tick
        for s in {0: [4, 7, 10], 1: [5, 9], 2: [13, 19], 3: [6, 2, 15], 4: [], 5: [8]}[i]:
            base["shell_x"][i, s] = sx + vx - base["shell_vx"][i, s] + rng.uniform(-3, 3)
            base["shell_y"][i, s] = sy + vy - base["shell_vy"][i, s] + rng.uniform(-3, 3)
        # shells about to leave the area
        for s in {0: [1], 1: [2], 2: [0], 3: [19, 0], 4: [1], 5: [3]}[i]:
            base["shell_x"][i, s] = 709.5
            base["shell_vx"][i, s] = 5.0
    env, orc = _load_both(sfa, O, "youturn", base)
    acts = np.zeros((3, n), np.uint8)
    obs, rew, done, info = run_device(env, acts)
    for t in range(len(acts)):
        oo, orw, od, oi = orc.step(acts[t].astype(np.int32))
        assert np.array_equal(rew[t], orw), t
        assert obs_close(obs[t], oo, True).all(), t
    sn = orc.snapshots()
    assert sn["stats"][:, 2].sum() >= 4  # shell deaths really happened
    assert not compare_state(env.state_dict(), sn)
    env.close()


def _fuzz_base(O, gametype, n, rng):
    """Random CONSTRUCTED states (not reachable by play): ships anywhere in the area, arbitrary velocities and timers,
    up to 20 missiles and 20 shells per lane, fortress dead or alive, any vulnerability.  Returns (snapshots, prev_vlner)."""
    base = O.OracleVecEnv(gametype, n).snapshots()
    tab = np.load(os.path.join(GOLDEN, "tables.npz"))["missile_vel_by_angle"]
    base["ship_alive"] = rng.integers(0, 2, n)
    base["ship_x"] = np.where(rng.random(n) < 0.5, rng.integers(150, 560, n), rng.uniform(150, 560, n))
    base["ship_y"] = np.where(rng.random(n) < 0.5, rng.integers(135, 495, n), rng.uniform(135, 495, n))
    base["ship_vx"] = rng.uniform(-4, 4, n) * (rng.random(n) < 0.9)
    base["ship_vy"] = rng.uniform(-4, 4, n) * (rng.random(n) < 0.9)
    base["ship_angle"] = rng.integers(0, 360, n)
    base["ship_death_timer"] = rng.integers(0, 1200, n)
    for k in ("fire_timer", "thrust_timer", "left_timer", "right_timer"):
        base[k] = rng.integers(-50, 50, n)
    for k in ("fire_flag", "thrust_flag", "left_flag", "right_flag"):
        base[k] = rng.integers(0, 2, n)
    if gametype != "youturn" and gametype != "test-youturn":
        base["left_flag"] = 0
        base["right_flag"] = 0
    base["fort_alive"] = rng.random(n) < 0.8
    base["fort_angle"] = rng.integers(0, 36, n) * 10
    base["fort_last_angle"] = rng.integers(0, 36, n) * 10
    base["fort_timer"] = rng.integers(0, 1100, n)
    base["fort_death_timer"] = rng.integers(0, 1100, n)
    base["fort_vuln_timer"] = rng.integers(0, 400, n)
    base["vlner"] = rng.integers(0, 14, n)
    base["points"] = rng.integers(0, 5, n).astype(np.float32) * np.float32(0.05)
    base["raw_points"] = base["points"] - np.float32(1.0)
    base["time"] = rng.integers(0, 5000, n) * 34
    base["tick"] = base["time"] // 34
    base["stats"] = rng.integers(0, 50, (n, 13))
    base["stats"][:, 3] = base["stats"][:, :3].sum(1)  # shipDeaths = bigHex + smallHex + shell deaths (killShip's three call sites); it is not stored
    nm = rng.integers(0, 21, n)
    ns = rng.integers(0, 21, n)
    for i in range(n):
        ms = rng.choice(20, nm[i], replace=False)
        base["missile_alive"][i, ms] = 1
        ss = rng.choice(20, ns[i], replace=False)
        base["shell_alive"][i, ss] = 1
    base["missile_x"] = rng.uniform(-10, 720, (n, 20))
    base["missile_y"] = rng.uniform(-10, 636, (n, 20))
    near = rng.random((n, 20)) < 0.25  # a quarter of the missiles about to hit the fortress
    ang = rng.integers(0, 360, (n, 20))
    base["missile_angle"] = ang
    base["missile_vx"] = tab[ang, 0]
    base["missile_vy"] = tab[ang, 1]
    base["missile_x"] = np.where(near, 355 - tab[ang, 0] + rng.uniform(-25, 25, (n, 20)), base["missile_x"])
    base["missile_y"] = np.where(near, 315 - tab[ang, 1] + rng.uniform(-25, 25, (n, 20)), base["missile_y"])
    base["shell_x"] = rng.uniform(-5, 715, (n, 20))
    base["shell_y"] = rng.uniform(-5, 631, (n, 20))
    base["shell_vx"] = rng.uniform(-6, 6, (n, 20))
    base["shell_vy"] = rng.uniform(-6, 6, (n, 20))
    hit = rng.random((n, 20)) < 0.15  # some shells about to hit the ship
    base["shell_x"] = np.where(hit, (base["ship_x"] + base["ship_vx"])[:, None] - base["shell_vx"] + rng.uniform(-14, 14, (n, 20)), base["shell_x"])
    base["shell_y"] = np.where(hit, (base["ship_y"] + base["ship_vy"])[:, None] - base["shell_vy"] + rng.uniform(-14, 14, (n, 20)), base["shell_y"])
    pv = rng.integers(0, 14, n)
    return base, pv


def _lockstep_from(sfa, O, gametype, base, pv, rng, T, fused=False):
    n = len(base)
    env, orc = _load_both(sfa, O, gametype, base, prev_vlner=pv)
    acts = rng.integers(0, env.n_actions, (T, n)).astype(np.uint8)
    cps = {}
    if fused:
        a = torch.from_numpy(acts).to(env.device)
        obs, rew, done, info = (x.cpu().numpy() for x in env.rollout(a))
        done, info = done.astype(bool), info.astype(bool)
        cps[T - 1] = env.state_dict()
    else:
        obs, rew, done, info = run_device(env, acts, state_every=10, state_cb=lambda t, sd: cps.__setitem__(t, sd))
    n_done = 0
    for t in range(T):
        oo, orw, od, oi = orc.step(acts[t].astype(np.int32))
        assert np.array_equal(rew[t], orw), (t, np.flatnonzero(rew[t] != orw)[:5])
        assert np.array_equal(done[t], od) and np.array_equal(info[t], oi), t
        ok = obs_close(obs[t], oo, True)
        assert ok.all(), (t, np.argwhere(~ok)[:5])
        n_done += int(od.sum())
        if t in cps:
            bad = compare_state(cps[t], orc.snapshots())
            assert not bad, (t, bad)
    env.close()
    return n_done


@pytest.mark.parametrize("gametype", ["youturn", "autoturn", "test-youturn"])
def test_fuzzed_states(sfa, oracle_mod, gametype):
    """Constructed states loaded into both engines, then 40 random ticks in lock-step."""
    rng = np.random.default_rng(len(gametype) * 7)
    base, pv = _fuzz_base(oracle_mod, gametype, 1024, rng)
    _lockstep_from(sfa, oracle_mod, gametype, base, pv, rng, 40)


@pytest.mark.parametrize("gametype,fused", [("youturn", False), ("autoturn", False), ("youturn", True)])
def test_lanes_of_a_tile_finish_at_different_ticks(sfa, oracle_mod, gametype, fused):
    """Episodes out of step inside one 64-lane tile: every lane's clock is set so that its game ends at a tick of its
    own within the run, with up to 20 missiles of its own in the tile's shared pool at that moment.  A lane that starts
    a new game while its neighbours play on must take exactly its entries out of the pool (and nobody else's)."""
    rng = np.random.default_rng(11 + len(gametype) + int(fused))
    n, T = 1024, 48
    base, pv = _fuzz_base(oracle_mod, gametype, n, rng)
    base["time"] = 180000 - 34 * rng.integers(1, T - 6, n)  # game over (time >= 180000) after 1 .. T-7 more ticks
    base["time"][::7] = 34 * 100                              # ... and some lanes nowhere near it
    base["tick"] = base["time"] // 34
    n_done = _lockstep_from(sfa, oracle_mod, gametype, base, pv, rng, T, fused=fused)
    assert n_done >= n - n // 7 - 1


@pytest.mark.parametrize("n,f64", [(1, False), (63, True), (65, False), (1000, False), (257, True)])
def test_odd_batch_sizes_and_unaligned_outputs(sfa, oracle_mod, n, f64):
    """Batch sizes that are not multiples of the 64-lane tile / 256-lane workgroup, observation
    buffers that are not 16-byte aligned (views into a bigger tensor), and NULL output pointers:
    the padded lanes must stay invisible and the obs flush must take its scalar tail path."""
    import ctypes as C
    from spacefortress_amd import _lib

    O = oracle_mod
    T = 120
    rng = np.random.default_rng(n)
    dt = torch.float64 if f64 else torch.float32
    env = sfa.SFVecEnv(n, gametype="youturn", spawn_stride=2, obs_dtype=dt)
    orc = O.OracleVecEnv("youturn", n, spawn_stride=2)
    acts = rng.integers(0, 5, (T, n)).astype(np.int32)
    dev = env.device
    a = torch.from_numpy(acts).to(dev)
    # one big buffer, rows offset by 1 element so that most rows are NOT 16-byte aligned
    big = torch.zeros(T * (n * env.obs_dim + 1) + 8, dtype=dt, device=dev)
    rew = torch.empty((T, n), dtype=torch.int32, device=dev)
    done = torch.empty((T, n), dtype=torch.uint8, device=dev)
    info = torch.empty((T, n), dtype=torch.uint8, device=dev)
    views = []
    for t in range(T):
        off = 1 + t * (n * env.obs_dim + 1)
        v = big[off:off + n * env.obs_dim].view(n, env.obs_dim)
        views.append(v)
        env.step_tensors(a[t], out=(v, rew[t], done[t], info[t]))
    torch.cuda.synchronize()
    assert any(v.data_ptr() % 16 for v in views)
    for t in range(T):
        oo, orw, od, oi = orc.step(acts[t])
        assert np.array_equal(rew[t].cpu().numpy(), orw), t
        assert obs_close(views[t].cpu().numpy(), oo, f64).all(), t
    # nothing was written between or after the rows
    gaps = [float(big[t * (n * env.obs_dim + 1)]) for t in range(T)]
    assert not any(gaps) and not big[-8:].any()
    assert not compare_state(env.state_dict(), orc.snapshots())
    # every output pointer may be NULL (sfmi.h)
    L = _lib.lib()
    rc = L.sf_step(env._h, C.c_void_p(a[0].data_ptr()), 4, None, None, None, None, env._stream())
    assert rc == 0
    orc.step(acts[0])
    torch.cuda.synchronize()
    assert not compare_state(env.state_dict(), orc.snapshots())
    env.close()


@pytest.mark.parametrize("gametype,obs_type,f64", [("youturn", "features", False), ("autoturn", "features", True),
                                                  ("test-youturn", "normalized-features", False)])
def test_fused_rollout_equals_single_steps(sfa, oracle_mod, gametype, obs_type, f64):
    """sf_rollout (K ticks in one launch, state in registers) against K sf_step launches from the
    same start state -- every output and the final state bit-identical -- and against the oracle.
    The start state is constructed so that the fused loop crosses an episode end (auto-reset inside
    the launch), and some lanes carry more projectiles than the register-resident groups hold."""
    O = oracle_mod
    n, K = 1500, 96
    rng = np.random.default_rng(11)
    dt = torch.float64 if f64 else torch.float32
    base = O.OracleVecEnv(gametype, n, obs_type=obs_type, spawn_stride=1).snapshots()
    base["time"] = np.where(np.arange(n) % 3 == 0, 180000 - 34 * rng.integers(1, 60, n), 34 * rng.integers(0, 3000, n))
    base["tick"] = base["time"] // 34
    tab = np.load(os.path.join(GOLDEN, "tables.npz"))["missile_vel_by_angle"]
    crowd = np.flatnonzero(np.arange(n) % 5 == 0)
    ang = rng.integers(0, 360, (n, 20))
    base["missile_angle"] = ang
    base["missile_vx"] = tab[ang, 0]
    base["missile_vy"] = tab[ang, 1]
    base["missile_x"] = rng.uniform(100, 600, (n, 20))
    base["missile_y"] = rng.uniform(100, 520, (n, 20))
    base["shell_x"] = rng.uniform(50, 650, (n, 20))
    base["shell_y"] = rng.uniform(50, 570, (n, 20))
    base["shell_vx"] = rng.uniform(-4, 4, (n, 20))
    base["shell_vy"] = rng.uniform(-4, 4, (n, 20))
    for i in crowd:
        base["missile_alive"][i, rng.choice(20, rng.integers(10, 21), replace=False)] = 1
        base["shell_alive"][i, rng.choice(20, rng.integers(5, 16), replace=False)] = 1
    acts = rng.integers(0, 5 if "you" in gametype else 3, (K, n)).astype(np.uint8)
    envs = []
    for _ in range(2):
        env, orc = _load_both(sfa, O, gametype, base, obs_type=obs_type, spawn_stride=1)
        env.close()
        env = sfa.SFVecEnv(n, gametype=gametype, obs_type=obs_type, spawn_stride=1, obs_dtype=dt)
        for k, v in snapshots_to_fields(base).items():
            env.set_field(k, v)
        envs.append(env)
    single, fused = envs
    obs1, rew1, done1, info1 = run_device(single, acts)
    d_acts = torch.from_numpy(acts).to(fused.device)
    o2, r2, d2, i2 = fused.rollout(d_acts)
    torch.cuda.synchronize()
    fused.check_actions()
    assert np.array_equal(o2.cpu().numpy(), obs1)
    assert np.array_equal(r2.cpu().numpy(), rew1)
    assert np.array_equal(d2.cpu().numpy().astype(bool), done1) and np.array_equal(i2.cpu().numpy().astype(bool), info1)
    assert done1.any() and not done1.all()
    s1, s2 = single.state_dict(), fused.state_dict()
    live_m = lambda sd: ((sd["missile_mask"][None, :] >> np.arange(20, dtype=np.uint32)[:, None]) & 1).astype(bool)
    live_s = lambda sd: ((sd["shell_mask"][None, :] >> np.arange(20, dtype=np.uint32)[:, None]) & 1).astype(bool)
    for k in s1:
        a_, b_ = s1[k], s2[k]
        if k.startswith("missile_") and k != "missile_mask":
            a_, b_ = a_[live_m(s1)], b_[live_m(s2)]
        if k.startswith("shell_") and k != "shell_mask":
            a_, b_ = a_[live_s(s1)], b_[live_s(s2)]
        assert np.array_equal(a_, b_), k
    # and both equal the oracle
    _, orc = _load_both(sfa, O, gametype, base, obs_type=obs_type, spawn_stride=1)
    for t in range(K):
        oo, orw, od, oi = orc.step(acts[t].astype(np.int32))
        assert np.array_equal(rew1[t], orw), t
        assert obs_close(obs1[t], oo, f64).all(), t
    assert not compare_state(s2, orc.snapshots())
    # a second rollout continues from the registers' final state written back
    o3, r3, d3, i3 = fused.rollout(d_acts[:7])
    obs4, rew4, _, _ = run_device(single, acts[:7])
    assert np.array_equal(o3.cpu().numpy(), obs4) and np.array_equal(r3.cpu().numpy(), rew4)
    for e in envs:
        e.close()


def test_autoturn_heading_on_the_spawn_lattice(sfa, oracle_mod):
    """autoturn rounds atan2 up to an integer degree (SRC/game.cpp:317-319).  A spawned ship sits on
    integer coordinates, where the exact heading can BE an integer (axes, diagonals): the device
    atan2 must land on the same side as glibc's for every lattice point of the spawn box
    (SRC/game.cpp:137-138), and the fortress sector (ceil to 10 degrees, :206) too."""
    O = oracle_mod
    xs, ys = np.meshgrid(np.arange(170, 550), np.arange(150, 480), indexing="ij")
    pts = np.stack([xs.ravel(), ys.ravel()], 1)
    keep = ~((pts[:, 0] == 355) & (pts[:, 1] == 315))
    pts = pts[keep]
    # a stride keeps the CPU side quick while covering every diagonal/axis point explicitly
    diag = (np.abs(pts[:, 0] - 355) == np.abs(pts[:, 1] - 315)) | (pts[:, 0] == 355) | (pts[:, 1] == 315)
    sel = np.flatnonzero(diag | (np.arange(len(pts)) % 7 == 0))
    pts = pts[sel]
    n = len(pts)
    base = O.OracleVecEnv("autoturn", n).snapshots()
    base["ship_x"] = pts[:, 0]
    base["ship_y"] = pts[:, 1]
    base["ship_vx"] = 0.0  # stay on the lattice so the fortress sees integer coordinates too
    base["ship_vy"] = 0.0
    env, orc = _load_both(sfa, O, "autoturn", base)
    acts = np.zeros((1, n), np.uint8)
    obs, rew, done, info = run_device(env, acts)
    oo, orw, od, oi = orc.step(acts[0].astype(np.int32))
    sd, sn = env.state_dict(), orc.snapshots()
    assert np.array_equal(sd["ship_angle"].astype(np.float64), sn["ship_angle"]), \
        pts[sd["ship_angle"] != sn["ship_angle"]][:10]
    assert np.array_equal(sd["fort_angle"].astype(np.float64), sn["fort_angle"])
    assert not compare_state(sd, sn)
    assert obs_close(obs[0], oo, True).all()
    env.close()


@pytest.mark.parametrize("gametype", ["youturn", "autoturn"])
def test_bearings_a_whisker_off_the_axes(sfa, oracle_mod, gametype):
    """Ships a few ulps to 1e-9 off the fortress column / row: the bearing is a hair off +-90 or +-180 degrees --
    sector boundaries -- and the side the ROUNDED atan2 falls on decides the fortress sector (and the autoturn
    heading).  The device forms atan2 there as glibc rounds it (sf_kernels.hip: sf_atan2); a soak once found sector
    280 against the reference's 270 for a ship at x = 355 + 2^-44."""
    O = oracle_mod
    rng = np.random.default_rng(5)
    offs = np.concatenate([[0.0], 2.0 ** -np.arange(20, 46), -(2.0 ** -np.arange(20, 46)),
                           np.ldexp(rng.integers(1, 4096, 200).astype(np.float64), -rng.integers(30, 52, 200)) *
                           rng.choice([-1.0, 1.0], 200)])
    far = np.concatenate([rng.uniform(45, 190, 64), [101.72240646055117, 60.0, 150.5]])
    pts = []
    for o in offs:
        for f in far[rng.integers(0, len(far), 6)]:
            pts += [(355.0 + o, 315.0 + f), (355.0 + o, 315.0 - f), (355.0 - f, 315.0 + o), (355.0 + f, 315.0 + o)]
    pts = np.array(pts)
    n = len(pts)
    base = O.OracleVecEnv(gametype, n).snapshots()
    base["ship_x"] = pts[:, 0]
    base["ship_y"] = pts[:, 1]
    base["ship_vx"] = 0.0  # stay where they are
    base["ship_vy"] = 0.0
    env, orc = _load_both(sfa, O, gametype, base)
    acts = np.zeros((1, n), np.uint8)
    obs, rew, done, info = run_device(env, acts)
    oo, orw, od, oi = orc.step(acts[0].astype(np.int32))
    sd, sn = env.state_dict(), orc.snapshots()
    bad = np.flatnonzero(sd["fort_angle"].astype(np.float64) != sn["fort_angle"])
    assert bad.size == 0, (pts[bad][:5], sd["fort_angle"][bad][:5], sn["fort_angle"][bad][:5])
    assert np.array_equal(sd["ship_angle"].astype(np.float64), sn["ship_angle"])
    assert not compare_state(sd, sn)
    assert obs_close(obs[0], oo, True).all()
    env.close()


def _atan2_correctly_rounded(y, x):
    """atan2(y, x) of two doubles, correctly rounded, by 70-digit arithmetic: the libm value t0 plus
    atan((y cos t0 - x sin t0) / (x cos t0 + y sin t0)), sin and cos of t0 by their series."""
    import math
    from decimal import Decimal, getcontext
    getcontext().prec = 70
    t0 = math.atan2(y, x)
    T = Decimal(t0)
    s, c, term, k = Decimal(0), Decimal(0), Decimal(1), 0
    while abs(term) > Decimal(10) ** -68:
        if k % 2 == 0:
            c += term if k % 4 == 0 else -term
        else:
            s += term if k % 4 == 1 else -term
        k += 1
        term = term * T / k
    Y, X = Decimal(y), Decimal(x)
    t = (Y * c - X * s) / (X * c + Y * s)
    theta = T + t - t ** 3 / 3  # (|t| < 1e-15: the next term is below 1e-75)
    cands = [t0, math.nextafter(t0, math.inf), math.nextafter(t0, -math.inf)]
    return min(cands, key=lambda v: abs(Decimal(v) - theta))


@pytest.mark.parametrize("gametype", ["autoturn", "youturn"])
def test_bearings_on_exact_degree_rays(sfa, oracle_mod, gametype):
    """Ships on (and an ulp or two off) the rays of integer degrees from the fortress -- where an autoturn ship that
    thrusts at the fortress flies.  ceil() of the bearing follows the last bit of atan2 there.  The device forms atan2
    correctly rounded (sf_atan2, RAZOR); the reference calls glibc 2.35's, whose dbl-64 atan2 returns its last stage's
    value without the multi-precision check older versions had, and is off by one ulp on about 0.08 % of such arguments
    (tools/atan2_razor).  Reproducing THOSE roundings would take glibc's own tables (sysdeps/ieee754/dbl-64/uatan.tbl,
    atnat2.h: not in this image, no network), so the statement tested is: wherever device and reference differ, the
    argument is one that glibc itself does not round correctly -- arbitrated here with 70-digit arithmetic --, and
    there are few of them; everywhere else the two are identical.  (The plain device libm differs in a large fraction.)"""
    import math
    O = oracle_mod
    rng = np.random.default_rng(11)
    pts = []
    for k in range(0, 360):
        th = np.deg2rad(float(k))
        for r in (47.0, 86.5, 120.25, 173.0 + rng.uniform(0, 10)):
            x, y = 355.0 + r * np.cos(th), 315.0 + r * np.sin(th)
            pts += [(x, y), (np.nextafter(x, 1e9), y), (np.nextafter(x, -1e9), y), (x, np.nextafter(y, 1e9))]
    pts = np.array([p for p in pts if 40 < p[0] < 670 and 40 < p[1] < 590])
    n = len(pts)
    base = O.OracleVecEnv(gametype, n).snapshots()
    base["ship_x"] = pts[:, 0]
    base["ship_y"] = pts[:, 1]
    base["ship_vx"] = 0.0
    base["ship_vy"] = 0.0
    env, orc = _load_both(sfa, O, gametype, base)
    acts = np.zeros((1, n), np.uint8)
    run_device(env, acts)
    orc.step(acts[0].astype(np.int32))
    sd, sn = env.state_dict(), orc.snapshots()
    diff = (sd["fort_angle"].astype(np.float64) != sn["fort_angle"]) | (sd["ship_angle"].astype(np.float64) != sn["ship_angle"])
    assert diff.mean() <= 0.005, (int(diff.sum()), n, pts[diff][:5])
    # every difference: glibc's atan2 of the lane's bearing argument (SRC/game.cpp:194-197, :317-319: fortress -> ship or
    # ship -> fortress) is not the correctly rounded double
    unexplained = []
    for i in np.flatnonzero(diff):
        dx, dy = float(sn["ship_x"][i]) - 355.0, float(sn["ship_y"][i]) - 315.0
        if all(math.atan2(sy, sx) == _atan2_correctly_rounded(sy, sx) for sy, sx in ((dy, dx), (-dy, -dx))):
            unexplained.append((float(sn["ship_x"][i]), float(sn["ship_y"][i])))
    assert not unexplained, unexplained[:5]
    env.close()


@pytest.mark.parametrize("n", [256, 4096])
def test_a_fresh_batch_is_the_oracles_initial_state(sfa, oracle_mod, n):
    """Right after sf_create, and again right after sf_reset, every field of every lane is what the oracle holds
    (the kernels that write the state from scratch; a store-data hazard in them once left prev_vlner of four lanes
    in sixteen holding a neighbouring chunk's word until the first step rewrote it)."""
    O = oracle_mod
    for gametype in ("youturn", "autoturn"):
        env = sfa.SFVecEnv(n, gametype=gametype, spawn_stride=3, spawn_skip=1)
        orc = O.OracleVecEnv(gametype, n, spawn_stride=3, spawn_skip=1)
        sd = env.state_dict()
        assert not compare_state(sd, orc.snapshots())
        assert not np.asarray(sd["prev_vlner"]).any() and not np.asarray(sd["stats"]).any()
        assert not np.asarray(sd["ep_return"]).any() and not np.asarray(sd["last_reward"]).any()
        acts = np.random.default_rng(n).integers(0, env.n_actions, (40, n)).astype(np.uint8)
        run_device(env, acts)
        for t in range(40):
            orc.step(acts[t].astype(np.int32))
        env.reset()
        orc.reset()
        sd = env.state_dict()
        assert not compare_state(sd, orc.snapshots())
        assert np.array_equal(np.asarray(sd["prev_vlner"]).ravel(), orc.prev_vlner())
        assert not np.asarray(sd["stats"]).any()
        env.close()


def test_no_auto_reset_and_reset_keeps_prev_vlner(sfa, oracle_mod):
    """auto_reset=False is the bare SSF_Env: a finished game keeps ticking (ENV:246 only reports);
    sf_reset starts new games and keeps prev_vlner (ENV:92,163-178)."""
    O = oracle_mod
    z, meta = load("autoturn_destroy")
    acts = z["actions"][:110]  # ends with vulnerability > 0
    env = sfa.SFVecEnv(2, gametype="autoturn", auto_reset=False, obs_dtype=torch.float64)
    o = O.OracleEnv("autoturn")
    run_device(env, np.repeat(acts[:, None], 2, 1))
    out = o.replay(acts)
    pv = int(out["snaps"]["vlner"][-1])
    assert pv > 0 and (env.get_field("prev_vlner") == pv).all()
    obs = env.reset().cpu().numpy()
    o_obs = o.reset()
    assert (env.get_field("prev_vlner") == pv).all() and o.prev_vlner == pv
    assert obs_close(obs[0], o_obs, True).all()
    # first step of the new episode sees vlner_change = 0 - prev_vlner
    obs, rew, done, info = run_device(env, np.zeros((1, 2), np.uint8))
    oo, orw, od, oi = o.step(0)
    assert rew[0, 0] == orw == -1
    env.close()


def test_bad_arguments(sfa):
    with pytest.raises(RuntimeError):  # SRC/pymodule.cpp:341
        sfa.SFVecEnv(4, gametype="nope")
    with pytest.raises(ValueError):
        sfa.SFVecEnv(4, action_set=5)
    env = sfa.SFVecEnv(4)
    with pytest.raises(IndexError):  # ENV:211-212
        env.step(np.array([0, 1, 5, 0]))
    env.step_tensors(torch.tensor([0, 1, 9, 0], device=env.device))
    with pytest.raises(IndexError):
        env.check_actions()
    env.check_actions()  # cleared
    env.close()


def test_numpy_api_and_single_env(sfa, oracle_mod):
    """The numpy round trip returns the reference's kinds: float obs, int64 rewards, bool arrays."""
    O = oracle_mod
    env = sfa.SFVecEnv(8, gametype="youturn")
    obs = env.reset(numpy=True)
    assert obs.shape == (8, 19) and env.observation_space.shape == (19,) and env.action_space.n == 5
    obs, r, d, i = env.step(np.array([1, 2, 3, 4, 0, 1, 2, 3]))
    assert obs.dtype == np.float32 and r.dtype == np.int64 and d.dtype == bool and i.dtype == bool
    assert sum(i) == 0
    env.close()
    e1 = sfa.SSF_Env(gametype="autoturn", obs_type="features")
    o = O.OracleEnv("autoturn")
    assert e1.action_space.n == 3 and e1.tickdur == 34 and e1.max_ticks == 5294.0
    ob = e1.reset()  # the second Game of this "process": __init__ made the first (ENV:93)
    assert np.allclose(ob, o.reset(), rtol=0, atol=1e-9)
    for a in (1, 0, 2, 2, 1, 0):
        ob, r, d, i = e1.step(a)
        oo, orw, od, oi = o.step(a)
        assert (r, d, i) == (orw, od, oi) and isinstance(r, int)
        assert np.allclose(ob, oo, rtol=0, atol=1e-9)
    with pytest.raises(KeyError):
        e1.step(7)
    e1.close()


# ------------------------------------------------------------------ full-size properties

@pytest.mark.parametrize("gametype,n", [("youturn", 65536), ("autoturn", 65536), ("youturn", 4096)])
def test_full_size_properties(sfa, oracle_mod, gametype, n):
    """BASELINE.json batch sizes.  (1) a random sample of lanes equals the oracle run on just
    those lanes (lanes are independent); (2) two runs give identical bits; (3) invariants:
    deaths add up, points >= 0, time = 34 * steps, masks within 20 bits, shots counted."""
    O = oracle_mod
    T = 300 if n >= 65536 else 600
    rng = np.random.default_rng(n + len(gametype))
    n_act = 5 if gametype == "youturn" else 3
    acts = rng.integers(0, n_act, (T, n)).astype(np.uint8)
    outs = []
    for rep in range(2):
        env = sfa.SFVecEnv(n, gametype=gametype, spawn_stride=1)
        obs, rew, done, info = run_device(env, acts)
        sd = env.state_dict()
        outs.append((obs, rew, sd))
        env.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert np.array_equal(outs[0][2][k], outs[1][2][k]), k
    obs, rew, sd = outs[0]
    st = sd["stats"]
    assert np.array_equal(st[3], st[0] + st[1] + st[2])          # shipDeaths = bigHex + smallHex + shell
    assert (sd["points"] >= 0).all() and (sd["time"] == 34 * T).all()
    assert (sd["missile_mask"] < (1 << 20)).all() and (sd["shell_mask"] < (1 << 20)).all()
    press = (acts == 1)
    edges = press[0].astype(np.int64) + (press[1:] & ~press[:-1]).sum(0)
    assert np.array_equal(st[7], edges)                           # totalShots = FIRE press edges
    assert np.isfinite(obs).all()
    lanes = np.sort(rng.choice(n, 192, replace=False))
    # the oracle on the sampled lanes only (spawn_skip = lane because spawn_stride = 1)
    snaps = []
    for lane in lanes:
        o = O.OracleEnv(gametype, spawn_skip=int(lane))
        out = o.replay(acts[:, lane], want_obs=True)
        assert np.array_equal(out["reward"], rew[:, lane]), lane
        assert obs_close(obs[:, lane], out["obs"], False).all(), lane
        snaps.append(out["snaps"][-1])
    bad = compare_state(sd, np.array(snaps), lanes=lanes)
    assert not bad, bad


@pytest.mark.parametrize("gametype", ["autoturn", "youturn"])
def test_full_size_hunter_sample(sfa, oracle_mod, gametype):
    """BASELINE.json's batch size under play that destroys the fortress (random actions never do): 65 536 lanes on the
    hunter pattern (tests/sfscript.py), 192 lanes against the oracle step for step -- the lanes that scored a kill first,
    random ones for the rest -- with every reward, info flag and observation compared and the final state bit-exact.
    At least one kill must be in the sample, and the kills the wrapper reported (sum of info) must be the fortresses the
    engines destroyed (stats row 5)."""
    from sfscript import open_loop_actions

    O = oracle_mod
    n, T = 65536, 640
    rng = np.random.default_rng(77 + len(gametype))
    n_act = 5 if gametype == "youturn" else 3
    acts = open_loop_actions("hunter", (T, n), n_act, rng, phase=rng.integers(0, 96, n))
    a = torch.from_numpy(acts).cuda()
    # pass 1: which lanes kill
    env = sfa.SFVecEnv(n, gametype=gametype, spawn_stride=1)  # (a new batch is reset: lane i plays spawn i first)
    kills = torch.zeros(n, dtype=torch.int32, device=env.device)
    for t in range(T):
        _, _, _, info = env.step_tensors(a[t])
        kills += info
    kills = kills.cpu().numpy()
    sd = env.state_dict()
    env.close()
    # no episode ends inside 640 steps, so the counters are the run's: every kill the wrapper reported is a destroyed fortress;
    # the other way round a FIRE press on the kill tick hides the kill from the wrapper ((int)(-0.05 + 1) = 0, ENV:233:
    # the `autoturn_destroy_truncated` golden)
    destroyed = int(sd["stats"][5].sum())
    assert destroyed // 2 <= int(kills.sum()) <= destroyed, (int(kills.sum()), destroyed)
    assert kills.sum() >= (1000 if gametype == "autoturn" else 1), kills.sum()
    killers = np.flatnonzero(kills)
    pick = killers[:96]
    rest = np.setdiff1d(np.arange(n), pick)
    lanes = np.sort(np.concatenate([pick, rng.choice(rest, 192 - len(pick), replace=False)]))
    assert kills[lanes].sum() >= 1
    # pass 2 (same seeds, same actions: the same games): the sampled lanes' outputs at every step
    env = sfa.SFVecEnv(n, gametype=gametype, spawn_stride=1)  # (a new batch is reset: lane i plays spawn i first)
    li = torch.from_numpy(lanes).to(env.device)
    obs = torch.empty((T, len(lanes), env.obs_dim), dtype=torch.float32, device=env.device)
    rew = torch.empty((T, len(lanes)), dtype=torch.int32, device=env.device)
    inf = torch.empty((T, len(lanes)), dtype=torch.uint8, device=env.device)
    for t in range(T):
        o, r, _, i = env.step_tensors(a[t])
        obs[t], rew[t], inf[t] = o[li], r[li], i[li]
    obs, rew, inf = obs.cpu().numpy(), rew.cpu().numpy(), inf.cpu().numpy().astype(bool)
    sd2 = env.state_dict()
    env.close()
    for k in sd:
        assert np.array_equal(sd[k], sd2[k]), k
    snaps = []
    for j, lane in enumerate(lanes):
        o = O.OracleEnv(gametype, spawn_skip=int(lane))
        out = o.replay(acts[:, lane], want_obs=True)
        if not np.array_equal(out["reward"], rew[:, j]):
            t0 = int(np.flatnonzero(out["reward"] != rew[:, j])[0])
            raise AssertionError("lane %d: first reward mismatch at step %d: device %d oracle %d; actions %s; oracle obs %s; device obs %s; device obs before %s" % (
                lane, t0, rew[t0, j], out["reward"][t0], acts[max(0, t0 - 6):t0 + 1, lane].tolist(), np.array2string(out["obs"][t0], precision=6),
                np.array2string(obs[t0, j], precision=6), np.array2string(obs[t0 - 1, j], precision=6)))
        assert np.array_equal(out["info"], inf[:, j]), lane
        assert obs_close(obs[:, j], out["obs"], False).all(), lane
        assert out["info"].sum() == kills[lane] and out["snaps"][-1]["stats"][5] == sd["stats"][5][lane]
        snaps.append(out["snaps"][-1])
    bad = compare_state(sd, np.array(snaps), lanes=lanes)
    assert not bad, bad


@pytest.mark.parametrize("gametype,obs_type", [("youturn", "features"), ("autoturn", "normalized-features")])
def test_batches_beyond_one_wave_per_simd(sfa, oracle_mod, gametype, obs_type):
    """Up to 65 536 envs a wave is alone on its SIMD; beyond, several share one, and the hardware's timing differs: a 128-bit
    buffer store whose data register the next instruction overwrites lost lanes 12-15 of every 16 there (found in round 4:
    the compiler covers that hazard only for stores with an immediate soffset; sf_buf_st128 in sf_kernels.hip).  262 144
    envs in ONE batch must play exactly the games of the same envs in four batches of 65 536 -- every field of the state,
    bit for bit -- and 96 lanes of the big batch, the hazard's lanes among them, the oracle's games."""
    O = oracle_mod
    nb, n, T = 4, 65536, 160
    rng = np.random.default_rng(5 + len(gametype))
    n_act = 5 if gametype == "youturn" else 3
    acts = rng.integers(0, n_act, (T, nb * n)).astype(np.uint8)
    a = torch.from_numpy(acts).cuda()
    big = sfa.SFVecEnv(nb * n, gametype=gametype, obs_type=obs_type, spawn_stride=1)
    lanes = np.sort(np.concatenate([rng.choice(nb * n // 16, 48, replace=False) * 16 + rng.integers(12, 16, 48),
                                    rng.choice(nb * n, 48, replace=False)]))
    li = torch.from_numpy(lanes).to(big.device)
    rew = torch.empty((T, len(lanes)), dtype=torch.int32, device=big.device)
    obs = torch.empty((T, len(lanes), big.obs_dim), dtype=torch.float32, device=big.device)
    Ts = T - 40  # ... the last 40 ticks as ONE fused launch (sf_rollout: the other family of instantiations)
    for t in range(Ts):
        o, r, _, _ = big.step_tensors(a[t])
        rew[t], obs[t] = r[li], o[li]
    o, r, _, _ = big.rollout(a[Ts:].contiguous())
    rew[Ts:], obs[Ts:] = r[:, li], o[:, li]
    del o, r
    sb = big.state_dict()
    big.close()
    rew, obs = rew.cpu().numpy(), obs.cpu().numpy()
    for k in range(nb):
        e = sfa.SFVecEnv(n, gametype=gametype, obs_type=obs_type, spawn_stride=1, spawn_skip=k * n)
        ak = a[:, k * n:(k + 1) * n].contiguous()
        for t in range(Ts):
            e.step_tensors(ak[t])
        e.rollout(ak[Ts:].contiguous(), want_obs=False)
        sd = e.state_dict()
        e.close()
        for key in sd:
            x, y = np.asarray(sb[key])[..., k * n:(k + 1) * n], np.asarray(sd[key])
            assert x.tobytes() == y.tobytes(), (k, key, np.argwhere(x != y)[:8].tolist())
    snaps = []
    for j, lane in enumerate(lanes):
        o = O.OracleEnv(gametype, obs_type=obs_type, spawn_skip=int(lane))
        out = o.replay(acts[:, lane], want_obs=True)
        assert np.array_equal(out["reward"], rew[:, j]), lane
        assert obs_close(obs[:, j], out["obs"], False).all(), lane
        snaps.append(out["snaps"][-1])
    bad = compare_state(sb, np.array(snaps), lanes=lanes)
    assert not bad, bad


@pytest.mark.parametrize("gametype", ["youturn", "autoturn"])
def test_full_size_every_lane_vs_oracle(sfa, oracle_mod, gametype):
    """BASELINE.json's batch, ALL 65 536 lanes in lock-step with the oracle (its vector env is created in O(n) since round 4):
    240 steps of hunter play -- every reward, done, info and observation of every lane at every step, the full state of every
    lane at the end.  (tools/soak.py does the same for thousands of steps: profiles/r04_soak_65536.txt.)"""
    from sfscript import open_loop_actions

    O = oracle_mod
    n, T = 65536, 240
    rng = np.random.default_rng(41 + len(gametype))
    env = sfa.SFVecEnv(n, gametype=gametype, spawn_stride=1)
    orc = O.OracleVecEnv(gametype, n, spawn_stride=1)
    acts = open_loop_actions("hunter", (T, n), env.n_actions, rng, phase=rng.integers(0, 96, n))
    assert obs_close(env.reset().cpu().numpy(), orc.reset(), False).all()
    obs, rew, done, info = run_device(env, acts)
    for t in range(T):
        oo, orw, od, oi = orc.step(acts[t].astype(np.int32))
        assert np.array_equal(rew[t], orw), (t, np.flatnonzero(rew[t] != orw)[:5])
        assert np.array_equal(done[t], od) and np.array_equal(info[t], oi), t
        ok = obs_close(obs[t], oo, False)
        assert ok.all(), (t, np.argwhere(~ok)[:5])
    bad = compare_state(env.state_dict(), orc.snapshots())
    assert not bad, bad
    assert info.sum() > (200 if gametype == "autoturn" else 0)
    env.close()


@pytest.mark.parametrize("gametype,obs_type,f64", [("youturn", "features", False), ("autoturn", "normalized-features", True),
                                                   ("youturn", "monitors", False)])
def test_every_workgroup_size_plays_the_same_games(sfa, oracle_mod, gametype, obs_type, f64):
    """sf_launch_step picks 64-, 128- or 256-thread workgroups (and split launches) by the batch's size: different
    instantiations of the kernel.  Lane i of a batch with spawn_stride 1 plays spawn i whatever the batch's size, so the
    first 8 192 lanes of batches of 8 192 (64-thread workgroups), 24 576 (128), 49 152 (256 / split) and 90 112 (256, two
    waves on some SIMDs) must agree in every output of every step and every state field, single steps and a fused segment;
    64 of them are checked against the oracle."""
    O = oracle_mod
    n0, T = 8192, 220
    rng = np.random.default_rng(9 + len(gametype) + len(obs_type))
    n_act = 5 if gametype == "youturn" else 3
    acts0 = rng.integers(0, n_act, (T, n0)).astype(np.uint8)
    ref = None
    for n in (n0, 24576, 49152, 90112):
        acts = np.concatenate([acts0, rng.integers(0, n_act, (T, n - n0)).astype(np.uint8)], axis=1)
        a = torch.from_numpy(acts).cuda()
        env = sfa.SFVecEnv(n, gametype=gametype, obs_type=obs_type, spawn_stride=1, obs_dtype=torch.float64 if f64 else torch.float32)
        outs = []
        for t in range(T - 60):
            outs.append(tuple(x[:n0].clone() for x in env.step_tensors(a[t])))
        fo, fr, fd, fi = env.rollout(a[T - 60:].contiguous())
        obs = torch.cat([torch.stack([o[0] for o in outs]), fo[:, :n0]]).cpu().numpy()
        rew = torch.cat([torch.stack([o[1] for o in outs]), fr[:, :n0]]).cpu().numpy()
        done = torch.cat([torch.stack([o[2] for o in outs]), fd[:, :n0]]).cpu().numpy()
        info = torch.cat([torch.stack([o[3] for o in outs]), fi[:, :n0]]).cpu().numpy()
        sd = {k: np.asarray(v)[..., :n0] for k, v in env.state_dict().items()}
        env.close()
        if ref is None:
            ref = (obs, rew, done, info, sd)
            continue
        assert obs.tobytes() == ref[0].tobytes(), (n, np.argwhere(obs != ref[0])[:5].tolist())
        assert np.array_equal(rew, ref[1]) and np.array_equal(done, ref[2]) and np.array_equal(info, ref[3]), n
        for k in sd:
            assert sd[k].tobytes() == ref[4][k].tobytes(), (n, k, np.argwhere(sd[k] != ref[4][k])[:5].tolist())
    obs, rew, done, info, sd = ref
    lanes = np.sort(rng.choice(n0, 64, replace=False))
    snaps = []
    for lane in lanes:
        o = O.OracleEnv(gametype, obs_type=obs_type, spawn_skip=int(lane))
        out = o.replay(acts0[:, lane], want_obs=True)
        assert np.array_equal(out["reward"], rew[:, lane]), lane
        assert obs_close(obs[:, lane], out["obs"], f64).all(), lane
        snaps.append(out["snaps"][-1])
    bad = compare_state(sd, np.array(snaps), lanes=lanes)
    assert not bad, bad


def test_split_launches_on_every_scenario():
    """Batches up to 65 536 envs step by SPLIT launches (sf_step_kernel<..., 1000 + BLK>: a second wave per tile moves the
    missile pool and hands its events over through LDS), which is what every test of the default observation in this file
    runs on; here the scenarios run again in a child process with SFMI_FORCE_SPLIT=2 -- the launcher then splits every batch
    the instantiation serves, the ones beyond 65 536 envs too, and says so on stderr: goldens, random and hunter lock-steps,
    exhausted slots, pools beyond three rows, lanes of a tile finishing at different ticks (the purge that re-reads the rows
    the missile wave wrote), fuzzed states."""
    import subprocess

    env = dict(os.environ, SFMI_FORCE_SPLIT="2")
    sel = "golden or random or hunter or rollover or exhausted or many_shells or fuzzed or different_ticks or odd_batch or full_size_properties"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", sel, "-s",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout[-3000:] + r.stderr[-2000:])
    assert r.returncode == 0, tail
    assert "sfmi: split launch" in r.stderr + r.stdout, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail

