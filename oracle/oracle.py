"""ctypes bindings for the TEST-ONLY checker libraries.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from spacefortress_amd/ (the product).

* ``Oracle*``  -> oracle/libsforacle.so  (our C restatement, sf_oracle.c)
* ``Ref*``     -> oracle/_ref/libsfref.so (the real reference engine + ref_driver.cpp;
                  present only where oracle/Makefile could build it or gpurun shipped it)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libsforacle.so")
REF_SO = os.path.join(HERE, "_ref", "libsfref.so")
REFDRAW_SO = os.path.join(HERE, "_ref", "libsfrefdraw.so")  # engine + the reference's REAL renderer (cairo): `make refdraw`

MAXP = 20

# mirrors sfo_snapshot in sf_oracle.h (natural C alignment)
SNAPSHOT_DTYPE = np.dtype(
    [
        ("time", "<i4"), ("tick", "<i4"), ("ship_alive", "<i4"),
        ("ship_x", "<f8"), ("ship_y", "<f8"), ("ship_vx", "<f8"), ("ship_vy", "<f8"), ("ship_angle", "<f8"),
        ("ship_death_timer", "<i4"), ("fire_timer", "<i4"), ("thrust_timer", "<i4"),
        ("left_timer", "<i4"), ("right_timer", "<i4"),
        ("thrust_flag", "<i4"), ("fire_flag", "<i4"), ("left_flag", "<i4"), ("right_flag", "<i4"),
        ("turn_flag", "<i4"), ("fort_alive", "<i4"),
        ("fort_angle", "<f8"), ("fort_last_angle", "<f8"),
        ("fort_timer", "<i4"), ("fort_death_timer", "<i4"), ("fort_vuln_timer", "<i4"),
        ("points", "<f4"), ("raw_points", "<f4"), ("vlner", "<i4"),
        ("stats", "<i4", (13,)),
        ("vdir", "<f8"), ("fdist", "<f8"), ("ndist", "<f8"), ("aim", "<f8"),
        ("missile_alive", "<i4", (MAXP,)),
        ("missile_x", "<f8", (MAXP,)), ("missile_y", "<f8", (MAXP,)),
        ("missile_vx", "<f8", (MAXP,)), ("missile_vy", "<f8", (MAXP,)),
        ("missile_angle", "<f8", (MAXP,)),
        ("shell_alive", "<i4", (MAXP,)),
        ("shell_x", "<f8", (MAXP,)), ("shell_y", "<f8", (MAXP,)),
        ("shell_vx", "<f8", (MAXP,)), ("shell_vy", "<f8", (MAXP,)),
        ("shell_angle", "<f8", (MAXP,)),
        ("collisions", "<i4"),
    ],
    align=True,
)

OBS_TYPES = {"features": 0, "normalized-features": 1, "monitors": 2}
FIRE_KEY, THRUST_KEY, LEFT_KEY, RIGHT_KEY = 1, 2, 3, 4

_oracle = None
_ref = None


def build(force=False):
    """Compile the checker libraries (idempotent)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "sf_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", HERE, "liboracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/python/spacefortress/src"):
        # the reference engine; and, where the image has cairo (the build container: /opt/conda), its real renderer, its real
        # CPython extension and the cairo probe -- what the fixtures under tests/golden/{frames,getters} were recorded from
        subprocess.check_call(["make", "-C", HERE, "ref", "refdraw", "refpy", "cairoprobe"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)


def oracle_lib():
    global _oracle
    if _oracle is None:
        build()  # mtime check; compiles only when sf_oracle.c is newer
        L = C.CDLL(ORACLE_SO)
        assert L.sfo_snapshot_size() == SNAPSHOT_DTYPE.itemsize, (L.sfo_snapshot_size(), SNAPSHOT_DTYPE.itemsize)
        L.sfo_env_new.restype = C.c_void_p
        L.sfo_env_new.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint, C.c_int]
        L.sfo_env_free.argtypes = [C.c_void_p]
        L.sfo_env_game.restype = C.c_void_p
        L.sfo_env_game.argtypes = [C.c_void_p]
        for f in ("sfo_env_obs_dim", "sfo_env_n_actions", "sfo_env_prev_vlner"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.sfo_env_action_keys.argtypes = [C.c_void_p, C.c_int]
        L.sfo_env_set_faithful_bugs.argtypes = [C.c_void_p, C.c_int]
        L.sfo_env_set_ref_reset_obs.argtypes = [C.c_void_p, C.c_int]
        L.sfo_vec_env_at.restype = C.c_void_p
        L.sfo_vec_env_at.argtypes = [C.c_void_p, C.c_int]
        L.sfo_env_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.sfo_env_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sfo_env_features.argtypes = [C.c_void_p, C.c_void_p]
        L.sfo_env_snapshot.argtypes = [C.c_void_p, C.c_void_p]
        L.sfo_press_key.argtypes = [C.c_void_p, C.c_int]
        L.sfo_release_key.argtypes = [C.c_void_p, C.c_int]
        L.sfo_step_one_tick.argtypes = [C.c_void_p, C.c_int]
        L.sfo_is_game_over.argtypes = [C.c_void_p]
        L.sfo_rollout.restype = C.c_long
        L.sfo_rollout.argtypes = [C.c_void_p, C.c_long, C.c_uint]
        L.sfo_vec_create.restype = C.c_void_p
        L.sfo_vec_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int]
        L.sfo_vec_destroy.argtypes = [C.c_void_p]
        L.sfo_vec_obs_dim.argtypes = [C.c_void_p]
        L.sfo_vec_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.sfo_vec_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.sfo_vec_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.sfo_vec_prev_vlner.argtypes = [C.c_void_p, C.c_int]
        L.sfo_env_hex_points.argtypes = [C.c_void_p, C.c_void_p]
        L.sfo_env_load_snapshot.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.sfo_vec_load_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.sfo_env_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int]
        L.sfo_srand.argtypes = [C.c_void_p, C.c_uint]
        L.sfo_rand.argtypes = [C.c_void_p]
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.sfref_create.restype = C.c_void_p
        L.sfref_create.argtypes = [C.c_char_p, C.c_uint, C.c_int]
        L.sfref_destroy.argtypes = [C.c_void_p]
        L.sfref_new_game.argtypes = [C.c_void_p]
        L.sfref_press_key.argtypes = [C.c_void_p, C.c_int]
        L.sfref_release_key.argtypes = [C.c_void_p, C.c_int]
        L.sfref_step_one_tick.argtypes = [C.c_void_p, C.c_int]
        L.sfref_is_game_over.argtypes = [C.c_void_p]
        L.sfref_snapshot.argtypes = [C.c_void_p, C.c_void_p]
        L.sfref_hex_points.argtypes = [C.c_void_p, C.c_void_p]
        L.sfref_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sfref_rollout.restype = C.c_long
        L.sfref_rollout.argtypes = [C.c_void_p, C.c_long, C.c_uint]
        _ref = L
    return _ref


def have_refdraw():
    return os.path.exists(REFDRAW_SO)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleRng:
    """glibc TYPE_3 rand() restatement (sfo_srand / sfo_rand)."""

    def __init__(self, seed=1):
        self.L = oracle_lib()
        self.buf = (C.c_int32 * 33)()
        self.L.sfo_srand(self.buf, seed)

    def rand(self):
        return self.L.sfo_rand(self.buf)


class _GameApi:
    """Engine-level surface shared by the oracle and the reference driver
    (`_spacefortress.Game` of SRC/pymodule.cpp:361-370)."""

    def apply_keys(self, keys, youturn):
        # ENV:213-229
        (self.press_key if keys & 1 else self.release_key)(FIRE_KEY)
        (self.press_key if keys & 2 else self.release_key)(THRUST_KEY)
        if youturn:
            (self.press_key if keys & 4 else self.release_key)(LEFT_KEY)
            (self.press_key if keys & 8 else self.release_key)(RIGHT_KEY)


class OracleEnv(_GameApi):
    """One SSF_Env restatement (sfo_env)."""

    def __init__(self, gametype="youturn", action_set=1, obs_type="features", seed=1, spawn_skip=0):
        self.L = oracle_lib()
        self.h = self.L.sfo_env_new(gametype.encode(), action_set, OBS_TYPES[obs_type], seed, spawn_skip)
        if not self.h:
            raise RuntimeError("unknown gametype/action_set: %r %r" % (gametype, action_set))
        self.g = self.L.sfo_env_game(self.h)
        self.gametype = gametype
        self.youturn = gametype in ("youturn", "test-youturn")
        self.obs_dim = self.L.sfo_env_obs_dim(self.h)
        self.n_actions = self.L.sfo_env_n_actions(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sfo_env_free(self.h)
            self.h = None

    def action_keys(self):
        return [self.L.sfo_env_action_keys(self.h, a) for a in range(self.n_actions)]

    def set_faithful_bugs(self, on):
        self.L.sfo_env_set_faithful_bugs(self.h, int(on))

    def set_ref_reset_obs(self, on):
        """A new Game's aim / vdir / ndist read 0 until its first tick, as in the reference on fresh memory (next reset on)."""
        self.L.sfo_env_set_ref_reset_obs(self.h, int(on))

    @property
    def prev_vlner(self):
        return self.L.sfo_env_prev_vlner(self.h)

    def reset(self):
        obs = np.empty(self.obs_dim, np.float64)
        self.L.sfo_env_reset(self.h, _ptr(obs))
        return obs

    def features(self):
        obs = np.empty(self.obs_dim, np.float64)
        self.L.sfo_env_features(self.h, _ptr(obs))
        return obs

    def step(self, action):
        obs = np.empty(self.obs_dim, np.float64)
        r, d, i = C.c_int(), C.c_int(), C.c_int()
        rc = self.L.sfo_env_step(self.h, int(action), _ptr(obs), C.byref(r), C.byref(d), C.byref(i))
        if rc != 0:
            raise IndexError("action %r out of range" % (action,))
        return obs, r.value, bool(d.value), bool(i.value)

    # engine-level
    def press_key(self, k):
        self.L.sfo_press_key(self.g, k)

    def release_key(self, k):
        self.L.sfo_release_key(self.g, k)

    def step_one_tick(self, ms=34):
        return self.L.sfo_step_one_tick(self.g, ms)

    def is_game_over(self):
        return bool(self.L.sfo_is_game_over(self.g))

    def new_game(self):
        self.L.sfo_env_reset(self.h, None)

    def snapshot(self):
        s = np.zeros((), SNAPSHOT_DTYPE)
        self.L.sfo_env_snapshot(self.h, _ptr(s))
        return s

    def rollout(self, n_steps, lcg_seed):
        return self.L.sfo_rollout(self.h, n_steps, lcg_seed)

    def replay(self, actions, want_obs=True, max_resets=8):
        """Run len(actions) wrapper steps (vec-env reset after done) in one call.
        Returns dict(snaps, obs, reward, done, info, reset_snaps)."""
        a = np.ascontiguousarray(actions, np.uint8)
        T = len(a)
        snaps = np.zeros(T, SNAPSHOT_DTYPE)
        obs = np.empty((T, self.obs_dim), np.float64) if want_obs else None
        reward = np.empty(T, np.int32)
        done = np.empty(T, np.uint8)
        info = np.empty(T, np.uint8)
        rs = np.zeros(max_resets, SNAPSHOT_DTYPE)
        n = self.L.sfo_env_replay(self.h, _ptr(a), T, _ptr(snaps), _ptr(obs), _ptr(reward), _ptr(done),
                                  _ptr(info), _ptr(rs), max_resets)
        if n < 0:
            raise IndexError("action out of range at step %d" % (-1 - n))
        return dict(snaps=snaps, obs=obs, reward=reward, done=done.astype(bool), info=info.astype(bool),
                    reset_snaps=rs[:min(n, max_resets)])

    def hex_points(self):
        out = np.empty(24, np.float64)
        self.L.sfo_env_hex_points(self.h, _ptr(out))
        return out


class OracleVecEnv:
    """SubprocVecEnv-shaped batch of oracle envs with auto-reset."""

    def __init__(self, gametype="youturn", n=1, action_set=1, obs_type="features", seed=1,
                 spawn_skip=0, spawn_stride=0):
        self.L = oracle_lib()
        self.h = self.L.sfo_vec_create(gametype.encode(), n, action_set, OBS_TYPES[obs_type], seed,
                                       spawn_skip, spawn_stride)
        if not self.h:
            raise RuntimeError("sfo_vec_create failed")
        self.n = n
        self.obs_dim = self.L.sfo_vec_obs_dim(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sfo_vec_destroy(self.h)
            self.h = None

    def reset(self):
        obs = np.empty((self.n, self.obs_dim), np.float64)
        self.L.sfo_vec_reset(self.h, _ptr(obs))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.int32)
        obs = np.empty((self.n, self.obs_dim), np.float64)
        rew = np.empty(self.n, np.int32)
        done = np.empty(self.n, np.uint8)
        info = np.empty(self.n, np.uint8)
        self.L.sfo_vec_step(self.h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done), _ptr(info))
        return obs, rew, done.astype(bool), info.astype(bool)

    def set_ref_reset_obs(self, on):
        for i in range(self.n):
            self.L.sfo_env_set_ref_reset_obs(self.L.sfo_vec_env_at(self.h, i), int(on))

    def snapshots(self):
        s = np.zeros(self.n, SNAPSHOT_DTYPE)
        for i in range(self.n):
            self.L.sfo_vec_snapshot(self.h, i, C.c_void_p(s.ctypes.data + i * SNAPSHOT_DTYPE.itemsize))
        return s

    def prev_vlner(self):
        return np.array([self.L.sfo_vec_prev_vlner(self.h, i) for i in range(self.n)], np.int32)

    def load_snapshots(self, snaps, prev_vlner=None):
        """Put every env into a constructed state (snaps: SNAPSHOT_DTYPE[n])."""
        snaps = np.ascontiguousarray(snaps, SNAPSHOT_DTYPE)
        for i in range(self.n):
            pv = 0 if prev_vlner is None else int(prev_vlner[i])
            self.L.sfo_vec_load_snapshot(self.h, i, C.c_void_p(snaps.ctypes.data + i * SNAPSHOT_DTYPE.itemsize), pv)


class RefGame(_GameApi):
    """The real reference engine (oracle/_ref/libsfref.so)."""

    def __init__(self, gametype="youturn", seed=1, spawn_skip=0):
        self.L = ref_lib()
        self.h = self.L.sfref_create(gametype.encode(), seed, spawn_skip)
        if not self.h:
            raise RuntimeError("unknown gametype %r" % (gametype,))
        self.gametype = gametype
        self.youturn = gametype in ("youturn", "test-youturn")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sfref_destroy(self.h)
            self.h = None

    def press_key(self, k):
        self.L.sfref_press_key(self.h, k)

    def release_key(self, k):
        self.L.sfref_release_key(self.h, k)

    def step_one_tick(self, ms=34):
        return self.L.sfref_step_one_tick(self.h, ms)

    def is_game_over(self):
        return bool(self.L.sfref_is_game_over(self.h))

    def new_game(self):
        self.L.sfref_new_game(self.h)

    def snapshot(self):
        s = np.zeros((), SNAPSHOT_DTYPE)
        self.L.sfref_snapshot(self.h, _ptr(s))
        return s

    def rollout(self, n_steps, lcg_seed):
        return self.L.sfref_rollout(self.h, n_steps, lcg_seed)

    def durations(self, which):
        """0 thrust_durations, 1 shot_durations, 2 shot_intervals_invul, 3 shot_intervals_vul (SRC/game.hh:98-101)."""
        buf = np.zeros(8192, np.int32)
        self.L.sfref_durations.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        n = self.L.sfref_durations(self.h, which, _ptr(buf), len(buf))
        assert n <= len(buf)
        return buf[:n].copy()

    def dump(self):
        """Game::dumpState(), the string `Game.dump()` returns (SRC/pymodule.cpp:361-370)."""
        buf = C.create_string_buffer(8192)
        self.L.sfref_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        n = self.L.sfref_dump(self.h, buf, len(buf))
        assert n < len(buf)
        return buf.value.decode()

    def replay(self, keys):
        """Run len(keys) engine ticks of key bits in one call (new Game after game over)."""
        k = np.ascontiguousarray(keys, np.uint8)
        T = len(k)
        snaps = np.zeros(T, SNAPSHOT_DTYPE)
        eng = np.empty(T, np.int32)
        done = np.empty(T, np.uint8)
        self.L.sfref_replay(self.h, _ptr(k), T, _ptr(snaps), _ptr(eng), _ptr(done))
        return dict(snaps=snaps, eng_reward=eng, done=done.astype(bool))

    def hex_points(self):
        out = np.empty(24, np.float64)
        self.L.sfref_hex_points(self.h, _ptr(out))
        return out


class RefDrawGame(RefGame):
    """The reference engine AND its real renderer (SRC/draw.cpp, SRC/wireframe.cpp against the image's cairo 1.16):
    oracle/_ref/libsfrefdraw.so, built by `make -C oracle refdraw` in the build container only.  Used by
    tests/golden/frames/make_frames_golden.py to draw the frame fixtures and by the live frame tests."""

    _lib = None

    def __init__(self, gametype="youturn", seed=1, spawn_skip=0):
        if RefDrawGame._lib is None:
            L = C.CDLL(REFDRAW_SO)
            L.sfref_create.restype = C.c_void_p
            L.sfref_create.argtypes = [C.c_char_p, C.c_uint, C.c_int]
            L.sfref_destroy.argtypes = [C.c_void_p]
            L.sfref_new_game.argtypes = [C.c_void_p]
            L.sfref_press_key.argtypes = [C.c_void_p, C.c_int]
            L.sfref_release_key.argtypes = [C.c_void_p, C.c_int]
            L.sfref_step_one_tick.argtypes = [C.c_void_p, C.c_int]
            L.sfref_is_game_over.argtypes = [C.c_void_p]
            L.sfref_snapshot.argtypes = [C.c_void_p, C.c_void_p]
            L.sfref_load_snapshot.argtypes = [C.c_void_p, C.c_void_p]
            L.sfref_hex_points.argtypes = [C.c_void_p, C.c_void_p]
            L.sfref_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            L.sfref_rollout.restype = C.c_long
            L.sfref_rollout.argtypes = [C.c_void_p, C.c_long, C.c_uint]
            L.sfref_draw.argtypes = [C.c_void_p, C.c_void_p]
            L.sfref_draw_geom.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_double, C.c_int, C.c_int, C.c_void_p]
            L.sfref_cairo_version.restype = C.c_char_p
            RefDrawGame._lib = L
        self.L = RefDrawGame._lib
        self.h = self.L.sfref_create(gametype.encode(), seed, spawn_skip)
        if not self.h:
            raise RuntimeError("unknown gametype %r" % (gametype,))
        self.gametype = gametype
        self.youturn = gametype in ("youturn", "test-youturn")

    def load_snapshot(self, snap):
        s = np.ascontiguousarray(snap, SNAPSHOT_DTYPE).reshape(())
        self.L.sfref_load_snapshot(self.h, _ptr(s))

    def draw(self, scale=None, viewport=(130, 80, 450, 460), ls=3.0, grayscale=True, channel=0):
        """What Game.draw() + pb_pixels give SSF_Env._draw (ENV:203-206), one channel: [int(vh * scale)][int(vw * scale)] uint8."""
        if scale is None:
            out = np.zeros((92, 90), np.uint8)
            assert self.L.sfref_draw(self.h, _ptr(out)) == 0
            return out
        w, h = int(viewport[2] * scale), int(viewport[3] * scale)
        out = np.zeros((h, w), np.uint8)
        assert self.L.sfref_draw_geom(self.h, w, h, int(viewport[0]), int(viewport[1]), int(viewport[2]), int(viewport[3]),
                                      float(ls), int(grayscale), channel, _ptr(out)) == 0
        return out

    def cairo_version(self):
        return self.L.sfref_cairo_version().decode()
