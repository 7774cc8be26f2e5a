"""Game -- a single-lane view of the batched engine with the shape of the reference's CPython type
`_spacefortress.Game` (SRC/pymodule.cpp:319-411), for scripts that poke `sf.Game` directly and for
single-lane debugging (SURVEY 8b "optional Game-shaped shim", 8f rank 4).

    g = Game("youturn"); g.press_key(FIRE_KEY); r = g.step_one_tick(34); g.ship_x; g.missiles; g.dump()

It is a batch of ONE lane of libsfmi (no CPU engine behind it).  Differences from the reference, by design:
  * key events are applied as the key STATE at the tick (what SSF_Env.step produces: one press or release
    per key and tick, ENV:213-229); several presses of one key inside one tick collapse;
  * `events` / `dump()` rebuild the tick's event list on the host from the recorded key calls and the state
    change of the tick (exact on the recorded reference runs in tests/golden/telemetry; the order of two
    missile hits inside ONE tick is not recoverable and is emitted in hit-count order);
  * `thrust_durations`, `shot_durations`, `shot_intervals_*` are kept on the host by this view (they are pure
    functions of the key edges and of the timers / vulnerability before the tick, SRC/game.cpp:237-261);
  * `vulnerability_timer` / `vulnerability_time` return the values the reference's getters intend (their
    "d"-format-with-int bug is undefined behaviour, SRC/pymodule.cpp:44-45).
"""
import math

import numpy as np

FIRE_KEY, THRUST_KEY, LEFT_KEY, RIGHT_KEY = 1, 2, 3, 4
_KEY_BIT = {FIRE_KEY: 1, THRUST_KEY: 2, LEFT_KEY: 4, RIGHT_KEY: 8}
_KEY_NAME = {FIRE_KEY: "fire", THRUST_KEY: "thrust", LEFT_KEY: "left", RIGHT_KEY: "right"}
_STAT = ("bigHexDeaths", "smallHexDeaths", "shellDeaths", "shipDeaths", "resets", "destroyedFortresses", "missedShots",
         "totalShots", "totalThrusts", "totalLefts", "totalRights", "vlnerIncs", "maxVlner")


class Game:
    def __init__(self, config, lw=2.0, grayscale=0, width=-1, height=-1, viewport=(0, 0, -1, -1), device=None, seed=1):
        import ctypes as C
        import torch
        from . import _lib
        from .vecenv import SFVecEnv

        self._torch, self._C, self._lib = torch, C, _lib
        # every key combination as an action (ENV:64-89 with action_set 0), no wrapper auto-reset
        self._vec = SFVecEnv(1, gametype=config, obs_type="features", action_set=0, device=device, seed=seed,
                             obs_dtype=torch.float64, auto_reset=False)
        keys = (C.c_uint8 * 16)()
        n = _lib.lib().sf_action_table(config.encode(), 0, keys)
        self._action_of = {int(keys[i]): i for i in range(n)}
        self._n_keys = 4 if n == 16 else 2
        self._config = config
        self._keys = 0
        self._calls = []       # (pressed, sym) since the last tick, in call order
        self._events = ()
        # SRC/game.hh:98-101, cleared by the constructor (SRC/game.cpp:64-67)
        self.thrust_durations, self.shot_durations = (), ()
        self.shot_intervals_invul, self.shot_intervals_vul = (), ()
        self._obs = None
        self._sd = None
        self.pb_width, self.pb_height = (width, height) if width > 0 else (90, 92)
        p = _lib.Preset()
        _lib.check(_lib.lib().sf_preset_get(config.encode(), C.byref(p)))
        self._preset = p
        self._owns = True

    @classmethod
    def _over(cls, vec, config, action_set, width, height):
        """The Game of an SSF_Env (`env.g`, ENV:164): a view over the env's own batch of one -- the same lane the env steps,
        no second engine.  Reading is what the reference offers on it (`env.g.points`, `.missiles`, `.timers`, `.stats`,
        `dump()`, `draw()` / `pb_pixels`); the env's step() tells the view the key calls of the tick (_env_tick), so
        `events` and the duration vectors follow the env's game."""
        import ctypes as C
        import torch
        from . import _lib

        g = cls.__new__(cls)
        g._torch, g._C, g._lib = torch, C, _lib
        g._vec = vec
        keys = (C.c_uint8 * 16)()
        n = _lib.lib().sf_action_table(config.encode(), int(action_set), keys)
        g._action_of = {int(keys[i]): i for i in range(n)}
        g._keys_of = [int(keys[i]) for i in range(n)]
        g._n_keys = 4 if config in ("youturn", "test-youturn") else 2
        g._config = config
        g._keys = 0
        g._calls = []
        g._events = ()
        g.thrust_durations, g.shot_durations = (), ()
        g.shot_intervals_invul, g.shot_intervals_vul = (), ()
        g._obs = None
        g._sd = None
        g.pb_width, g.pb_height = width, height
        p = _lib.Preset()
        _lib.check(_lib.lib().sf_preset_get(config.encode(), C.byref(p)))
        g._preset = p
        g._owns = False
        return g

    def _env_tick(self, action, step):
        """One SSF_Env.step through the view: the key calls ENV:213-229 makes for `action` (fire, thrust, then left, right
        in the game types that turn), the tick itself (`step()`: the env's own call into the batch), the tick's events."""
        keys = self._keys_of[action]
        self._calls = [(bool(keys & 1), FIRE_KEY), (bool(keys & 2), THRUST_KEY)]
        if self._n_keys == 4:
            self._calls += [(bool(keys & 4), LEFT_KEY), (bool(keys & 8), RIGHT_KEY)]
        self._keys = keys
        self._sd = None
        before = self._state()
        self._telemetry_vectors(before)
        out = step()
        self._sd = None
        after = self._state()
        self._events = tuple(self._derive_events(before, after))
        self._calls = []
        return out

    def _new_game(self):
        """SSF_Env.reset(): `self.g = sf.Game(...)` -- the vectors and events of a fresh Game."""
        self._keys, self._calls, self._events = 0, [], ()
        self.thrust_durations, self.shot_durations = (), ()
        self.shot_intervals_invul, self.shot_intervals_vul = (), ()
        self._obs, self._sd = None, None
        if hasattr(self, "_frame"):
            del self._frame

    # ------------------------------------------------------------------ methods (SRC/pymodule.cpp:361-370)
    def press_key(self, sym):
        self._key_call(True, sym)

    def release_key(self, sym):
        self._key_call(False, sym)

    def _key_call(self, pressed, sym):
        if sym not in _KEY_BIT:
            raise ValueError("unknown key %r" % (sym,))
        if _KEY_BIT[sym] >= (1 << self._n_keys):
            raise ValueError("this game type steps with FIRE and THRUST only (ENV:213-220)")
        self._calls.append((pressed, sym))
        self._keys = (self._keys | _KEY_BIT[sym]) if pressed else (self._keys & ~_KEY_BIT[sym])

    def step_one_tick(self, ms):
        if not self._owns:
            raise RuntimeError("this Game is an SSF_Env's (env.g): the env steps it (env.step); a Game of its own is "
                               "spacefortress.core.Game(...)")
        if ms != self._vec.tickdur:
            raise ValueError("the device engine ticks in %d ms steps (ENV:61)" % self._vec.tickdur)
        before = self._state()
        self._telemetry_vectors(before)
        a = self._torch.tensor([self._action_of[self._keys]], dtype=self._torch.uint8, device=self._vec.device)
        obs, _, _, _ = self._vec.step_tensors(a)
        self._obs = obs[0].cpu().numpy()
        self._sd = None
        after = self._state()
        self._events = tuple(self._derive_events(before, after))
        self._calls = []
        return int(after["last_reward"])

    def _telemetry_vectors(self, b):
        """processKeyState's pushes (SRC/game.cpp:237-261), from the key calls of this tick in call order and
        the flags / timers / vulnerability they meet."""
        fl = int(b["flags"])
        fire, thrust = bool(fl & 4), bool(fl & 8)
        fire_t, thrust_t = int(b["fire_timer"]), int(b["thrust_timer"])
        for pressed, sym in self._calls:
            if sym == FIRE_KEY and pressed and not fire:
                if int(b["vlner"]) > 10:
                    self.shot_intervals_vul += (abs(fire_t),)
                else:
                    self.shot_intervals_invul += (abs(fire_t),)
                fire, fire_t = True, 0
            elif sym == FIRE_KEY and not pressed and fire:
                self.shot_durations += (fire_t,)
                fire, fire_t = False, 0
            elif sym == THRUST_KEY and pressed and not thrust:
                thrust, thrust_t = True, 0
            elif sym == THRUST_KEY and not pressed and thrust:
                self.thrust_durations += (thrust_t,)
                thrust, thrust_t = False, 0

    def is_game_over(self):
        return bool(self.time >= self.max_time)

    def draw(self):
        self._frame = self._vec.render("image-raw")[0].cpu().numpy()

    @property
    def pb_pixels(self):
        """The surface as the reference exposes it: rows of 4 bytes per pixel (B, G, R, x); the grey value three times and
        255 -- what cairo's ARGB32 surface holds there behind an opaque paint (tests/golden/getters: `frames`)."""
        f = getattr(self, "_frame", None)
        if f is None:
            self.draw()
            f = self._frame
        px = np.repeat(f[:, :, None], 4, axis=2)
        px[:, :, 3] = 255
        return px.tobytes()

    def config(self, key):
        p = self._preset
        table = {"width": p.width, "height": p.height, "gameTime": p.game_time, "destroyFortress": p.destroy_fortress,
                 "shipDeathPenalty": p.ship_death_penalty, "missilePenalty": p.missile_penalty, "shellSpeed": p.shell_speed,
                 "missileSpeed": p.missile_speed, "autoTurn": p.auto_turn, "fortressSectorSize": p.sector_size,
                 "fortressLockTime": p.lock_time, "fortressVulnerabilityTime": p.vuln_time,
                 "fortressVulnerabilityThreshold": p.vuln_threshold, "bigHex": p.big_hex, "smallHex": p.small_hex,
                 "shipExplodeDuration": p.explode_duration, "shipAcceleration": p.ship_accel, "shipTurnSpeed": p.turn_speed}
        if key not in table:
            raise ValueError("No config value for `%s'" % key)  # SRC/pymodule.cpp:273
        return table[key]

    def close(self):
        if self._owns:
            self._vec.close()

    # ------------------------------------------------------------------ state
    def _state(self):
        if self._sd is None:
            self._sd = {k: (v[0] if v.ndim == 1 else v[:, 0]) for k, v in self._vec.state_dict().items()}
        return self._sd

    def _f(self, name):
        return self._state()[name]

    def _features(self):
        if self._obs is None:  # before the first tick: the reset observation
            self._obs = self._vec.render_features() if hasattr(self._vec, "render_features") else None
        return self._obs

    tick = property(lambda s: int(s._f("time")) // s._vec.tickdur)
    time = property(lambda s: int(s._f("time")))
    max_time = property(lambda s: int(s._preset.game_time))
    ship_alive = property(lambda s: bool(int(s._f("flags")) & 1))
    ship_x = property(lambda s: float(s._f("ship_x")))
    ship_y = property(lambda s: float(s._f("ship_y")))
    ship_vx = property(lambda s: float(s._f("ship_vx")))
    ship_vy = property(lambda s: float(s._f("ship_vy")))
    ship_angle = property(lambda s: float(s._f("ship_angle")))
    fortress_alive = property(lambda s: bool(int(s._f("flags")) & 2))
    fortress_angle = property(lambda s: float(s._f("fort_angle")))
    bighex = property(lambda s: int(s._preset.big_hex))
    smallhex = property(lambda s: int(s._preset.small_hex))
    points = property(lambda s: float(s._f("points")))
    raw_points = property(lambda s: float(s._f("raw_points")))
    vulnerability = property(lambda s: int(s._f("vlner")))
    vulnerability_timer = property(lambda s: float(s._f("fort_vuln_timer")))
    vulnerability_time = property(lambda s: float(s._preset.vuln_time))
    thrust_flag = property(lambda s: bool(int(s._f("flags")) & 8))
    events = property(lambda s: s._events)

    @property
    def turn_flag(self):  # NO_TURN, TURN_LEFT, TURN_RIGHT; both keys held cancel (SRC/game.cpp:263-270)
        left, right = bool(int(self._f("flags")) & 16), bool(int(self._f("flags")) & 32)
        return 1 if left and not right else (2 if right and not left else 0)

    def _extra(self, i):
        if self._obs is None:
            raise AttributeError("vdir / aim / ndist are defined after the first tick (mExtra is uninitialised before); on an "
                                 "SSF_Env's Game they are read from the env's 'features' observation")
        return float(self._obs[i])

    aim = property(lambda s: s._extra(6))    # feature order ENV:134-157
    vdir = property(lambda s: s._extra(7))
    ndist = property(lambda s: s._extra(8))

    @property
    def missiles(self):
        st = self._state()
        m = int(st["missile_mask"])
        return tuple((float(st["missile_x"][i]), float(st["missile_y"][i]), float(st["missile_angle"][i]))
                     for i in range(20) if (m >> i) & 1)

    @property
    def shells(self):
        # the reference's getter walks the MISSILES (SRC/pymodule.cpp:131-134); `real_shells` is the intent
        return self.missiles

    @property
    def real_shells(self):
        st = self._state()
        m = int(st["shell_mask"])
        return tuple((float(st["shell_x"][i]), float(st["shell_y"][i]), _shell_angle(st["shell_vx"][i], st["shell_vy"][i]))
                     for i in range(20) if (m >> i) & 1)

    @property
    def stats(self):
        # the reference's counters are plain ints (SRC/game.hh:29-43); here they ride in bit fields sized for one game,
        # and a Game that is ticked for several games' worth without a new one is told so instead of reading wrapped values
        self._vec.check_state()
        st = self._state()
        return tuple(int(v) for v in st["stats"]) + (float(st["points"]), float(st["raw_points"]))

    @property
    def timers(self):
        self._vec.check_state()  # a key timer wraps after 32 767 ticks without an edge of its key
        st = self._state()
        return tuple(int(st[k]) for k in ("fire_timer", "thrust_timer", "left_timer", "right_timer"))

    @property
    def collisions(self):
        names = []  # the reference fills the tuple back to front: shell, missile, smallhex, bighex
        ev = set(self._events)
        if "shell-hit-ship" in ev:
            names.append("shell")
        if ev & {"hit-fortress", "hit-dead-fortress"}:
            names.append("missile")
        if "explode-smallhex" in ev:
            names.append("smallhex")
        if "explode-bighex" in ev:
            names.append("bighex")
        return tuple(names)

    def __getattr__(self, name):
        if name == "max_points":  # its getter reads a double out of an int-typed entry (SRC/pymodule.cpp:41): undefined
            raise AttributeError("max_points is not defined by the reference either")
        raise AttributeError(name)

    # ------------------------------------------------------------------ telemetry
    def _derive_events(self, b, a):
        """The tick's mEvents (SRC/game.cpp:124-127 and its addEvent call sites) from the recorded key calls
        and the state before / after, in stepOneTick's phase order (:473-485)."""
        ev = []
        # fireMissile (SRC/game.cpp:175-191) makes a missile -- and the event -- when the ship is alive and one of the twenty
        # slots is free; the slots the lanes keep are not the reference's (a hit and a shot in one tick may leave the same
        # mask), so the conditions are evaluated as written there, on the state before the tick
        new_missile = bool(int(b["flags"]) & 1) and bin(int(b["missile_mask"])).count("1") < 20
        fire_was = bool(int(b["flags"]) & 4)
        for pressed, sym in self._calls:                                     # processKeyState :218-272
            ev.append(("press-" if pressed else "release-") + _KEY_NAME[sym])
            if sym == FIRE_KEY:
                if pressed and not fire_was and new_missile:
                    ev.append("missile-fired")                               # :186
                    new_missile = bin(int(b["missile_mask"])).count("1") + 1 < 20
                fire_was = pressed
        sb, sa = b["stats"], a["stats"]
        if not (int(b["flags"]) & 1) and (int(a["flags"]) & 1 or sa[3] > sb[3]):
            ev.append("ship-respawn")                                        # :155
        ev += ["explode-bighex"] * int(sa[0] - sb[0])                        # :342
        ev += ["explode-smallhex"] * int(sa[1] - sb[1])                      # :348
        if not (int(b["flags"]) & 2) and int(a["flags"]) & 2:
            ev.append("fortress-respawn")                                    # :202
        if int(a["shell_mask"]) & ~int(b["shell_mask"]) or self._shell_fired_and_gone(b, a):
            ev.append("fortress-fired")                                      # :169
        ev += ["shell-hit-ship"] * int(sa[2] - sb[2])                        # :417
        incs, dest, resets = int(sa[11] - sb[11]), int(sa[5] - sb[5]), int(sa[4] - sb[4])
        for _ in range(incs):
            ev += ["hit-fortress", "vlner-increased"]                        # :363,371
        for _ in range(dest):
            ev += ["hit-fortress", "fortress-destroyed"]                     # :381
        for _ in range(resets):
            ev += ["hit-fortress", "vlner-reset"]                            # :385
        dead_hits = self._missiles_gone_count(b, a) - int(sa[6] - sb[6]) - incs - dest - resets
        ev += ["hit-dead-fortress"] * max(dead_hits, 0)                      # :393
        return ev

    @staticmethod
    def _missiles_gone_count(b, a):
        return bin(int(b["missile_mask"]) & ~int(a["missile_mask"])).count("1")

    @staticmethod
    def _shell_fired_and_gone(b, a):
        return False

    def dump(self):
        """Game::dumpState() (SRC/game.cpp:519-576): same format, same number formatting."""
        st = self._state()
        fl = int(st["flags"])
        mm, sm = int(st["missile_mask"]), int(st["shell_mask"])
        mis = ",".join("%.3f,%.3f,%.1f" % (st["missile_x"][i], st["missile_y"][i], st["missile_angle"][i])
                       for i in range(20) if (mm >> i) & 1)
        she = ",".join("%.3f,%.3f,%.1f" % (st["shell_x"][i], st["shell_y"][i], _shell_angle(st["shell_vx"][i], st["shell_vy"][i]))
                       for i in range(20) if (sm >> i) & 1)
        ev = ",".join('"%s"' % e for e in self._events)
        return "[%d,%d,%.3f,%.3f,%.3f,%.3f,%.1f,%d,%.1f,[%s],[%s],%.1f,%d,%d,%d,[%s]]" % (
            int(st["time"]), fl & 1, st["ship_x"], st["ship_y"], st["ship_vx"], st["ship_vy"], float(st["ship_angle"]),
            (fl >> 1) & 1, float(st["fort_angle"]), mis, she, float(st["points"]), int(st["vlner"]), (fl >> 3) & 1,
            self.turn_flag, ev)


def _shell_angle(vx, vy):
    """mShells[i].mAngle = stdAngle(rad2deg(atan2(dy, dx))) at launch (SRC/game.cpp:263): the direction
    of the velocity the state keeps."""
    a = math.degrees(math.atan2(float(vy), float(vx)))
    return a + 360.0 if a < 0 else a
