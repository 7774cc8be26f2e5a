"""SSF_Env -- the single-environment gym.Env surface of the reference (ENV:43-269), as a
batch-of-one view over the HIP engine.  Same constructor arguments, `action_space`,
`observation_space`, `reset()`, `step(a) -> (obs, int reward, bool done, bool info)`,
`tickdur`, `max_ticks`, `actions_taken`, `g` (the env's Game, ENV:164), `np_random` (ENV:159-161).  Like the reference it
does NOT auto-reset.
"""
import numpy as np

from .vecenv import SFVecEnv

try:  # pragma: no cover
    import gym  # type: ignore

    _Base = gym.Env
except Exception:
    _Base = object


class SSF_Env(_Base):
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 30}  # ENV:45-48

    def __init__(self, gametype="youturn", scale=.2, viewport=(130, 80, 450, 460), ls=3, action_set=1,
                 obs_type="image", device=None, seed=1, ref_reset_obs=False):
        assert obs_type in ("image", "features", "normalized-features", "monitors")  # ENV:51
        # scale / viewport / ls only shape the picture (ENV:56-60 -> sf.Game(width, height, viewport, lw)).  The default
        # geometry has the fast frame kernel (90x92 surface, 0.6-pixel strokes compiled in); any other one is drawn by the
        # general renderer (sfmi.h: sf_set_image_geometry) as long as the surface is 84 .. 251 pixels wide and high -- the
        # trainer's 84x84 INTER_AREA image must be a shrink.  A geometry outside that is accepted for the symbolic
        # observations and refused (ValueError) where a frame would have to be drawn with it.
        self._geometry = (float(scale), tuple(float(v) for v in viewport), float(ls))
        self._default_geometry = self._geometry == (.2, (130., 80., 450., 460.), 3.0)
        self.obs_type = obs_type
        self.gametype = gametype
        self.viewport = viewport
        self.ls = ls
        self.w = int(viewport[2] * scale)
        self.h = int(viewport[3] * scale)
        self.action_set = action_set
        self.last_action = None
        # 'image' here is the bare [92, 90] grey frame (ENV:171); the 84x84 shrink is the trainer's wrapper
        self._vec = SFVecEnv(1, gametype=gametype, obs_type="image-raw" if obs_type == "image" else obs_type,
                             action_set=action_set, device=device,
                             seed=seed, obs_dtype=__import__("torch").float64, auto_reset=False, ref_reset_obs=ref_reset_obs)
        self._drawable = True
        if not self._default_geometry:
            try:
                self._vec.set_image_geometry(scale, viewport, ls)
            except ValueError:
                self._drawable = False
                if obs_type == "image":
                    self._vec.close()
                    raise
        self.tickdur = self._vec.tickdur
        self.max_ticks = float(self._vec.max_ticks)
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        if obs_type == "image":  # ENV:169 declares (h, w, 3) although the observation is the grey (h, w) frame
            from .spaces import Box
            self.observation_space = Box(0, 255, (self.h, self.w, 3), np.uint8)
        self.actions_taken = {i: 0 for i in range(self.action_space.n)}  # ENV:91
        self._g = None
        self.seed()  # ENV:53: __init__ seeds np_random (an RNG the game never reads)
        # ENV:93: __init__ ends with reset(), i.e. the first Game -- sf_create already made it

    @property
    def g(self):
        """The env's Game (ENV:164 `self.g = sf.Game(...)`): a `spacefortress.core.Game`-shaped view over this env's own lane
        (game.py: Game._over) -- `env.g.points`, `.missiles`, `.timers`, `.stats`, `.events`, `.dump()`, `.draw()` /
        `.pb_pixels`.  Made on first use; from then on step() reports the tick's key calls to it (events, duration vectors)."""
        if self._g is None:
            from .game import Game
            self._g = Game._over(self._vec, self.gametype, self.action_set, self.w, self.h)
        return self._g

    def seed(self, seed=None):
        """ENV:159-161: `self.np_random, seed = seeding.np_random(seed)` -- an RNG the game never reads (the ship's spawn comes
        from the C library's rand(), SRC/game.cpp:137-148).  np_random is a numpy RandomState seeded with `seed`."""
        self.np_random = np.random.RandomState(None if seed is None else int(seed) & 0xFFFFFFFF)
        return [seed]

    def reset(self):
        if self._g is not None:
            self._g._new_game()
        return self._vec.reset(numpy=True)[0]

    def step(self, action):
        self.actions_taken[action] += 1  # KeyError for an unknown action, like ENV:211
        if self._g is not None:
            obs, r, d, i = self._g._env_tick(action, lambda: self._vec.step(np.array([action])))
            self._g._obs = obs[0] if self.obs_type == "features" else None
        else:
            obs, r, d, i = self._vec.step(np.array([action]))
        self.last_action = action
        return obs[0], int(r[0]), bool(d[0]), bool(i[0])

    def render(self, mode="human", close=False):
        """ENV:180-198.  'rgb_array' returns game_gray_rgb, the grey frame replicated to [92, 90, 3];
        the pyglet window of mode 'human' is not part of this library."""
        if close:
            return None
        if not self._drawable:
            raise ValueError("render(): a %d x %d surface is outside what the renderer draws (84 .. 251 pixels each way)" % (self.w, self.h))
        if mode != "rgb_array":
            raise NotImplementedError("render(mode='human') opens a pyglet window in the reference; "
                                      "use mode='rgb_array'")
        frame = self._vec.render("image-raw")[0].cpu().numpy()
        return np.repeat(frame[:, :, None], 3, axis=2)  # cv2.COLOR_GRAY2RGB

    def close(self):
        self._vec.close()
