"""SSF_Env -- the single-environment gym.Env surface of the reference (ENV:43-269), as a
batch-of-one view over the HIP engine.  Same constructor arguments, `action_space`,
`observation_space`, `reset()`, `step(a) -> (obs, int reward, bool done, bool info)`,
`tickdur`, `max_ticks`, `actions_taken`.  Like the reference it does NOT auto-reset.
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
                 obs_type="image", device=None, seed=1):
        assert obs_type in ("image", "features", "normalized-features", "monitors")  # ENV:51
        if obs_type == "image":
            raise NotImplementedError("image observations (SURVEY 8f rank 1) are not built yet; "
                                      "use obs_type='features'")
        self.obs_type = obs_type
        self.gametype = gametype
        self.viewport = viewport
        self.ls = ls
        self.w = int(viewport[2] * scale)
        self.h = int(viewport[3] * scale)
        self.action_set = action_set
        self.last_action = None
        self._vec = SFVecEnv(1, gametype=gametype, obs_type=obs_type, action_set=action_set, device=device,
                             seed=seed, obs_dtype=__import__("torch").float64, auto_reset=False)
        self.tickdur = self._vec.tickdur
        self.max_ticks = float(self._vec.max_ticks)
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self.actions_taken = {i: 0 for i in range(self.action_space.n)}  # ENV:91
        # ENV:93: __init__ ends with reset(), i.e. the first Game -- sf_create already made it

    def seed(self, seed=None):  # ENV:159-161: seeds an RNG the game never reads
        return [seed]

    def reset(self):
        return self._vec.reset(numpy=True)[0]

    def step(self, action):
        self.actions_taken[action] += 1  # KeyError for an unknown action, like ENV:211
        obs, r, d, i = self._vec.step(np.array([action]))
        self.last_action = action
        return obs[0], int(r[0]), bool(d[0]), bool(i[0])

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering is outside the env.step() hot path")

    def close(self):
        self._vec.close()
