"""`import spacefortress.gym` -- what rl/train.py:7 and rl/envs.py rely on.

Registers the reference's four gym ids with the reference's own kwargs -- `obs_type: 'image'` on every one of them
(python/spacefortress.gym/spacefortress/gym/__init__.py:3-29 of the reference) -- when `gym` is importable, and
always exposes

    SSF_Env       the single-env gym.Env surface (batch of one lane); obs_type='image' yields the bare [92, 90]
                  grey frame, as ENV:171,203-206
    WrapPyTorch   rl/envs.py:19-30: the frame shrunk to 84x84 with INTER_AREA, shape [1, 84, 84] uint8
    make_env      rl/envs.py:10-16's thunk factory: gym.make(env_id) -> seed -> WrapPyTorch
    SFVecEnv      the on-device batch that replaces gym_vecenv.SubprocVecEnv
    make_vec_env(env_id_or_gametype, num_envs, ...)  the whole `SubprocVecEnv([make_env(id, seed, i) ...])` in one call

The symbolic observations ('features', 'normalized-features', 'monitors') are an explicit `obs_type=` away, as they
are in the reference (ENV:50).
"""
import numpy as np

from spacefortress_amd.env import SSF_Env
from spacefortress_amd.vecenv import SFVecEnv

# id -> constructor kwargs, exactly as the reference registers them
ENV_IDS = {
    "SpaceFortress-youturn-image-v0": {"gametype": "youturn", "obs_type": "image"},
    "SpaceFortress-autoturn-image-v0": {"gametype": "autoturn", "obs_type": "image"},
    "SpaceFortress-testyouturn-image-v0": {"gametype": "test-youturn", "obs_type": "image"},
    "SpaceFortress-testautoturn-image-v0": {"gametype": "test-autoturn", "obs_type": "image"},
}
ENTRY_POINT = "spacefortress.gym.envs:SSF_Env"


def _resolve(env_id):
    if env_id in ENV_IDS:
        return dict(ENV_IDS[env_id])
    if env_id in ("youturn", "autoturn", "test-youturn", "test-autoturn"):
        return {"gametype": env_id, "obs_type": "image"}
    raise KeyError("unknown Space Fortress env id %r" % (env_id,))


class WrapPyTorch:
    """rl/envs.py:19-30: `cv2.resize(frame, (84, 84), interpolation=cv2.INTER_AREA)` + a leading axis, and the
    Box(0, 255, [1, 84, 84], uint8) the trainer sizes its network from.  The shrink is the library's INTER_AREA
    (sfmi.h: sf_resize_area_u8 -- OpenCV's published resizeArea_ arithmetic; the same taps the render kernel applies
    on the device for obs_type='image' batches)."""

    def __init__(self, env=None):
        from spacefortress_amd.spaces import Box

        self.env = env
        self.action_space = env.action_space
        self.observation_space = Box(0, 255, [1, 84, 84], dtype=np.uint8)
        self.metadata = getattr(env, "metadata", {})

    def observation(self, observation):
        import ctypes as C

        from spacefortress_amd import _lib

        src = np.ascontiguousarray(observation, np.uint8)
        if src.ndim != 2:
            raise ValueError("WrapPyTorch wraps image observations (a [h, w] grey frame); got shape %s -- build the "
                             "env with obs_type='image' (the registered ids do)" % (src.shape,))
        dst = np.empty((84, 84), np.uint8)
        _lib.check(_lib.lib().sf_resize_area_u8(src.ctypes.data_as(C.c_void_p), src.shape[1], src.shape[0],
                                                dst.ctypes.data_as(C.c_void_p), 84, 84))
        return np.expand_dims(dst, 0)

    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self.observation(obs), reward, done, info

    def __getattr__(self, name):  # gym.Wrapper forwards everything else to the wrapped env
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)


def make_env(env_id, seed, rank, **kw):
    """rl/envs.py:10-16: a thunk `gym.make(env_id)` -> `env.seed(seed + rank)` -> `WrapPyTorch(env)`; its observations
    are [1, 84, 84] uint8.  (env.seed is a no-op for the game, ENV:159-161.)  A symbolic observation is an explicit
    kwarg -- make_env(id, seed, rank, obs_type='features') -- and is not wrapped (there is no frame to shrink)."""
    def _thunk():
        cfg = _resolve(env_id)
        cfg.update(kw)
        env = SSF_Env(**cfg)
        env.seed(seed + rank)
        return WrapPyTorch(env) if cfg["obs_type"] == "image" else env
    return _thunk


def make_vec_env(env_id, num_envs, obs_type="image", spawn_skip=1, **kw):
    """The whole `SubprocVecEnv([make_env(id, seed, i) for i in range(N)])` as one device batch: with the registered
    obs_type 'image' its observations are what rl/envs.py:19-30 wraps each worker into, uint8 [N, 1, 84, 84]; pass
    obs_type='features' (…) for the symbolic vectors.  spawn_skip=1 reproduces the trainer: the parent built one
    throw-away env before forking (rl/train.py:17), so every worker's first Game is the second spawn of the libc
    stream."""
    cfg = _resolve(env_id)
    cfg["obs_type"] = obs_type
    return SFVecEnv(num_envs, spawn_skip=spawn_skip, **cfg, **kw)


def _register():
    """The reference's four `register(...)` calls, when gym is there to take them."""
    try:
        from gym.envs.registration import register
    except Exception:  # gym is not installed in the build image
        return 0
    n = 0
    for _id, _kw in ENV_IDS.items():
        try:
            register(id=_id, entry_point=ENTRY_POINT, kwargs=dict(_kw), nondeterministic=False)
            n += 1
        except Exception:  # already registered (module reloaded)
            pass
    return n


_register()
