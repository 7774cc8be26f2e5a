"""`import spacefortress.gym` -- what rl/train.py:7 and rl/envs.py rely on.

Registers the reference's four gym ids (python/spacefortress.gym/spacefortress/gym/__init__.py:3-29
of the reference) when `gym` is importable, and always exposes

    SSF_Env     the single-env gym.Env surface (batch of one lane)
    SFVecEnv    the on-device batch that replaces gym_vecenv.SubprocVecEnv
    make_env    rl/envs.py:10-16's thunk factory
    make_vec_env(env_id_or_gametype, num_envs, ...)  one call instead of SubprocVecEnv([thunks])
"""
from spacefortress_amd.env import SSF_Env
from spacefortress_amd.vecenv import SFVecEnv

# id -> constructor kwargs, as registered by the reference (obs_type 'image' there; pass
# obs_type="image" for that, the default here stays the symbolic 'features' observation)
ENV_IDS = {
    "SpaceFortress-youturn-image-v0": {"gametype": "youturn"},
    "SpaceFortress-autoturn-image-v0": {"gametype": "autoturn"},
    "SpaceFortress-testyouturn-image-v0": {"gametype": "test-youturn"},
    "SpaceFortress-testautoturn-image-v0": {"gametype": "test-autoturn"},
}


def _resolve(env_id):
    if env_id in ENV_IDS:
        return dict(ENV_IDS[env_id])
    if env_id in ("youturn", "autoturn", "test-youturn", "test-autoturn"):
        return {"gametype": env_id}
    raise KeyError("unknown Space Fortress env id %r" % (env_id,))


def make_env(env_id, seed, rank, obs_type="features", **kw):
    """rl/envs.py:10-16: a thunk that builds one env (env.seed is a no-op for the game, ENV:159-161)."""
    def _thunk():
        env = SSF_Env(obs_type=obs_type, **_resolve(env_id), **kw)
        env.seed(seed + rank)
        return env
    return _thunk


def make_vec_env(env_id, num_envs, obs_type="features", spawn_skip=1, **kw):
    """The whole `SubprocVecEnv([make_env(id, seed, i) for i in range(N)])` as one device batch.
    obs_type="image" yields what rl/envs.py:19-30 wraps each worker into: uint8 [N, 1, 84, 84].
    spawn_skip=1 reproduces the trainer: the parent built one throw-away env before forking
    (rl/train.py:17), so every worker's first Game is the second spawn of the libc stream."""
    return SFVecEnv(num_envs, obs_type=obs_type, spawn_skip=spawn_skip, **_resolve(env_id), **kw)


try:  # pragma: no cover - gym is not installed in the build image
    from gym.envs.registration import register

    for _id, _kw in ENV_IDS.items():
        try:
            register(id=_id, entry_point="spacefortress.gym.envs:SSF_Env",
                     kwargs=dict(_kw, obs_type="features"), nondeterministic=False)
        except Exception:
            pass
except Exception:
    pass
