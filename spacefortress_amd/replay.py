"""Replay files: a batch's game on disk, and a device-side player that verifies it.

The reference's only record of a game is the string `Game::dumpState()` builds per tick (SRC/game.cpp:519-576: time,
ship, fortress, projectiles, points, vlner, events; exposed as `Game.dump()`, SRC/pymodule.cpp:290-293), which a caller
may log; nothing in the reference reads such a log back.  The engine is deterministic given (preset, libc seed, the
spawns already drawn, the actions) -- rand() is the only source of chance and feeds resetShip alone (SRC/game.cpp:133-149)
-- so what has to be kept to have the whole game again is small:

    replay.npz
      meta      JSON: format version, gametype, action_set, seed, spawn_skip, spawn_stride, n_envs, auto_reset,
                tick_ms, the library's build id (sf_build_id), the number of steps
      actions   uint8 [T, N]   what every env played (given actions, or the ones sf_step_sampled drew on the device:
                               `actions_out_dev`)
      returns   int64 [N]      sum of the wrapper's rewards per env over the T steps            } what the
      kills     int64 [N]      sum of info (fortress kills the wrapper reported)                 } recorded run
      dones     int64 [N]      episode ends per env                                              } produced:
      digest    the SHA-256 of the final state, field by field (SFVecEnv.state_dict())           } verified by run()

`Replay.run()` creates a fresh batch with the recorded parameters and plays the actions through sf_rollout (K ticks per
launch), checks returns / kills / dones / the final state against the file and raises ReplayMismatch on any difference;
with keep_steps=True it also returns reward / done / info of every step.  The Game view of a lane (spacefortress_amd.game)
gives `dump()` strings for any step of a replayed game: play to that step and ask.
"""
import hashlib
import json

import numpy as np
import torch

from . import _lib

FORMAT = 1


class ReplayMismatch(AssertionError):
    pass


def state_digest(sd):
    """SHA-256 over the fields of a SFVecEnv.state_dict(), in name order (dtype, shape and bytes of each)."""
    h = hashlib.sha256()
    for k in sorted(sd):
        a = np.ascontiguousarray(sd[k])
        h.update(k.encode() + b"\0" + str(a.dtype).encode() + b"\0" + str(a.shape).encode() + b"\0")
        h.update(a.tobytes())
    return h.hexdigest()


class Replay:
    def __init__(self, meta, actions, returns=None, kills=None, dones=None, digest=None):
        self.meta = dict(meta)
        self.actions = np.ascontiguousarray(actions, np.uint8)
        if self.actions.ndim != 2 or self.actions.shape[1] != int(self.meta["n_envs"]):
            raise ValueError("actions must be [T, n_envs]")
        self.meta["steps"] = int(self.actions.shape[0])
        self.returns = None if returns is None else np.asarray(returns, np.int64)
        self.kills = None if kills is None else np.asarray(kills, np.int64)
        self.dones = None if dones is None else np.asarray(dones, np.int64)
        self.digest = digest

    # ------------------------------------------------------------------ files
    def save(self, path):
        arrays = {"meta": np.array(json.dumps(dict(self.meta, format=FORMAT))), "actions": self.actions}
        for k in ("returns", "kills", "dones"):
            if getattr(self, k) is not None:
                arrays[k] = getattr(self, k)
        if self.digest is not None:
            arrays["digest"] = np.array(self.digest)
        with open(path, "wb") as f:  # (a file object: numpy would append ".npz" to a bare name)
            np.savez_compressed(f, **arrays)
        return path

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        meta = json.loads(str(z["meta"]))
        if meta.get("format") != FORMAT:
            raise ValueError("%s: replay format %r, this library reads %d" % (path, meta.get("format"), FORMAT))
        return cls(meta, z["actions"], z["returns"] if "returns" in z else None, z["kills"] if "kills" in z else None,
                   z["dones"] if "dones" in z else None, str(z["digest"]) if "digest" in z else None)

    # ------------------------------------------------------------------ playing
    def make_env(self, device=None, **kw):
        from .vecenv import SFVecEnv

        m = self.meta
        return SFVecEnv(int(m["n_envs"]), gametype=m["gametype"], action_set=int(m["action_set"]), seed=int(m["seed"]),
                        spawn_skip=int(m["spawn_skip"]), spawn_stride=int(m["spawn_stride"]),
                        auto_reset=bool(m.get("auto_reset", True)), obs_type=kw.pop("obs_type", "features"), device=device, **kw)

    def run(self, device=None, chunk=256, keep_steps=False, verify=True, env=None):
        """Play the file on the device (sf_rollout, `chunk` ticks per launch).  Returns a dict: returns / kills / dones per
        env, digest of the final state, `env` (the batch in its final state; the caller closes it) and, with keep_steps,
        reward / done / info of every step as numpy [T, N].  verify: compare with what the file recorded."""
        own = env is None
        env = env or self.make_env(device)
        T, n = self.actions.shape
        dev = env.device
        ret = torch.zeros(n, dtype=torch.int64, device=dev)
        kil = torch.zeros(n, dtype=torch.int64, device=dev)
        don = torch.zeros(n, dtype=torch.int64, device=dev)
        steps = {"reward": [], "done": [], "info": []} if keep_steps else None
        for c in range(0, T, chunk):
            a = torch.from_numpy(self.actions[c:c + chunk]).to(dev)
            _, rew, done, info = env.rollout(a, want_obs=False)
            ret += rew.sum(0, dtype=torch.int64)
            kil += info.sum(0, dtype=torch.int64)
            don += done.sum(0, dtype=torch.int64)
            if keep_steps:
                steps["reward"].append(rew.cpu().numpy())
                steps["done"].append(done.cpu().numpy().astype(bool))
                steps["info"].append(info.cpu().numpy().astype(bool))
        env.check_actions()
        out = {"returns": ret.cpu().numpy(), "kills": kil.cpu().numpy(), "dones": don.cpu().numpy(),
               "digest": state_digest(env.state_dict()), "env": env}
        if keep_steps:
            out.update({k: np.concatenate(v) if v else np.zeros((0, n)) for k, v in steps.items()})
        if verify:
            bad = [k for k in ("returns", "kills", "dones") if getattr(self, k) is not None and not np.array_equal(getattr(self, k), out[k])]
            if self.digest is not None and self.digest != out["digest"]:
                bad.append("digest")
            if bad:
                lanes = {k: np.flatnonzero(getattr(self, k) != out[k])[:5].tolist() for k in bad if k != "digest"}
                if own:
                    env.close()
                raise ReplayMismatch("replay of %d steps x %d envs differs from the recording in %s (first lanes: %s); recorded with "
                                     "build %s, played with %s" % (T, n, bad, lanes, self.meta.get("build_id"), build_id()))
        return out


def build_id():
    return _lib.lib().sf_build_id().decode()


class Recorder:
    """What SFVecEnv.start_recording() keeps: the parameters the batch was made with and every action row played since."""

    def __init__(self, env):
        self.rows = []
        self.meta = {"gametype": env.gametype, "action_set": env._create["action_set"], "seed": env._create["seed"],
                     "spawn_skip": env._create["spawn_skip"], "spawn_stride": env._create["spawn_stride"], "n_envs": env.num_envs,
                     "auto_reset": env._create["auto_reset"], "tick_ms": env.tickdur, "build_id": build_id()}
        n = env.num_envs
        self.ret = torch.zeros(n, dtype=torch.int64, device=env.device)
        self.kil = torch.zeros(n, dtype=torch.int64, device=env.device)
        self.don = torch.zeros(n, dtype=torch.int64, device=env.device)

    def add(self, actions_u8, rew, done, info):
        """One step ([N] tensors) or a fused launch's K steps ([K, N])."""
        a = actions_u8.reshape(-1, self.ret.numel())
        self.rows.append(a.clone())
        self.ret += rew.reshape(a.shape).sum(0, dtype=torch.int64)
        self.kil += info.reshape(a.shape).sum(0, dtype=torch.int64)
        self.don += done.reshape(a.shape).sum(0, dtype=torch.int64)

    def replay(self, env):
        acts = torch.cat(self.rows).cpu().numpy() if self.rows else np.zeros((0, env.num_envs), np.uint8)
        return Replay(self.meta, acts, self.ret.cpu().numpy(), self.kil.cpu().numpy(), self.don.cpu().numpy(), state_digest(env.state_dict()))
